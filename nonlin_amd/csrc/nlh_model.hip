// nlh_model.hip -- several GPUs behind the boundary (device sets) and device residual models behind host arrays: what
// a Fortran / C caller without device pointers uses to reach the batched device path (the extension of vecfcn_helper
// SURVEY.md section 7 asks for: set_device_model).
#include "nlh_internal.h"


// ===========================================================================
// Device residual models behind host arrays: what a Fortran / C caller without device pointers uses to reach the
// batched device path (the extension of vecfcn_helper SURVEY.md section 7 asks for: set_device_model).
// A model owns device copies of the data of nprob dense-quadratic problems (SURVEY 8(d) family:
// r = (u + gamma u u) - b, u = A x); the solves stage x / fvec through the handle's buffers.
// ===========================================================================
// ---- several GPUs behind the boundary (SURVEY 8(b) `nlx_init(device, comm)`, 8(e)) ---------------------------------
// A device set owns one handle -- own non-blocking stream, own workspaces -- per entry of its device list.  A model
// created on a set is DEALT over the entries block-cyclically (problem k -> entry k mod ndev: iteration counts differ
// per problem) and a solve on it runs one host thread per entry: independent problems, no collective, the same bits as
// on one device (a problem's arithmetic never depends on its batch).
struct nlh_device_set {
    std::vector<nlh_handle *> handles;
    std::string err;
    std::atomic<int> refs{1};          // the creator's reference + one per model dealt over the set: nlh_device_set_destroy
};                                     // only drops the creator's, the handles go when the last model has gone too

int nlh_device_set_create(nlh_device_set **out, const int32_t *devices, int32_t ndev)
{
    if (!out) return NLH_ERR_BAD_HANDLE;
    *out = nullptr;
    const int visible = nlh_device_count();
    if (visible <= 0) return NLH_ERR_NO_DEVICE;
    std::vector<int32_t> ids;
    if (!devices || ndev <= 0) for (int d = 0; d < visible; ++d) ids.push_back(d);
    else ids.assign(devices, devices + ndev);
    for (int32_t d : ids) if (d < 0 || d >= visible) return NLH_INVALID_INPUT_ERROR;
    nlh_device_set *set = new nlh_device_set();
    for (int32_t d : ids) {
        nlh_handle *h = nullptr;
        hipStream_t st = nullptr;
        int rc = hipSetDevice(d) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess ? 0 : NLH_ERR_HIP;
        if (!rc) rc = nlh_create(&h, d, st);
        if (rc) {
            if (st) hipStreamDestroy(st);
            nlh_device_set_destroy(set);
            return rc;
        }
        h->own_stream = true;
        set->handles.push_back(h);
    }
    *out = set;
    return 0;
}

static void device_set_release(nlh_device_set *set)
{
    if (!set || set->refs.fetch_sub(1) != 1) return;
    for (auto *h : set->handles) nlh_destroy(h);
    delete set;
}

void nlh_device_set_destroy(nlh_device_set *set) { device_set_release(set); }

int32_t nlh_device_set_size(const nlh_device_set *set) { return set ? (int32_t)set->handles.size() : 0; }

nlh_handle *nlh_device_set_handle(nlh_device_set *set, int32_t i)
{
    return (set && i >= 0 && i < (int32_t)set->handles.size()) ? set->handles[i] : nullptr;
}

const char *nlh_device_set_last_error(const nlh_device_set *set) { return set ? set->err.c_str() : "null device set"; }

// ---- device residual models behind host arrays ------------------------------------------------------------------------
struct DqPart {                        // the share of one device: problems first, first + stride, ... (cnt of them)
    int32_t device = 0, cnt = 0, first = 0, stride = 1;
    nlh_handle *h = nullptr;           // the set's handle for this share; NULL: the caller's handle (single-device model)
    double *dA = nullptr, *db = nullptr, *dx = nullptr, *df = nullptr;
};

struct nlh_dq_model {
    int32_t nprob, m, n;
    double gamma;
    nlh_device_set *set = nullptr;     // not owned
    std::vector<DqPart> parts;
    // a USER'S device residual instead of the dense-quadratic family (nlh_device_fcn_model_create): launchers + context,
    // no data of the library's own; the solves go through the *_batch_device_h entry points on the caller's handle
    nlh_device_vecfcn ufcn = nullptr;
    nlh_device_jacfcn ujac = nullptr;
    void *uctx = nullptr;
};

int nlh_device_fcn_model_create(int32_t nprob, int32_t m, int32_t n, nlh_device_vecfcn fcn, nlh_device_jacfcn jacfcn, void *ctx,
                                nlh_dq_model **out)
{
    if (!out || nprob < 1 || m < 1 || n < 1) return NLH_INVALID_INPUT_ERROR;
    *out = nullptr;
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;
    nlh_dq_model *md = new nlh_dq_model();
    md->nprob = nprob; md->m = m; md->n = n; md->gamma = 0.0;
    md->ufcn = fcn; md->ujac = jacfcn; md->uctx = ctx;
    *out = md;
    return 0;
}

static int model_part_upload(nlh_handle *h, const nlh_dq_model *md, DqPart &pt, const double *A, const double *b)
{
    const int m = md->m, n = md->n;
    const size_t mn = (size_t)m * n, cnt = (size_t)pt.cnt;
    HIPCHK(h, hipSetDevice(pt.device));
    double *base = nullptr;
    if (hipMalloc(&base, sizeof(double) * cnt * (mn + 2 * (size_t)m + n)) != hipSuccess) {
        h->err = "hipMalloc (device model)";
        return NLH_OUT_OF_MEMORY_ERROR;
    }
    pt.dA = base; pt.db = base + cnt * mn; pt.df = pt.db + cnt * m; pt.dx = pt.df + cnt * m;
    hipError_t e = hipSuccess;
    if (pt.stride == 1) {
        e = hipMemcpyAsync(pt.dA, A + (size_t)pt.first * mn, sizeof(double) * cnt * mn, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(pt.db, b + (size_t)pt.first * m, sizeof(double) * cnt * m, hipMemcpyHostToDevice, h->stream);
    } else {
        for (size_t i = 0; i < cnt && e == hipSuccess; ++i) {
            const size_t k = (size_t)pt.first + i * pt.stride;
            e = hipMemcpyAsync(pt.dA + i * mn, A + k * mn, sizeof(double) * mn, hipMemcpyHostToDevice, h->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(pt.db + i * m, b + k * m, sizeof(double) * m, hipMemcpyHostToDevice, h->stream);
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        hipFree(base);
        pt.dA = nullptr;
        h->err = std::string("hipMemcpy (device model): ") + hipGetErrorString(e);
        return NLH_ERR_HIP;
    }
    return 0;
}

int nlh_dq_model_create(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, const double *A, const double *b,
                        double gamma, nlh_dq_model **out)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (!out || !A || !b || nprob < 1 || m < 1 || n < 1) return NLH_INVALID_INPUT_ERROR;
    *out = nullptr;
    nlh_dq_model *md = new nlh_dq_model();
    md->nprob = nprob; md->m = m; md->n = n; md->gamma = gamma;
    DqPart pt;
    pt.device = h->device; pt.cnt = nprob;
    const int rc = model_part_upload(h, md, pt, A, b);
    if (rc) { delete md; return rc; }
    md->parts.push_back(pt);
    *out = md;
    return 0;
}

int nlh_dq_model_create_on(nlh_device_set *set, int32_t nprob, int32_t m, int32_t n, const double *A, const double *b,
                           double gamma, nlh_dq_model **out)
{
    if (!set || set->handles.empty()) return NLH_ERR_BAD_HANDLE;
    if (!out || !A || !b || nprob < 1 || m < 1 || n < 1) return NLH_INVALID_INPUT_ERROR;
    *out = nullptr;
    nlh_dq_model *md = new nlh_dq_model();
    md->nprob = nprob; md->m = m; md->n = n; md->gamma = gamma; md->set = set;
    set->refs.fetch_add(1);                                      // released by nlh_dq_model_destroy
    const int nd = (int)set->handles.size();
    for (int d = 0; d < nd; ++d) {
        DqPart pt;
        pt.h = set->handles[d];
        pt.device = pt.h->device; pt.first = d; pt.stride = nd;
        pt.cnt = d < nprob ? (nprob - d + nd - 1) / nd : 0;
        if (pt.cnt > 0) {
            const int rc = model_part_upload(pt.h, md, pt, A, b);
            if (rc) { set->err = pt.h->err; nlh_dq_model_destroy(md); return rc; }
        }
        md->parts.push_back(pt);
    }
    *out = md;
    return 0;
}

void nlh_dq_model_destroy(nlh_dq_model *md)
{
    if (!md) return;
    for (auto &pt : md->parts)
        if (pt.dA) { hipSetDevice(pt.device); hipFree(pt.dA); }
    device_set_release(md->set);
    delete md;
}

void nlh_dq_model_shape(const nlh_dq_model *md, int32_t *nprob, int32_t *m, int32_t *n)
{
    if (nprob) *nprob = md ? md->nprob : 0;
    if (m) *m = md ? md->m : 0;
    if (n) *n = md ? md->n : 0;
}

int32_t nlh_dq_model_device_count(const nlh_dq_model *md) { return md ? (int32_t)md->parts.size() : 0; }

// One operation on every share of a model.  x [nprob][n] goes in (and, when x_out, comes back), per_part works on the
// share's device buffers (pt.dx in / out, pt.df out) and fills the share's ib / status rows; f [nprob][m] comes back.
// A single-device model runs on the caller's handle and stream; a model on a device set runs one host thread per share.
typedef std::function<int(nlh_handle *, const DqPart &, nlh_iteration_behavior *, int32_t *)> PartOp;

static int model_part_run(nlh_handle *h, const nlh_dq_model *md, const DqPart &pt, double *x, bool x_out, double *f,
                          nlh_iteration_behavior *ib, int32_t *status, const PartOp &op)
{
    if (pt.cnt == 0) return 0;
    const size_t n = md->n, m = md->m, cnt = pt.cnt;
    HIPCHK(h, hipSetDevice(pt.device));
    if (pt.stride == 1) {                                        // contiguous share: straight from / to the caller's arrays
        double *xs = x + (size_t)pt.first * n, *fs = f + (size_t)pt.first * m;
        HIPCHK(h, hipMemcpyAsync(pt.dx, xs, sizeof(double) * cnt * n, hipMemcpyHostToDevice, h->stream));
        const int rc = op(h, pt, ib ? ib + pt.first : nullptr, status ? status + pt.first : nullptr);
        if (rc) return rc;
        if (x_out) HIPCHK(h, hipMemcpyAsync(xs, pt.dx, sizeof(double) * cnt * n, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(fs, pt.df, sizeof(double) * cnt * m, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return 0;
    }
    // the broadcast / gather ends of the dealt batch, staged through a PINNED buffer of the share's handle (its own, apart
    // from the one the solvers keep their read-back state in: a pageable staging vector makes every one of these copies a
    // synchronous bounce through the runtime's own pinned pool)
    if (int rcp = ensure_staging(h, sizeof(double) * cnt * (n + m))) return rcp;
    double *xs = (double *)h->staging, *fs = xs + cnt * n;
    std::vector<nlh_iteration_behavior> ibs(ib ? cnt : 0);
    std::vector<int32_t> sts(status ? cnt : 0);
    for (size_t i = 0; i < cnt; ++i) memcpy(&xs[i * n], x + ((size_t)pt.first + i * pt.stride) * n, sizeof(double) * n);
    HIPCHK(h, hipMemcpyAsync(pt.dx, xs, sizeof(double) * cnt * n, hipMemcpyHostToDevice, h->stream));
    const int rc = op(h, pt, ib ? ibs.data() : nullptr, status ? sts.data() : nullptr);
    if (rc) return rc;
    if (x_out) HIPCHK(h, hipMemcpyAsync(xs, pt.dx, sizeof(double) * cnt * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(fs, pt.df, sizeof(double) * cnt * m, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < cnt; ++i) {
        const size_t k = (size_t)pt.first + i * pt.stride;
        if (x_out) memcpy(x + k * n, &xs[i * n], sizeof(double) * n);
        memcpy(f + k * m, &fs[i * m], sizeof(double) * m);
        if (ib) ib[k] = ibs[i];
        if (status) status[k] = sts[i];
    }
    return 0;
}

static int model_run(nlh_handle *h, const nlh_dq_model *md, double *x, bool x_out, double *f, nlh_iteration_behavior *ib,
                     int32_t *status, const PartOp &op)
{
    if (!md || !x || !f) return NLH_INVALID_INPUT_ERROR;
    if (!md->set) {
        if (!h) return NLH_ERR_BAD_HANDLE;
        if (h->device != md->parts[0].device) {                  // the model's buffers live on the device of the handle that
            h->err = "device model used with a handle on another device";   // created it: another device's stream cannot run it
            return NLH_INVALID_INPUT_ERROR;
        }
        return model_part_run(h, md, md->parts[0], x, x_out, f, ib, status, op);
    }
    const int nd = (int)md->parts.size();
    std::vector<int> rcs(nd, 0);
    std::vector<std::thread> pool;
    for (int d = 0; d < nd; ++d)
        pool.emplace_back([&, d]() {
            rcs[d] = model_part_run(md->parts[d].h, md, md->parts[d], x, x_out, f, ib, status, op);
        });
    for (auto &t : pool) t.join();
    for (int d = 0; d < nd; ++d)
        if (rcs[d]) { md->set->err = md->parts[d].h->err; return rcs[d]; }
    return 0;
}

// vecfcn of the model: f = F(x) for every problem, host arrays x [nprob][n], f [nprob][m].
int nlh_dq_model_eval(nlh_handle *h, const nlh_dq_model *md, const double *x, double *f)
{
    if (!md) return NLH_INVALID_INPUT_ERROR;
    if (md->ufcn) {                                              // a user's device function: one point per problem
        if (!h) return NLH_ERR_BAD_HANDLE;
        if (!x || !f) return NLH_INVALID_INPUT_ERROR;
        HIPCHK(h, hipSetDevice(h->device));
        int rc;
        const size_t nx = (size_t)md->nprob * md->n, nf = (size_t)md->nprob * md->m;
        if ((rc = ensure(h, h->xdev, sizeof(double) * nx))) return rc;
        if ((rc = ensure(h, h->fdev, sizeof(double) * nf))) return rc;
        HIPCHK(h, hipMemcpyAsync(h->xdev.p, x, sizeof(double) * nx, hipMemcpyHostToDevice, h->stream));
        ResidualSource rs;
        rs.fcn = md->ufcn; rs.jac = md->ujac; rs.ctx = md->uctx;
        if ((rc = residual_eval(h, rs, md->nprob, md->m, md->n, (const double *)h->xdev.p, (double *)h->fdev.p, nullptr, nullptr, -1))) return rc;
        HIPCHK(h, hipMemcpyAsync(f, h->fdev.p, sizeof(double) * nf, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return 0;
    }
    return model_run(h, md, const_cast<double *>(x), false, f, nullptr, nullptr,
                     [&](nlh_handle *ph, const DqPart &pt, nlh_iteration_behavior *, int32_t *) -> int {
                         launch_dq_residual(ph, pt.cnt, md->m, md->n, pt.dA, pt.db, md->gamma, pt.dx, pt.df, nullptr, nullptr, -1);
                         HIPCHK(ph, hipGetLastError());          // a launch that failed must not return stale residuals as success
                         return 0;
                     });
}

// least_squares_solver%solve on every problem of the model (nlh_dq_lm_solve_batch behind host arrays).
int nlh_dq_model_lm_solve(nlh_handle *h, const nlh_options *o, const nlh_dq_model *md, double *x, double *fvec,
                          nlh_iteration_behavior *ib, int32_t *status)
{
    if (!md || !o) return NLH_INVALID_INPUT_ERROR;
    // a batch stays silent (the reference prints between the iterations of ONE solve): a share of a dealt batch may hold a
    // single problem and would otherwise print from its host thread
    nlh_options oq = *o;
    if (md->nprob > 1) oq.print_status = 0;
    o = &oq;
    if (md->ufcn) return nlh_lm_solve_batch_device_h(h, o, md->nprob, md->m, md->n, md->ufcn, md->ujac, md->uctx, x, fvec, ib, status);
    return model_run(h, md, x, true, fvec, ib, status,
                     [&](nlh_handle *ph, const DqPart &pt, nlh_iteration_behavior *pib, int32_t *pst) -> int {
                         return nlh_dq_lm_solve_batch(ph, o, pt.cnt, md->m, md->n, pt.dA, pt.db, md->gamma, pt.dx, pt.df, pib, pst);
                     });
}

// newton_solver%solve on every (square) problem of the model; analytic != 0: the model's own Jacobian
// J(i,j) = (1 + 2 gamma u_i) A(i,j) plays the role of a jacobianfcn, otherwise forward differences.
int nlh_dq_model_newton_solve(nlh_handle *h, const nlh_options *o, const nlh_dq_model *md, int32_t analytic, double *x,
                              double *fvec, nlh_iteration_behavior *ib, int32_t *status)
{
    if (!md || !o) return NLH_INVALID_INPUT_ERROR;
    nlh_options oq = *o;
    if (md->nprob > 1) oq.print_status = 0;               // (see nlh_dq_model_lm_solve)
    o = &oq;
    if (md->m != md->n) return NLH_INVALID_INPUT_ERROR;         // src/nonlin_solve.f90:519
    if (md->ufcn)                                                // (analytic: whether to use the user's jacobianfcn launcher)
        return nlh_newton_solve_batch_device_h(h, o, md->nprob, md->n, md->ufcn, analytic ? md->ujac : nullptr, md->uctx, x, fvec, ib, status);
    return model_run(h, md, x, true, fvec, ib, status,
                     [&](nlh_handle *ph, const DqPart &pt, nlh_iteration_behavior *pib, int32_t *pst) -> int {
                         return nlh_dq_newton_solve_batch(ph, o, pt.cnt, md->n, pt.dA, pt.db, md->gamma, analytic, pt.dx, pt.df, pib, pst);
                     });
}

// quasi_newton_solver%solve on every (square) problem of the model.
int nlh_dq_model_quasi_newton_solve(nlh_handle *h, const nlh_options *o, const nlh_dq_model *md, int32_t jdelta,
                                    int32_t analytic, double *x, double *fvec, nlh_iteration_behavior *ib, int32_t *status)
{
    if (!md || !o) return NLH_INVALID_INPUT_ERROR;
    nlh_options oq = *o;
    if (md->nprob > 1) oq.print_status = 0;               // (see nlh_dq_model_lm_solve)
    o = &oq;
    if (md->m != md->n) return NLH_INVALID_INPUT_ERROR;         // src/nonlin_solve.f90:241
    if (md->ufcn)
        return nlh_quasi_newton_solve_batch_device_h(h, o, jdelta, md->nprob, md->n, md->ufcn, analytic ? md->ujac : nullptr, md->uctx, x, fvec,
                                                     ib, status);
    return model_run(h, md, x, true, fvec, ib, status,
                     [&](nlh_handle *ph, const DqPart &pt, nlh_iteration_behavior *pib, int32_t *pst) -> int {
                         return nlh_dq_quasi_newton_solve_batch(ph, o, jdelta, pt.cnt, md->n, pt.dA, pt.db, md->gamma, analytic, pt.dx, pt.df, pib, pst);
                     });
}

// constrained_least_squares_solver%solve on every problem of the model; xl / xu: n entries each (or NULL), the same box
// for every problem.
int nlh_dq_model_cls_solve(nlh_handle *h, const nlh_options *o, const nlh_dq_model *md, double delta0, double stepscale0,
                           const double *xl, const double *xu, double *x, double *fvec, nlh_iteration_behavior *ib,
                           int32_t *status)
{
    if (!md || !o) return NLH_INVALID_INPUT_ERROR;
    nlh_options oq = *o;
    if (md->nprob > 1) oq.print_status = 0;               // (see nlh_dq_model_lm_solve)
    o = &oq;
    if (md->ufcn) return nlh_cls_solve_batch_device_h(h, o, delta0, stepscale0, xl, xu, md->nprob, md->m, md->n, md->ufcn, md->ujac, md->uctx, x,
                                                      fvec, ib, status);
    return model_run(h, md, x, true, fvec, ib, status,
                     [&](nlh_handle *ph, const DqPart &pt, nlh_iteration_behavior *pib, int32_t *pst) -> int {
                         return nlh_dq_cls_solve_batch(ph, o, delta0, stepscale0, xl, xu, pt.cnt, md->m, md->n, pt.dA, pt.db, md->gamma,
                                                       pt.dx, pt.df, pib, pst);
                     });
}

// bfgs%solve on 0.5 ||F(x)||^2 of every problem of the model (forward-difference gradient); fout [nprob]: the objective
// at the solution, fvec [nprob][m]: F there.
int nlh_dq_model_bfgs_solve(nlh_handle *h, const nlh_options *o, const nlh_dq_model *md, double *x, double *fvec, double *fout,
                            nlh_iteration_behavior *ib, int32_t *status)
{
    if (!md || !o) return NLH_INVALID_INPUT_ERROR;
    nlh_options oq = *o;
    if (md->nprob > 1) oq.print_status = 0;               // (see nlh_dq_model_lm_solve)
    o = &oq;
    if (md->ufcn) {
        // bfgs minimises a scalar fcnnvar: a user's model with ONE function is exactly that (its launcher is called with
        // m = 1, its jacobianfcn launcher is the gradient); a vector launcher is not bfgs's plugin
        if (md->m != 1) return NLH_INVALID_OPERATION_ERROR;
        const int rc = nlh_bfgs_solve_batch_device_h(h, o, md->nprob, md->n, md->ufcn, md->ujac, md->uctx, x, fout, ib, status);
        if (rc) return rc;
        if (fvec && fout)                                     // (the model's one "equation": f at the solution)
            for (int32_t p = 0; p < md->nprob; ++p) fvec[p] = fout[p];
        return 0;
    }
    return model_run(h, md, x, true, fvec, ib, status,
                     [&](nlh_handle *ph, const DqPart &pt, nlh_iteration_behavior *pib, int32_t *pst) -> int {
                         std::vector<double> fo(pt.cnt, 0.0);
                         const int rc = nlh_dq_bfgs_solve_batch(ph, o, pt.cnt, md->m, md->n, pt.dA, pt.db, md->gamma, pt.dx, fo.data(), pib, pst);
                         if (rc) return rc;
                         if (fout)
                             for (int i = 0; i < pt.cnt; ++i) fout[(size_t)pt.first + (size_t)i * pt.stride] = fo[i];
                         launch_dq_residual(ph, pt.cnt, md->m, md->n, pt.dA, pt.db, md->gamma, pt.dx, pt.df, nullptr, nullptr, -1);
                         HIPCHK(ph, hipGetLastError());
                         return 0;
                     });
}
