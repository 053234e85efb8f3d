// nlh_internal.h -- what the translation units of libnonlin_hip.so share: the handle, the error macro, the event
// brackets of a kernel group, and the helpers one unit defines for the others (nlh_core.hip unless noted).
// Units: nlh_core.hip (handle, options, timing, generator, residual / FD launches, worker handles), nlh_lm.hip
// (least_squares_solver: lss_solve and its stages), nlh_square.hip (newton_solver, quasi_newton_solver, LU, the
// Householder steps), nlh_cls.hip (constrained_least_squares_solver), nlh_bfgs.hip (bfgs, fcnnvar_helper%gradient),
// nlh_poly.hip (polynomial%fit), nlh_model.hip (device sets, device residual models behind host arrays), nlh_qrx.hip
// (the exact lmfactor).  Kernels live in the nlh_kernels_*.h headers with internal linkage: a unit compiles the ones it
// launches.
#pragma once
#include "../../include/nonlin_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <atomic>
#include <string>
#include <vector>

#include "nlh_common.h"
#include "nlh_lm_head.h"


// ---------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct nlh_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    uint32_t timing = 0;              // bit k: kernel group k is bracketed by HIP events
    struct Pair { hipEvent_t a, b; int kid; };
    std::vector<Pair> pending;
    std::vector<hipEvent_t> pool;
    double ms[NLH_K_COUNT] = {0};
    int64_t launches[NLH_K_COUNT] = {0};
    int sample_kid = -1;              // kernel group whose per-launch durations are kept (nlh_timing_samples)
    std::vector<float> samples;
    std::vector<DevBuf *> bufs;       // every workspace buffer, for destroy
    // named workspace buffers (grown on demand, reused across calls)
    DevBuf J, P, wa4, scratch, G, Gpart, vecs, ipvt, gvec, part, state, info, misc, lu, xdev, fdev, Adev, bdev, W2, R,
           qnQ, qnR, qnV, bfB, bfR, bfV, qxV, lumv,
           dvX, dvF, dvIdx, dvP;          // user device residuals: points, compact residuals, problem lists, panel chunk (nlh_devfcn.hip)
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
    DevBuf cholmc;                     // side buffer of the multi-CU Cholesky (solved panels, bad-pivot flags)
    void *staging = nullptr;           // pinned staging of a device-set share's rows of the caller's host arrays
    size_t staging_bytes = 0;
    bool qrx_open_on = false; hipEvent_t qrx_a{}, qrx_b{}; int qrx_kid = 0;   // open bracket of a nlh_qrx.hip launch
    std::vector<nlh_handle *> workers;   // private handles (own stream + workspace) for concurrent host-loop solves
    hipStream_t lu_side = nullptr;       // the blocked LU's look-ahead: the bulk of a step's update runs here, under the next panel
    hipEvent_t lu_panel_done = nullptr, lu_bulk_done[2] = {nullptr, nullptr};
};

#define HIPCHK(h, call)                                                                 \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);               \
            return NLH_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)


int ensure(nlh_handle *h, DevBuf &b, size_t bytes);              // grow-on-demand device workspace, registered for destroy
int ensure_staging(nlh_handle *h, size_t bytes);
int ensure_pinned(nlh_handle *h, size_t bytes);
void timing_flush(nlh_handle *h);
hipEvent_t ev_get(nlh_handle *h);

struct Timed {
    nlh_handle *h; int kid; hipEvent_t a{}, b{}; bool on;
    Timed(nlh_handle *h_, int kid_) : h(h_), kid(kid_), on((h_->timing >> kid_) & 1u)
    {
        if (on) { a = ev_get(h); b = ev_get(h); hipEventRecord(a, h->stream); }
    }
    ~Timed()
    {
        if (on) {
            hipEventRecord(b, h->stream);
            h->pending.push_back({a, b, kid});
            if (h->pending.size() > 65536) timing_flush(h);
        }
    }
};
int ensure_workers(nlh_handle *h, int T);
int run_problems(nlh_handle *h, int nprob, const std::function<int(nlh_handle *, int)> &solve_one);

// per-unit kernel attributes (dynamic LDS limits), called by nlh_create on the handle's device
void nlh_lm_init_device(int lds_max);
void nlh_square_init_device(int lds_max);

// launches of the residual family (nlh_core.hip)
void launch_dq_residual(nlh_handle *h, int nprob, int m, int n, const double *A, const double *b, double gamma, const double *x,
                        double *f, double *part, const LmState *st, int want);
void launch_dq_panel(nlh_handle *h, int nprob, int m, int n, const double *A, const double *b, double gamma, const double *x,
                     double *P, const LmState *st, int want, const double *f0_fused = nullptr, bool to_qrx = false);
void launch_fd(nlh_handle *h, int nprob, int m, int n, const double *P, const double *f0, const double *x, double *J,
               const LmState *st, int want);
// Which residual a lock-step driver evaluates (nlh_devfcn.hip): the built-in dense-quadratic family (dA, db, gamma), or a
// user's launchers (include/nonlin_hip.h: nlh_device_vecfcn / nlh_device_jacfcn).  pbase: index, in the caller's batch,
// of the first problem of the range the driver works on (slices, sub-batches) -- what the user's dprob entries count from.
struct ResidualSource {
    const double *dA = nullptr, *db = nullptr;
    double gamma = 0.0;
    nlh_device_vecfcn fcn = nullptr;
    nlh_device_jacfcn jac = nullptr;
    void *ctx = nullptr;
    int32_t pbase = 0;
    bool user() const { return fcn != nullptr; }
    ResidualSource shifted(int32_t p0, int m, int n) const
    {
        ResidualSource r = *this;
        if (user()) r.pbase += p0;
        else { r.dA += (size_t)p0 * m * n; r.db += (size_t)p0 * m; }
        return r;
    }
};
// F(x) for the problems at stage `want` (st == nullptr: every problem): x [nprob][n] -> f [nprob][m]; part (optional):
// the per-block partial sums of squares k_dq_residual leaves (the non-exact policies' norms).
int residual_eval(nlh_handle *h, const ResidualSource &rs, int nprob, int m, int n, const double *x, double *f, double *part,
                  const LmState *st, int want);
// vfh_jac_fcn for the problems at stage `want`: the n perturbed evaluations + jac(:,j) = (f_j - f0) / h_j (or the user's
// jacobianfcn when use_jac and one is set).  out: column-major [nprob][n][m], or -- to_qrx -- the exact factorisation's
// working matrix; panel: scratch of nprob * m * n doubles, distinct from out (the dense-quadratic family's unfused form
// only: a user's panel lives in a chunk buffer of the handle).  fuse: dense-quadratic family only.
int residual_jacobian(nlh_handle *h, const ResidualSource &rs, int nprob, int m, int n, const double *x, const double *f0,
                      double *out, double *panel, const LmState *st, int want, bool to_qrx, bool fuse, bool use_jac, int known_cnt = -1);

void launch_sumsq_part(nlh_handle *h, int nprob, int m, int n, const double *f, double *part);   // nlh_lm.hip

// nlh_square.hip
void launch_lu_factor(nlh_handle *h, int nprob, int n, double *dA, int32_t *dipvt, int32_t *dinfo, const LmState *st = nullptr,
                      int want = -1);
void launch_house_steps(nlh_handle *h, int nprob, int rows, int ncA, int ncE, double *dA, double *dE, double *vbuf, double *wbuf,
                        double *st, const LmState *gst = nullptr, int gwant = -1);


static const int RB = 256;   // rows per block of the residual kernels

// host-side scalar helpers of the solvers' O(n) logic (the reference's operation order)

static inline double h_dot(int n, const double *a, const double *b)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s = s + a[i] * b[i];
    return s;
}

// NORM2 as the flang runtime evaluates it (processor-dependent intrinsic); the host-side
// Newton logic uses it only for stpmax and limit_search_vector.
static inline double h_norm2(int n, const double *x)
{
    double mx = 0.0, s = 0.0;
    for (int i = 0; i < n; ++i) {
        const double a = fabs(x[i]);
        if (mx == 0.0) mx = a;
        else if (a > mx) { const double t = mx / a, tsq = t * t; s = s * tsq; s = s + tsq; mx = a; }
        else if (a != 0.0) { const double t = a / mx; s = s + t * t; }
    }
    return mx * sqrt(1.0 + s);
}

// min_backtrack_search, src/nonlin_linesearch.f90:495-551
static inline double min_backtrack_search(int mode, double f0, double f, double f1, double alam, double alam1, double slope)
{
    double lam;
    if (mode == 1) {
        lam = -slope / (2.0 * (f - f0 - slope));
    } else {
        const double rhs1 = f - f0 - alam * slope;
        const double rhs2 = f1 - f0 - alam1 * slope;
        const double a = (rhs1 / (alam * alam) - rhs2 / (alam1 * alam1)) / (alam - alam1);
        const double b = (-alam1 * rhs1 / (alam * alam) + alam * rhs2 / (alam1 * alam1)) / (alam - alam1);
        if (a == 0.0) {
            lam = -slope / (2.0 * b);
        } else {
            const double disc = b * b - 3.0 * a * slope;
            if (disc < 0.0) lam = 0.5 * alam;
            else if (b <= 0.0) lam = (-b + sqrt(disc)) / (3.0 * a);
            else lam = -slope / (b + sqrt(disc));
        }
        if (lam > 0.5 * alam) lam = 0.5 * alam;
    }
    return lam;
}
void format_e10_3(double v, char out[16]);
void nlh_bfgs_init_device(int lds_max);
void nlh_cls_init_device(int lds_max);
void nlh_poly_init_device(int lds_max);
void nlh_devfcn_init_device(int lds_max);        // nlh_devfcn.hip: the built-in family's launcher kernels keep x in LDS
// columns the built-in dense-quadratic family's kernels accept (x in LDS, lds_max of nlh_create): beyond it NLH_ARRAY_SIZE_ERROR
static const int32_t NLH_DQ_MAX_N = 20000;

static inline int factor_threads(int n) { return n >= 96 ? 1024 : 256; }
void print_status(int iter, int nfeval, int njaceval, double xnorm, double fnorm);


// Several kernels of the lock-step drivers carry the problem index in gridDim.y / .z (at most 65535): a larger batch is
// solved in slices of NLH_MAX_LOCKSTEP problems, one after the other (independent problems: the same bits).
static const int32_t NLH_MAX_LOCKSTEP = 65535;
int lockstep_slices(int32_t nprob, const std::function<int(int32_t, int32_t)> &run);         // run(first, count)

static const int QN_MAX_N = 8192;      // k_qn_retri / k_bf_chol_*: 8 columns per thread at most (4 up to n = 4096)
// eight columns per thread: beyond 4096 columns -- or, NLH_QN_FORCE_NC8 (tests), wherever the four-column instance would run
static inline bool qn_nc8(int n)
{
    static const bool force = [] { const char *e = getenv("NLH_QN_FORCE_NC8"); return e && atoi(e) != 0; }();
    return n > 4096 || force;
}
static const int QN_LDS_ROWS = 18000;  // up to here k_qn_house_dot keeps the reflector (rows doubles) in LDS; beyond: in global memory
