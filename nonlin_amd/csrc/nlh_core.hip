// nlh_core.hip -- the handle, options, kernel timing, the synthetic generator, the launches of the residual
// family every solver shares (vecfcn of the device model, the n perturbed evaluations of vfh_jac_fcn,
// src/nonlin_multi_eqn_mult_var.f90:198-277), worker handles for host-loop batches, print_status
// (src/nonlin_helper.f90:17-33).  There is no CPU fallback: without a device every compute entry point fails.
#include "nlh_internal.h"
#include "nlh_kernels_model.h"
#include "nlh_qrx.h"

int ensure(nlh_handle *h, DevBuf &b, size_t bytes)
{
    if (bytes <= b.bytes && b.p) return 0;
    if (b.p) { hipFree(b.p); b.p = nullptr; b.bytes = 0; }
    size_t want = bytes < 256 ? 256 : bytes;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) { h->err = std::string("hipMalloc: ") + hipGetErrorString(e); return NLH_OUT_OF_MEMORY_ERROR; }
    b.bytes = want;
    bool known = false;
    for (auto *q : h->bufs) if (q == &b) known = true;
    if (!known) h->bufs.push_back(&b);
    return 0;
}

int ensure_staging(nlh_handle *h, size_t bytes)
{
    if (bytes <= h->staging_bytes) return 0;
    if (h->staging) hipHostFree(h->staging);
    h->staging = nullptr; h->staging_bytes = 0;
    if (hipHostMalloc(&h->staging, bytes, hipHostMallocDefault) != hipSuccess) { h->err = "hipHostMalloc (staging)"; return NLH_OUT_OF_MEMORY_ERROR; }
    h->staging_bytes = bytes;
    return 0;
}

int ensure_pinned(nlh_handle *h, size_t bytes)
{
    if (bytes <= h->pinned_bytes) return 0;
    if (h->pinned) hipHostFree(h->pinned);
    h->pinned = nullptr; h->pinned_bytes = 0;
    hipError_t e = hipHostMalloc(&h->pinned, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { h->err = std::string("hipHostMalloc: ") + hipGetErrorString(e); return NLH_OUT_OF_MEMORY_ERROR; }
    h->pinned_bytes = bytes;
    return 0;
}

static const char *k_names[NLH_K_COUNT] = {
    "k_dq_residual", "k_dq_panel", "k_fd_jacobian", "k_gram_mfma", "k_gram_reduce", "k_jtf",
    "k_chol_factor", "k_lmpar", "k_qr_factor", "k_lm_update", "k_lu_factor", "k_dq_jacobian", "k_qrx_pass",
    "k_qrx_pivot"};

void timing_flush(nlh_handle *h)
{
    if (h->pending.empty()) return;
    hipStreamSynchronize(h->stream);
    for (auto &pr : h->pending) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, pr.a, pr.b) == hipSuccess) h->ms[pr.kid] += (double)t;
        h->launches[pr.kid] += 1;
        if (pr.kid == h->sample_kid) h->samples.push_back(t);
        h->pool.push_back(pr.a);
        h->pool.push_back(pr.b);
    }
    h->pending.clear();
}

hipEvent_t ev_get(nlh_handle *h)
{
    if (!h->pool.empty()) { hipEvent_t e = h->pool.back(); h->pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
void nlh_default_options(nlh_options *o)
{
    o->max_evals = 100;          // src/nonlin_multi_eqn_mult_var.f90:69
    o->ftol = 1.0e-8;            // :71
    o->xtol = 1.0e-12;           // :73
    o->gtol = 1.0e-12;           // :75
    o->print_status = 0;         // :77
    o->factor = 100.0;           // src/nonlin_least_squares.f90:25
    o->use_line_search = 1;      // src/nonlin_solve.f90:30
    o->ls_max_evals = 100;       // src/nonlin_linesearch.f90:35
    o->ls_alpha = 1.0e-4;        // :38
    o->ls_factor = 0.1;          // :46
    o->factor_policy = NLH_FACTOR_EXACT;   // the parity-carrying policy; AUTO / QR are explicit opt-ins
    o->ne_pivot_tol = 1.0e-4;
    o->fuse_fd = 1;
    o->sub_batches = 0;
}

int nlh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *nlh_version(void) { return "nonlin_hip 0.1 (gfx950)"; }

int nlh_create(nlh_handle **out, int32_t device, void *hip_stream)
{
    if (!out) return NLH_ERR_BAD_HANDLE;
    *out = nullptr;
    if (nlh_device_count() <= 0) return NLH_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return NLH_ERR_NO_DEVICE;
    nlh_handle *h = new nlh_handle();
    h->device = device;
    h->stream = (hipStream_t)hip_stream;     // NULL = the device's default (null) stream
    // allow the single-workgroup kernels their full dynamic LDS (n-vectors live there)
    const int lds_max = 160 * 1024 - 2048;
    qrx_init_device();
    nlh_lm_init_device(lds_max);
    nlh_square_init_device(lds_max);
    nlh_cls_init_device(lds_max);
    nlh_bfgs_init_device(lds_max);
    nlh_poly_init_device(lds_max);
    nlh_devfcn_init_device(lds_max);
    (void)hipFuncSetAttribute((const void *)k_dq_residual<RB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    (void)hipFuncSetAttribute((const void *)k_dq_residual2<RB / 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    (void)hipGetLastError();
    *out = h;
    return 0;
}

void nlh_destroy(nlh_handle *h)
{
    if (!h) return;
    for (auto *wk : h->workers) nlh_destroy(wk);
    h->workers.clear();
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    timing_flush(h);
    for (auto e : h->pool) hipEventDestroy(e);
    for (auto *b : h->bufs) if (b->p) hipFree(b->p);
    if (h->pinned) hipHostFree(h->pinned);
    if (h->staging) hipHostFree(h->staging);
    if (h->lu_side) { hipStreamSynchronize(h->lu_side); hipStreamDestroy(h->lu_side); }
    if (h->lu_panel_done) hipEventDestroy(h->lu_panel_done);
    for (auto e : h->lu_bulk_done) if (e) hipEventDestroy(e);
    if (h->own_stream) hipStreamDestroy(h->stream);
    delete h;
}

const char *nlh_last_error(const nlh_handle *h) { return h ? h->err.c_str() : "null handle"; }

void nlh_timing_enable(nlh_handle *h, int32_t on)
{
    if (!h) return;
    // 0 = off, 1 = every kernel group, otherwise bit (k + 1) selects group NLH_K_<k> (timing costs two event records
    // per launch, so a caller that needs one kernel's durations can leave the others unbracketed)
    h->timing = on == 0 ? 0u : on == 1 ? 0xffffffffu : ((uint32_t)on >> 1);
}
void nlh_timing_reset(nlh_handle *h)
{
    if (!h) return;
    timing_flush(h);
    for (int k = 0; k < NLH_K_COUNT; ++k) { h->ms[k] = 0; h->launches[k] = 0; }
    h->samples.clear();
}
int64_t nlh_timing_samples(nlh_handle *h, int32_t kid, float *out_ms, int64_t cap)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (kid != h->sample_kid) {                 // select the group; samples accumulate from the next reset on
        timing_flush(h);
        h->sample_kid = kid;
        h->samples.clear();
        return 0;
    }
    timing_flush(h);
    const int64_t cnt = (int64_t)h->samples.size();
    for (int64_t i = 0; i < cnt && i < cap; ++i) out_ms[i] = h->samples[(size_t)i];
    return cnt;
}
int nlh_timing_get(nlh_handle *h, int32_t kid, double *total_ms, int64_t *launches)
{
    if (!h || kid < 0 || kid >= NLH_K_COUNT) return NLH_ERR_BAD_HANDLE;
    timing_flush(h);
    if (total_ms) *total_ms = h->ms[kid];
    if (launches) *launches = h->launches[kid];
    return 0;
}
const char *nlh_kernel_name(int32_t kid) { return (kid >= 0 && kid < NLH_K_COUNT) ? k_names[kid] : "?"; }




// splitmix64 as a counter-based generator; k = 0-based draw index
__device__ __forceinline__ double sm64_u(uint64_t seed, uint64_t k)
{
    uint64_t z = seed + (k + 1ULL) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (double)(z >> 11) * 0x1.0p-53;
}

__global__ void k_gen_A(int m, int n, uint64_t seed0, uint64_t stride, int square_shift, double *A, double *xtrue)
{
    const int p = blockIdx.y;
    const uint64_t seed = seed0 + (uint64_t)p * stride;
    const size_t mn = (size_t)m * n;
    const double rs = sqrt((double)n);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < mn + (size_t)n; e += (size_t)gridDim.x * blockDim.x) {
        const double u = 2.0 * sm64_u(seed, e) - 1.0;
        if (e < mn) {
            double v = u / rs;
            const size_t i = e % m, j = e / m;
            if (square_shift && i == j) v = 2.0 + v;
            A[(size_t)p * mn + e] = v;
        } else {
            xtrue[(size_t)p * n + (e - mn)] = u;
        }
    }
}

__global__ void k_gen_bx(int m, int n, uint64_t seed0, uint64_t stride, double sigma, double spread, double *b,
                         const double *xtrue, double *x0)
{
    const int p = blockIdx.y;
    const uint64_t seed = seed0 + (uint64_t)p * stride;
    const size_t base = (size_t)m * n + (size_t)n;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)m + n; e += (size_t)gridDim.x * blockDim.x) {
        const double u = 2.0 * sm64_u(seed, base + e) - 1.0;
        if (e < (size_t)m) b[(size_t)p * m + e] = b[(size_t)p * m + e] + sigma * u;
        else x0[(size_t)p * n + (e - m)] = xtrue[(size_t)p * n + (e - m)] + spread * u;
    }
}

// ---------------------------------------------------------------------------
// launch wrappers (each is one timed kernel family)
// ---------------------------------------------------------------------------

void launch_dq_residual(nlh_handle *h, int nprob, int m, int n, const double *A, const double *b,
                               double gamma, const double *x, double *f, double *part,
                               const LmState *st, int want)
{
    Timed t(h, NLH_K_DQ_RESIDUAL);
    size_t sh = sizeof(double) * (size_t)(n + 32);
    const bool vec2 = (m % 2 == 0) && ((((uintptr_t)A | (uintptr_t)b | (uintptr_t)f) & 15) == 0);
    const int nblk = (m + RB - 1) / RB;
    // the problem index rides in gridDim.y (65535 at most): more problems than that go in slices (the lock-step drivers
    // slice their batches themselves; the stage-level and model entry points come here with whatever the caller has)
    for (int p0 = 0; p0 < nprob; p0 += NLH_MAX_LOCKSTEP) {
        const int cnt = std::min<int>(NLH_MAX_LOCKSTEP, nprob - p0);
        dim3 grid(nblk, cnt);
        const double *Ap = A + (size_t)p0 * m * n, *bp = b + (size_t)p0 * m, *xp = x + (size_t)p0 * n;
        double *fp = f + (size_t)p0 * m, *pp = part ? part + (size_t)p0 * nblk * 2 : nullptr;
        const LmState *sp = st ? st + p0 : nullptr;
        if (vec2)       // two rows per thread, 16-byte accesses; a block covers the same RB rows
            hipLaunchKernelGGL(k_dq_residual2<RB / 2>, grid, dim3(RB / 2), sh, h->stream, m, n, Ap, bp, gamma, xp, fp, pp, sp, want);
        else
            hipLaunchKernelGGL(k_dq_residual<RB>, grid, dim3(RB), sh, h->stream, m, n, Ap, bp, gamma, xp, fp, pp, sp, want);
    }
}

void launch_dq_panel(nlh_handle *h, int nprob, int m, int n, const double *A, const double *b,
                            double gamma, const double *x, double *P, const LmState *st, int want,
                            const double *f0_fused, bool to_qrx)
{
    Timed t(h, NLH_K_DQ_PANEL);
    // 32 columns per thread: A is re-read from L2 n/32 times (the kernel is L2->CU bound at 16) and the
    // register budget still leaves 5 waves per SIMD; measured 2.49 ms (16) / 1.97 (32) / 1.95 (48) per
    // 256 x 4096 x 256 launch.  f0_fused != null: the epilogue writes the Jacobian column instead of the residual;
    // to_qrx: in the layout of the exact factorisation's working matrix (P is then that matrix).
    constexpr int JT = 32;
    dim3 grid((m + RB - 1) / RB, (n + JT - 1) / JT, nprob);
    size_t sh = sizeof(double) * (size_t)n;
    const int tld = to_qrx ? qrx_ld(n) : 0, tcoff = to_qrx ? qrx_ld(n) - (n + 1) : 0;
    const size_t tst = to_qrx ? qrx_matrix_stride(m, n) : 0;
    if (f0_fused)
        hipLaunchKernelGGL((k_dq_panel<RB, JT, true>), grid, dim3(RB), sh, h->stream, m, n, A, b, gamma, x, P, f0_fused, st, want,
                           tld, tcoff, tst);
    else
        hipLaunchKernelGGL((k_dq_panel<RB, JT, false>), grid, dim3(RB), sh, h->stream, m, n, A, b, gamma, x, P,
                           (const double *)nullptr, st, want, 0, 0, (size_t)0);
}

void launch_fd(nlh_handle *h, int nprob, int m, int n, const double *P, const double *f0,
                      const double *x, double *J, const LmState *st, int want)
{
    Timed t(h, NLH_K_FD_JACOBIAN);
    constexpr int CJ = 8;
    const bool vec2 = (m % 2 == 0) && ((((uintptr_t)P | (uintptr_t)J | (uintptr_t)f0) & 15) == 0);
    if (vec2) {
        dim3 grid((m / 2 + RB - 1) / RB, (n + CJ - 1) / CJ, nprob);
        hipLaunchKernelGGL((k_fd_jacobian<RB, CJ, true>), grid, dim3(RB), 0, h->stream, m, n, P, f0, x, J, st, want);
    } else {
        dim3 grid((m + RB - 1) / RB, (n + CJ - 1) / CJ, nprob);
        hipLaunchKernelGGL((k_fd_jacobian<RB, CJ, false>), grid, dim3(RB), 0, h->stream, m, n, P, f0, x, J, st, want);
    }
}

// Fortran edit descriptor E10.3 (src/nonlin_helper.f90:32): three significant digits as 0.dddE+ee, right-justified in
// ten columns; a three-digit exponent drops the letter (0.123+100), as the standard prescribes.
void format_e10_3(double v, char out[16])
{
    char body[16];
    if (std::isnan(v)) { snprintf(out, 16, "%10s", "NaN"); return; }
    if (std::isinf(v)) { snprintf(out, 16, "%10s", v < 0 ? "-Inf" : "Inf"); return; }
    char sci[32];
    snprintf(sci, sizeof sci, "%.2e", fabs(v));                 // d.dde+XX, correctly rounded to 3 digits
    int ex = atoi(sci + 5);
    if (v != 0.0) ex += 1;                                      // d.dd x 10^X = 0.ddd x 10^(X+1)
    const char sign = std::signbit(v) && v != 0.0 ? '-' : ' ';
    if (abs(ex) < 100) snprintf(body, sizeof body, "%c0.%c%c%cE%c%02d", sign, sci[0], sci[2], sci[3], ex < 0 ? '-' : '+', abs(ex));
    else snprintf(body, sizeof body, "%c0.%c%c%c%c%03d", sign, sci[0], sci[2], sci[3], ex < 0 ? '-' : '+', abs(ex));
    snprintf(out, 16, "%10s", body);
}

// print_status, src/nonlin_helper.f90:17-33: `print *, ""` (a blank), then A,I0 / A,E10.3 lines.
void print_status(int iter, int nfeval, int njaceval, double xnorm, double fnorm)
{
    char a[16], b[16];
    format_e10_3(xnorm, a);
    format_e10_3(fnorm, b);
    printf(" \nIteration: %d\nFunction Evaluations: %d\n", iter, nfeval);
    if (njaceval > 0) printf("Jacobian Evaluations: %d\n", njaceval);
    printf("Change in Variable: %s\nResidual: %s\n", a, b);
    fflush(stdout);
}

int nlh_format_status(int32_t iter, int32_t nfeval, int32_t njaceval, double xnorm, double fnorm, char *buf,
                                 int32_t len)
{
    char a[16], b[16], jl[48] = "";
    format_e10_3(xnorm, a);
    format_e10_3(fnorm, b);
    if (njaceval > 0) snprintf(jl, sizeof jl, "Jacobian Evaluations: %d\n", njaceval);
    return snprintf(buf, len > 0 ? (size_t)len : 0, " \nIteration: %d\nFunction Evaluations: %d\n%sChange in Variable: %s\nResidual: %s\n",
                    iter, nfeval, jl, a, b);
}
int lockstep_slices(int32_t nprob, const std::function<int(int32_t, int32_t)> &run)         // run(first, count)
{
    for (int32_t p0 = 0; p0 < nprob; p0 += NLH_MAX_LOCKSTEP) {
        const int rc = run(p0, std::min<int32_t>(NLH_MAX_LOCKSTEP, nprob - p0));
        if (rc) return rc;
    }
    return 0;
}


// Host-loop solvers (Newton, quasi-Newton, constrained least squares, bfgs) over a batch of independent problems:
// the problems are dealt to a few host threads, each with a private handle (own HIP stream and workspace), so the
// latency-bound kernels of different problems overlap on the device.  NLH_WORKERS sets the thread count (default 8).
// Private handles (own non-blocking stream + workspace) for work the caller's handle deals out to host threads.
int ensure_workers(nlh_handle *h, int T)
{
    while ((int)h->workers.size() < T) {
        nlh_handle *wk = new nlh_handle();
        wk->device = h->device;
        if (hipStreamCreateWithFlags(&wk->stream, hipStreamNonBlocking) != hipSuccess) { delete wk; h->err = "hipStreamCreate"; return NLH_ERR_HIP; }
        wk->own_stream = true;
        h->workers.push_back(wk);
    }
    return 0;
}

int run_problems(nlh_handle *h, int nprob, const std::function<int(nlh_handle *, int)> &solve_one)
{
    int T = 8;
    if (const char *e = getenv("NLH_WORKERS")) T = atoi(e);
    T = std::max(1, std::min(T, nprob));
    if (T == 1) {
        for (int p = 0; p < nprob; ++p) {
            const int rc = solve_one(h, p);
            if (rc != 0) return rc;
        }
        return 0;
    }
    { const int rcw = ensure_workers(h, T); if (rcw) return rcw; }
    HIPCHK(h, hipStreamSynchronize(h->stream));                 // inputs written on the caller's stream are complete
    std::atomic<int> next(0), err(0);
    std::vector<std::thread> pool;
    for (int t = 0; t < T; ++t)
        pool.emplace_back([&, t]() {
            nlh_handle *wk = h->workers[t];
            if (hipSetDevice(wk->device) != hipSuccess) { err = NLH_ERR_HIP; return; }
            for (;;) {
                const int p = next.fetch_add(1);
                if (p >= nprob || err.load() != 0) break;
                const int rc = solve_one(wk, p);
                if (rc != 0) { err = rc; break; }
            }
            hipStreamSynchronize(wk->stream);
        });
    for (auto &th : pool) th.join();
    if (err.load() != 0) {
        for (auto *wk : h->workers) if (!wk->err.empty()) { h->err = wk->err; break; }
        return err.load();
    }
    return 0;
}

// ===========================================================================
// Synthetic inputs + stage-level entry points
// ===========================================================================
int nlh_dq_generate(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, uint64_t seed0, uint64_t seed_stride, double gamma,
                    double sigma, double spread, int32_t square_shift, double *dA, double *db,
                    double *dxtrue, double *dx0)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t mn = (size_t)m * n;
    unsigned gx = (unsigned)((mn + n + 255) / 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_gen_A, dim3(gx, nprob), dim3(256), 0, h->stream, m, n, seed0, seed_stride, square_shift, dA, dxtrue);
    HIPCHK(h, hipMemsetAsync(db, 0, sizeof(double) * (size_t)nprob * m, h->stream));
    launch_dq_residual(h, nprob, m, n, dA, db, gamma, dxtrue, db, nullptr, nullptr, -1);   // b = model(x_true) - 0
    unsigned gb = (unsigned)(((size_t)m + n + 255) / 256);
    hipLaunchKernelGGL(k_gen_bx, dim3(gb, nprob), dim3(256), 0, h->stream, m, n, seed0, seed_stride, sigma, spread, db, dxtrue, dx0);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int nlh_dq_residual(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, const double *dA, const double *db,
                    double gamma, const double *dx, double *df)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    HIPCHK(h, hipSetDevice(h->device));
    launch_dq_residual(h, nprob, m, n, dA, db, gamma, dx, df, nullptr, nullptr, -1);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int nlh_dq_fd_panel(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, const double *dA, const double *db,
                    double gamma, const double *dx, double *dP)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (m < 1 || n < 1) return NLH_INVALID_INPUT_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    launch_dq_panel(h, nprob, m, n, dA, db, gamma, dx, dP, nullptr, -1);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int nlh_fd_jacobian_panel(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, const double *dP,
                          const double *df0, const double *dx, double *dJ)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    HIPCHK(h, hipSetDevice(h->device));
    launch_fd(h, nprob, m, n, dP, df0, dx, dJ, nullptr, -1);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int nlh_dq_jacobian(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, const double *dA, double gamma,
                    const double *dx, double *dJ)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    HIPCHK(h, hipSetDevice(h->device));
    {
        Timed t(h, NLH_K_DQ_JACOBIAN);
        hipLaunchKernelGGL(k_dq_jacobian<RB>, dim3((m + RB - 1) / RB, nprob), dim3(RB), sizeof(double) * n, h->stream,
                           m, n, dA, gamma, dx, dJ, (const LmState *)nullptr, -1);
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}
