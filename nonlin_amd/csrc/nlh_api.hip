// nlh_api.hip -- C ABI (include/nonlin_hip.h) and host drivers of libnonlin_hip.so.
//
// Host side of the path: lss_solve (src/nonlin_least_squares.f90:118-391) as a batched,
// device-resident state machine; ns_solve (src/nonlin_solve.f90:452-638) and ls_search_mimo
// (src/nonlin_linesearch.f90:152-326) as host loops around device kernels; vfh_jac_fcn
// (src/nonlin_multi_eqn_mult_var.f90:198-277) for host callbacks.
// There is no CPU fallback: without a device every compute entry point fails.
#include "../../include/nonlin_hip.h"

#include <hip/hip_runtime.h>

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
#include "nlh_kernels_model.h"
#include "nlh_kernels_gram.h"
#include "nlh_kernels_factor.h"
#include "nlh_kernels_lm.h"
#include "nlh_kernels_lu.h"
#include "nlh_kernels_newton.h"
#include "nlh_kernels_cls.h"
#include "nlh_kernels_broyden.h"
#include "nlh_kernels_bfgs.h"
#include "nlh_kernels_bfgs_batch.h"
#include "nlh_kernels_exact.h"
#include "nlh_qrx.h"

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
           qnQ, qnR, qnV, bfB, bfR, bfV, qxV;
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
    DevBuf cholmc;                     // side buffer of the multi-CU Cholesky (solved panels, bad-pivot flags)
    void *staging = nullptr;           // pinned staging of a device-set share's rows of the caller's host arrays
    size_t staging_bytes = 0;
    bool qrx_open_on = false; hipEvent_t qrx_a{}, qrx_b{}; int qrx_kid = 0;   // open bracket of a nlh_qrx.hip launch
    std::vector<nlh_handle *> workers;   // private handles (own stream + workspace) for concurrent host-loop solves
};

#define HIPCHK(h, call)                                                                 \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);               \
            return NLH_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

static int ensure(nlh_handle *h, DevBuf &b, size_t bytes)
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

static int ensure_staging(nlh_handle *h, size_t bytes)
{
    if (bytes <= h->staging_bytes) return 0;
    if (h->staging) hipHostFree(h->staging);
    h->staging = nullptr; h->staging_bytes = 0;
    if (hipHostMalloc(&h->staging, bytes, hipHostMallocDefault) != hipSuccess) { h->err = "hipHostMalloc (staging)"; return NLH_OUT_OF_MEMORY_ERROR; }
    h->staging_bytes = bytes;
    return 0;
}

static int ensure_pinned(nlh_handle *h, size_t bytes)
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

static void timing_flush(nlh_handle *h)
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

static hipEvent_t ev_get(nlh_handle *h)
{
    if (!h->pool.empty()) { hipEvent_t e = h->pool.back(); h->pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

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

// Brackets for the launches of nlh_qrx.hip (another translation unit): which = 0 pivot kernel, 1 trailing pass, 2 rest.
static void qrx_time_begin(nlh_handle *h, int which, hipStream_t s)
{
    const int kid = which == 1 ? NLH_K_QRX_PASS : which == 0 ? NLH_K_QRX_PIVOT : NLH_K_QR;
    h->qrx_open_on = (h->timing >> kid) & 1u;
    if (h->qrx_open_on) { h->qrx_a = ev_get(h); h->qrx_b = ev_get(h); hipEventRecord(h->qrx_a, s); }
    h->qrx_kid = kid;
}
static void qrx_time_end(nlh_handle *h, int, hipStream_t s)
{
    if (!h->qrx_open_on) return;
    hipEventRecord(h->qrx_b, s);
    h->pending.push_back({h->qrx_a, h->qrx_b, h->qrx_kid});
    if (h->pending.size() > 65536) timing_flush(h);
}

static int ensure_workers(nlh_handle *h, int T);

__global__ void k_lmpar_standalone(int n, double *Rall, int ldr, const int32_t *ipvt_all, const double *diag_all,
                                   const double *qtf_all, const double *delta_all, const double *tailsq_all,
                                   double *par_all, double *x_all, double *sdiag_all, double *Wall);

extern "C" {

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
    hipFuncSetAttribute((const void *)k_gram_tri<16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_gram_tri<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_gram_512, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_chol_factor, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_chol_nopiv<16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_chol_mc_step<16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qr_factor, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_lmpar<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_lmpar<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_lmpar_standalone, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_lu_solve, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_lu_panel_lds, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_house_dot<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_house_dot2, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_house_fused<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_house_fused<16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_bf_solve_upper_t, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_bf_chol_update<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_bf_chol_update<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_bf_downdate_apply, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_matvec_cm, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_rot_q, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_hess_r, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_retri<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_retri<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_solve_upper, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_resid, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
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

}  // extern "C"

// ---------------------------------------------------------------------------
// small helper kernels of the drivers
// ---------------------------------------------------------------------------
__global__ void k_stage_advance(int nprob, LmState *st, int from, int to, int njac_inc)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nprob) return;
    if (st[p].stage == from) { st[p].stage = to; st[p].njac += njac_inc; }
}

// :211-218: fnorm of the starting residual, counters.
__global__ void k_lm_init(int nprob, int nblk, const double *part, LmState *st, int first_stage)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nprob) return;
    double sq = 0.0;
    for (int k = 0; k < nblk; ++k) sq = sq + part[((size_t)p * nblk + k) * 2];
    LmState s;
    memset(&s, 0, sizeof s);
    s.fnorm = sqrt(sq);
    s.neval = 1;
    s.iter = 1;
    s.par = 0.0;
    s.stage = first_stage;
    st[p] = s;
}

__global__ void k_count_active(int nprob, const LmState *st, int *out)
{
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int c = 0;
    for (int p = threadIdx.x; p < nprob; p += blockDim.x) c += (st[p].stage != ST_DONE);
    atomicAdd(&cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) *out = cnt;
}

// partial sums of squares of a device vector, same block structure as k_dq_residual
template <int BS>
__global__ void k_sumsq_part(int m, int n, const double *__restrict__ f, double *__restrict__ part)
{
    __shared__ double red[16];
    const int p = blockIdx.y;
    const int i = blockIdx.x * BS + threadIdx.x;
    const double v = (i < m) ? f[(size_t)p * m + i] : 0.0;
    const double sq = v * v;
    const double tq = (i >= n) ? sq : 0.0;
    const double s = block_reduce_sum(sq, red);
    const double t = block_reduce_sum(tq, red);
    if (threadIdx.x == 0) {
        part[((size_t)p * gridDim.x + blockIdx.x) * 2 + 0] = s;
        part[((size_t)p * gridDim.x + blockIdx.x) * 2 + 1] = t;
    }
}

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
static const int RB = 256;   // rows per block of the residual kernels

static void launch_dq_residual(nlh_handle *h, int nprob, int m, int n, const double *A, const double *b,
                               double gamma, const double *x, double *f, double *part,
                               const LmState *st, int want)
{
    Timed t(h, NLH_K_DQ_RESIDUAL);
    dim3 grid((m + RB - 1) / RB, nprob);
    size_t sh = sizeof(double) * (size_t)(n + 32);
    const bool vec2 = (m % 2 == 0) && ((((uintptr_t)A | (uintptr_t)b | (uintptr_t)f) & 15) == 0);
    if (vec2)       // two rows per thread, 16-byte accesses; a block covers the same RB rows
        hipLaunchKernelGGL(k_dq_residual2<RB / 2>, grid, dim3(RB / 2), sh, h->stream, m, n, A, b, gamma, x, f, part, st, want);
    else
        hipLaunchKernelGGL(k_dq_residual<RB>, grid, dim3(RB), sh, h->stream, m, n, A, b, gamma, x, f, part, st, want);
}

static void launch_dq_panel(nlh_handle *h, int nprob, int m, int n, const double *A, const double *b,
                            double gamma, const double *x, double *P, const LmState *st, int want,
                            const double *f0_fused = nullptr, bool to_qrx = false)
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

static void launch_fd(nlh_handle *h, int nprob, int m, int n, const double *P, const double *f0,
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

// K-splits of the Gram contraction.  The split count and the kernel are functions of the problem SHAPE only, never of how
// many problems share the launch: G = sum over splits (in split order) of a row-ascending accumulation, so a problem's
// bits do not depend on its batch, the round it is active in, the sub-batch or the rank it was dealt to.
static int gram_splits(int m)
{
    const int s = (m + 1023) / 1024;                  // 1024 rows per split (4096 rows per split was measured: no gain for
                                                      // 512 x 4096x256, and one problem alone 3.3 -> 7.0 ms per solve)
    return s < 1 ? 1 : s;
}

static bool gram512_on() { const char *e = getenv("NLH_GRAM512"); return !e || atoi(e) != 0; }   // (0: k_gram_mfma for 256 < n <= 512, for comparison)

static int launch_gram(nlh_handle *h, int nprob, int m, int n, const double *J, const double *f,
                       double *G, double *g, const LmState *st, int want)
{
    const int nb = (n + GRAM_BT - 1) / GRAM_BT;
    const int nblk = nb * (nb + 1) / 2;
    const int ns = gram_splits(m);
    int rps = (m + ns - 1) / ns;
    rps = ((rps + GRAM_KT - 1) / GRAM_KT) * GRAM_KT;
    int rc = ensure(h, h->Gpart, sizeof(double) * ((size_t)nprob * ns * n * n + (size_t)nprob * ns * n));
    if (rc) return rc;
    double *Gp = (double *)h->Gpart.p;
    double *gp = Gp + (size_t)nprob * ns * n * n;
    {
        Timed t(h, NLH_K_GRAM);
        const long items = (long)ns * nprob;
        const bool tri16 = n > 224 && n <= 256;
        const bool tri8 = n > 96 && n <= 128;
        if (tri16 || tri8) {
            // whole lower triangle per workgroup, J staged once
            const int nt = tri16 ? 16 : 8;
            const size_t sh = sizeof(double) * (size_t)(16 * nt * GRAM_LD + GRAM_KT + 64 * nt);
            const bool direct = ns == 1;
            if (tri16)
                hipLaunchKernelGGL(k_gram_tri<16>, dim3((unsigned)items), dim3(512), sh, h->stream, m, n, rps, J, Gp,
                                   g ? f : (const double *)nullptr, gp, st, want, ns, direct ? G : (double *)nullptr,
                                   direct ? g : (double *)nullptr);
            else
                hipLaunchKernelGGL(k_gram_tri<8>, dim3((unsigned)items), dim3(256), sh, h->stream, m, n, rps, J, Gp,
                                   g ? f : (const double *)nullptr, gp, st, want, ns, direct ? G : (double *)nullptr,
                                   direct ? g : (double *)nullptr);
            if (direct) return 0;          // one split: G and g are final, nothing to reduce
        } else if (n > 256 && n <= 512 && gram512_on()) {
            // four workgroups per item: the two diagonal 256-column blocks and the two halves of the square between them
            const long groups = (items + 7) / 8;
            const size_t sh = sizeof(double) * (size_t)(384 * GRAM_LD + GRAM_KT + 1024);
            hipLaunchKernelGGL(k_gram_512, dim3((unsigned)(groups * 32)), dim3(512), sh, h->stream, m, n, rps, J, Gp,
                               g ? f : (const double *)nullptr, gp, st, want, ns, nprob);
        } else {
            const long groups = (items + 7) / 8;
            hipLaunchKernelGGL(k_gram_mfma, dim3((unsigned)(groups * 8 * nblk)), dim3(256), 0, h->stream, m, n, rps, J, Gp,
                               g ? f : (const double *)nullptr, gp, st, want, nblk, ns, nprob);
        }
    }
    {
        Timed t(h, NLH_K_GRAM_REDUCE);
        dim3 grid((unsigned)(((size_t)n * n + 255) / 256), nprob);
        hipLaunchKernelGGL(k_gram_reduce, grid, dim3(256), 0, h->stream, n, ns, (const double *)Gp, G,
                           (const double *)gp, g, st, want);
    }
    return 0;
}

static int factor_threads(int n) { return n >= 96 ? 1024 : 256; }

// lu_factor: unblocked single-workgroup kernel for small n, blocked multi-kernel path otherwise.
static void launch_lu_factor(nlh_handle *h, int nprob, int n, double *dA, int32_t *dipvt, int32_t *dinfo,
                             const LmState *st = nullptr, int want = -1)
{
    Timed t(h, NLH_K_LU);
    if (n < 128) {
        hipLaunchKernelGGL(k_lu_factor, dim3(nprob), dim3(n >= 96 ? 1024 : 256), 0, h->stream, n, dA, dipvt, dinfo, st, want);
        return;
    }
    if (dinfo) hipMemsetAsync(dinfo, 0, sizeof(int32_t) * (size_t)nprob, h->stream);
    // panels of 16 columns factored in registers (thread per row) while n - jb <= 1024 rows, 32-column panels in global memory before
    for (int jb = 0; jb < n;) {
        const bool lds = (n - jb) <= LU_PROWS;
        const int pw = lds ? LU_PNB : LU_NB;
        const int nb = (n - jb < pw) ? (n - jb) : pw;
        if (lds)
            // one thread per panel row: waves without rows would still run the step's instruction stream and barriers
            hipLaunchKernelGGL(k_lu_panel_lds, dim3(nprob), dim3(std::min(1024, (n - jb + 63) & ~63)), 0, h->stream, n, dA, dipvt, dinfo, jb, nb,
                               st, want);
        else
            hipLaunchKernelGGL(k_lu_panel, dim3(nprob), dim3(1024), 0, h->stream, n, dA, dipvt, dinfo, jb, nb, st, want);
        if (n - nb > 0)
            hipLaunchKernelGGL(k_lu_swap_trsm, dim3((n - nb + 255) / 256, nprob), dim3(256), 0, h->stream, n, dA,
                               (const int32_t *)dipvt, jb, nb, st, want);
        const int nt = n - jb - nb;
        if (nt > 0) {
            hipLaunchKernelGGL(k_lu_gemm, dim3((nt + 63) / 64, (nt + 63) / 64, nprob), dim3(256), 0, h->stream, n, dA, jb, nb, st, want);
        }
        jb += nb;
    }
}

struct LmWs {
    double *J, *P, *wa4, *scratch, *G, *g, *part, *W2, *R;
    LmVecs v;
    LmState *st;
    int32_t *info;
    int nblk;
};

static int lm_workspace(nlh_handle *h, int nprob, int m, int n, LmWs &w, bool need_panel, bool need_J = true)
{
    int rc;
    const size_t mn = (size_t)nprob * m * n, pm = (size_t)nprob * m, pn = (size_t)nprob * n;
    // need_J = false: the exact policy with the fused FD epilogue writes the Jacobian straight into the factorisation's
    // working matrix (the panel buffer) and nothing reads a column-major J: 17 GB less at 2048 x 4096x256
    if (need_J && (rc = ensure(h, h->J, sizeof(double) * mn))) return rc;
    // the panel doubles as the exact factorisation's row-major working matrix (nlh_qrx.hip) and as lmsolve's scratch
    if (need_panel && (rc = ensure(h, h->P, sizeof(double) * std::max(mn + pm + (size_t)512 * (n + 1),
                                                                      qrx_matrix_doubles(nprob, m, n))))) return rc;
    if ((rc = ensure(h, h->wa4, sizeof(double) * pm))) return rc;
    if ((rc = ensure(h, h->scratch, sizeof(double) * pm))) return rc;
    if ((rc = ensure(h, h->G, sizeof(double) * (size_t)nprob * n * n))) return rc;
    if ((rc = ensure(h, h->W2, sizeof(double) * (size_t)nprob * n * n))) return rc;
    if ((rc = ensure(h, h->R, sizeof(double) * (size_t)nprob * n * n))) return rc;
    if ((rc = ensure(h, h->vecs, sizeof(double) * pn * 10))) return rc;
    if ((rc = ensure(h, h->ipvt, sizeof(int32_t) * pn))) return rc;
    if ((rc = ensure(h, h->gvec, sizeof(double) * pn))) return rc;
    w.nblk = (m + RB - 1) / RB;
    if ((rc = ensure(h, h->part, sizeof(double) * (size_t)nprob * w.nblk * 2))) return rc;
    if ((rc = ensure(h, h->state, sizeof(LmState) * (size_t)nprob))) return rc;
    if ((rc = ensure(h, h->info, sizeof(int32_t) * (size_t)(nprob + 16)))) return rc;
    w.J = (double *)h->J.p; w.P = (double *)h->P.p; w.wa4 = (double *)h->wa4.p;
    w.scratch = (double *)h->scratch.p; w.G = (double *)h->G.p; w.g = (double *)h->gvec.p;
    w.part = (double *)h->part.p; w.st = (LmState *)h->state.p; w.info = (int32_t *)h->info.p;
    w.W2 = (double *)h->W2.p;
    w.R = (double *)h->R.p;
    double *vb = (double *)h->vecs.p;
    w.v.diag = vb; w.v.diag_prev = vb + pn; w.v.qtf = vb + 2 * pn; w.v.acnorm = vb + 3 * pn;
    w.v.rdiag = vb + 4 * pn; w.v.g = vb + 5 * pn; w.v.wa1 = vb + 6 * pn; w.v.wa2 = vb + 7 * pn;
    w.v.wa3 = vb + 8 * pn; w.v.sdiag = vb + 9 * pn;
    w.v.ipvt = (int32_t *)h->ipvt.p;
    return 0;
}

// One pass over the factorisation + lmpar stages for every problem whose Jacobian is in
// w.J (stage ST_HAVE_JAC or ST_NEED_QR) or whose factors are ready (inner-loop repeat).
static int lm_factor_and_step(nlh_handle *h, const nlh_options *o, int nprob, int m, int n, LmWs &w,
                              double *dx, const double *dfvec, int nact = -1, bool jac_in_qrx_layout = false)
{
    const int ft = factor_threads(n);
    const size_t shl = sizeof(double) * (size_t)(6 * n + 72);
    if (o->factor_policy == NLH_FACTOR_EXACT) {
        // reference operation order: exact lmfactor + Q^T f (streaming form, the batch advances through the
        // Householder steps in lock step: nlh_qrx.hip), exact lmpar
        {
            int rc;
            if ((rc = ensure(h, h->qxV, qrx_workspace_bytes(nprob, m, n)))) return rc;
            QrxTimer tm{h, [](void *c, int which, hipStream_t s) { qrx_time_begin((nlh_handle *)c, which, s); },
                        [](void *c, int which, hipStream_t s) { qrx_time_end((nlh_handle *)c, which, s); }};
            qrx_factor(h->stream, nprob, m, n, jac_in_qrx_layout ? (const double *)nullptr : w.J, w.P, dfvec, w.R, w.v, w.wa4,
                       w.scratch, dx, w.st, o->factor, o->gtol, h->qxV.p, &tm, nact);
        }
        {
            Timed t(h, NLH_K_LMPAR);
            hipLaunchKernelGGL(k_lmpar<true>, dim3(nprob), dim3(ft), shl + sizeof(double) * (3 * NLH_NCH + 8),
                               h->stream, m, n, w.R, w.v, dx, w.wa4, w.P, w.J, w.W2, w.st, (int)ST_QR_READY);
        }
        return 0;
    }
    int rc = launch_gram(h, nprob, m, n, w.J, dfvec, w.G, w.g, w.st, ST_HAVE_JAC);
    if (rc) return rc;
    constexpr int NB = 16;
    {   // fast path: blocked Cholesky in natural order, G -> R
        Timed t(h, NLH_K_CHOL);
        size_t sh = sizeof(double) * ((size_t)NB * n + NB * NB + n + NB + 64);
        // a handful of problems (BASELINE config 5's one 65536 x 512 problem): a launch per panel step over many CUs
        // instead of one workgroup per problem; same bits (nlh_kernels_factor.h)
        const char *mc_e = getenv("NLH_CHOL_MC");                   // (read per call: tests switch between the two forms)
        const int mc_max = mc_e ? atoi(mc_e) : 8;
        const int na = nact < 0 ? nprob : nact;
        if (sh <= 150 * 1024 && n >= 192 && na <= mc_max) {
            int rc2;
            if ((rc2 = ensure(h, h->cholmc, sizeof(double) * (size_t)nprob * 2 * (NB * n + NB * NB + NB) + sizeof(int32_t) * (size_t)nprob + 64))) return rc2;
            double *side = (double *)h->cholmc.p;
            double *fact = side + (size_t)nprob * 2 * NB * n;
            int32_t *bad = (int32_t *)(fact + (size_t)nprob * 2 * (NB * NB + NB));
            hipLaunchKernelGGL(k_chol_mc_begin<NB>, dim3(64, nprob), dim3(256), 0, h->stream, n, (const double *)w.G, (const double *)w.g, w.R, w.v,
                               fact, bad, (const LmState *)w.st, o->ne_pivot_tol);
            const size_t shm = sizeof(double) * ((size_t)NB * n + 2 * (NB * NB + 2 * NB));
            for (int jb = 0; jb < n; jb += NB)
                hipLaunchKernelGGL(k_chol_mc_step<NB>, dim3(CHOLMC_NWG + 2, nprob), dim3(512), shm, h->stream, n, jb, w.R, w.v, side, fact, bad,
                                   (const LmState *)w.st, o->ne_pivot_tol);
            hipLaunchKernelGGL(k_chol_mc_end<NB>, dim3(nprob), dim3(ft), sizeof(double) * (size_t)n, h->stream, n, w.R, w.v, (const double *)side, (const int32_t *)bad, dx,
                               w.st, o->factor, o->gtol);
        } else if (sh <= 150 * 1024) {
            // more problems than CUs: 512-thread workgroups, two of which fit a CU (128 VGPRs each), so that the
            // latency-bound phases of one factorisation overlap the MFMA phase of the other
            const int ct = (ft == 1024 && (nact < 0 ? nprob : nact) > 256) ? 512 : ft;
            hipLaunchKernelGGL(k_chol_nopiv<NB>, dim3(nprob), dim3(ct), sh, h->stream, n, w.G, w.g, w.R, w.v, dx, w.st,
                               o->factor, o->gtol, o->ne_pivot_tol);
        } else {    // panel does not fit LDS: go straight to the pivoted (unblocked) factorisation
            size_t sh2 = sizeof(double) * (size_t)(3 * n + 64);
            hipLaunchKernelGGL(k_chol_factor, dim3(nprob), dim3(ft), sh2, h->stream, n, w.R, w.G, w.g, w.v, dx, w.st,
                               (int32_t *)nullptr, o->factor, o->gtol, o->ne_pivot_tol, 0, (int)ST_HAVE_JAC);
        }
    }
    {
        Timed t(h, NLH_K_LMPAR);
        const int lt = (ft == 1024 && n <= 512 && (nact < 0 ? nprob : nact) > 256) ? 512 : ft;   // two workgroups per CU, as for the Cholesky
        hipLaunchKernelGGL(k_lmpar<false>, dim3(nprob), dim3(lt), shl, h->stream, m, n, w.R, w.v, dx, w.wa4, w.P,
                           w.J, w.W2, w.st, (int)ST_NE_READY);
    }
    {   // problems whose lmpar iteration needs lmfactor's pivot order (or with a weak pivot)
        Timed t(h, NLH_K_CHOL);
        size_t sh = sizeof(double) * (size_t)(3 * n + 64);
        hipLaunchKernelGGL(k_chol_factor, dim3(nprob), dim3(ft), sh, h->stream, n, w.R, w.G, w.g, w.v, dx, w.st,
                           (int32_t *)nullptr, o->factor, o->gtol, o->ne_pivot_tol, 0, (int)ST_NEED_PCHOL);
    }
    {
        Timed t(h, NLH_K_LMPAR);
        hipLaunchKernelGGL(k_lmpar<false>, dim3(nprob), dim3(ft), shl, h->stream, m, n, w.R, w.v, dx, w.wa4, w.P,
                           w.J, w.W2, w.st, (int)ST_NE_READY);
    }
    {
        Timed t(h, NLH_K_QR);
        size_t sh = sizeof(double) * (size_t)(3 * n + 64);
        hipLaunchKernelGGL(k_qr_factor, dim3(nprob), dim3(1024), sh, h->stream, m, n, w.J, dfvec, w.R, w.v,
                           w.wa4, w.scratch, dx, w.st, o->factor, o->gtol, 0);
    }
    {
        Timed t(h, NLH_K_LMPAR);
        hipLaunchKernelGGL(k_lmpar<false>, dim3(nprob), dim3(ft), shl, h->stream, m, n, w.R, w.v, dx, w.wa4, w.P,
                           w.J, w.W2, w.st, (int)ST_QR_READY);
    }
    return 0;
}

static void lm_update(nlh_handle *h, const nlh_options *o, int nprob, int m, int n, LmWs &w, double *dx, double *dfvec)
{
    Timed t(h, NLH_K_UPDATE);
    if (o->factor_policy == NLH_FACTOR_EXACT)
        hipLaunchKernelGGL(k_lm_update<true>, dim3(nprob), dim3(256), 0, h->stream, m, n, w.nblk, w.part, w.v, dx,
                           dfvec, w.wa4, w.st, o->ftol, o->xtol, (int)o->max_evals);
    else
        hipLaunchKernelGGL(k_lm_update<false>, dim3(nprob), dim3(256), 0, h->stream, m, n, w.nblk, w.part, w.v, dx,
                           dfvec, w.wa4, w.st, o->ftol, o->xtol, (int)o->max_evals);
}

static void fill_ib(const LmState &s, nlh_iteration_behavior *ib)
{
    ib->iter_count = s.iter;
    ib->fcn_count = s.neval;
    ib->jacobian_count = s.njac;
    ib->gradient_count = 0;
    ib->converge_on_fcn = s.fcnvrg;
    ib->converge_on_chng = s.xcnvrg;
    ib->converge_on_zero_diff = s.gcnvrg;
}

// Fortran edit descriptor E10.3 (src/nonlin_helper.f90:32): three significant digits as 0.dddE+ee, right-justified in
// ten columns; a three-digit exponent drops the letter (0.123+100), as the standard prescribes.
static void format_e10_3(double v, char out[16])
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
static void print_status(int iter, int nfeval, int njaceval, double xnorm, double fnorm)
{
    char a[16], b[16];
    format_e10_3(xnorm, a);
    format_e10_3(fnorm, b);
    printf(" \nIteration: %d\nFunction Evaluations: %d\n", iter, nfeval);
    if (njaceval > 0) printf("Jacobian Evaluations: %d\n", njaceval);
    printf("Change in Variable: %s\nResidual: %s\n", a, b);
    fflush(stdout);
}

extern "C" int nlh_format_status(int32_t iter, int32_t nfeval, int32_t njaceval, double xnorm, double fnorm, char *buf,
                                 int32_t len)
{
    char a[16], b[16], jl[48] = "";
    format_e10_3(xnorm, a);
    format_e10_3(fnorm, b);
    if (njaceval > 0) snprintf(jl, sizeof jl, "Jacobian Evaluations: %d\n", njaceval);
    return snprintf(buf, len > 0 ? (size_t)len : 0, " \nIteration: %d\nFunction Evaluations: %d\n%sChange in Variable: %s\nResidual: %s\n",
                    iter, nfeval, jl, a, b);
}

static int check_opts_lm(const nlh_options *o, int m, int n)
{
    if (!o) return NLH_INVALID_INPUT_ERROR;
    if (n > m) return NLH_UNDERDEFINED_PROBLEM_ERROR;          // :189
    if (n < 1 || m < 1) return NLH_INVALID_INPUT_ERROR;
    if (n > 3000) return NLH_ARRAY_SIZE_ERROR;                 // LDS-resident n-vectors
    return 0;
}

extern "C" {

// ===========================================================================
// Device-model LM, batched: lss_solve as a lock-step state machine.
// ===========================================================================
static int lm_solve_range(nlh_handle *h, const nlh_options *o, int32_t nprob, int32_t m, int32_t n,
                          const double *dA, const double *db, double gamma, double *dx, double *dfvec,
                          nlh_iteration_behavior *ib, int32_t *status)
{
    int rc;
    HIPCHK(h, hipSetDevice(h->device));
    LmWs w;
    const bool jac_in_place = o->fuse_fd && o->factor_policy == NLH_FACTOR_EXACT;
    if ((rc = lm_workspace(h, nprob, m, n, w, true, !jac_in_place))) return rc;
    if ((rc = ensure_pinned(h, sizeof(LmState) * (size_t)nprob + 64))) return rc;
    int *d_active = (int *)(w.info + nprob);
    int *h_active = (int *)h->pinned;
    LmState *h_state = (LmState *)((char *)h->pinned + 64);
    const int pb = (nprob + 255) / 256;
    const int first_stage = ST_NEED_JAC;

    // :211-213  f(x0), fnorm
    launch_dq_residual(h, nprob, m, n, dA, db, gamma, dx, dfvec, w.part, nullptr, -1);
    if (o->factor_policy == NLH_FACTOR_EXACT)
        hipLaunchKernelGGL(k_lm_init_exact, dim3(nprob), dim3(256), 0, h->stream, m, dfvec, w.st, first_stage);
    else
        hipLaunchKernelGGL(k_lm_init, dim3(pb), dim3(256), 0, h->stream, nprob, w.nblk, w.part, w.st, first_stage);

    const int max_rounds = o->max_evals + 8;
    const bool echo = o->print_status && nprob == 1;
    int last_printed_iter = -1;
    int nact = nprob;                                           // problems still iterating (from the previous round)
    for (int round = 0; round < max_rounds; ++round) {
        // outer-loop head for problems that need a Jacobian (:221): n perturbed evaluations + FD
        if (o->fuse_fd && o->factor_policy == NLH_FACTOR_EXACT) {
            // the Jacobian is only ever read by the exact factorisation: written in its working layout, no re-layout pass
            launch_dq_panel(h, nprob, m, n, dA, db, gamma, dx, w.P, w.st, ST_NEED_JAC, dfvec, true);
        } else if (o->fuse_fd) {
            launch_dq_panel(h, nprob, m, n, dA, db, gamma, dx, w.J, w.st, ST_NEED_JAC, dfvec);
        } else {
            launch_dq_panel(h, nprob, m, n, dA, db, gamma, dx, w.P, w.st, ST_NEED_JAC);
            launch_fd(h, nprob, m, n, w.P, dfvec, dx, w.J, w.st, ST_NEED_JAC);
        }
        hipLaunchKernelGGL(k_stage_advance, dim3(pb), dim3(256), 0, h->stream, nprob, w.st, (int)ST_NEED_JAC,
                           o->factor_policy != NLH_FACTOR_AUTO ? (int)ST_NEED_QR : (int)ST_HAVE_JAC, 1);
        if ((rc = lm_factor_and_step(h, o, nprob, m, n, w, dx, dfvec, nact,
                                     o->fuse_fd && o->factor_policy == NLH_FACTOR_EXACT))) return rc;
        // trial residual (:297-299)
        launch_dq_residual(h, nprob, m, n, dA, db, gamma, w.v.wa2, w.wa4, w.part, w.st, ST_TRIAL_READY);
        hipLaunchKernelGGL(k_stage_advance, dim3(pb), dim3(256), 0, h->stream, nprob, w.st, (int)ST_TRIAL_READY,
                           (int)ST_TRIAL_DONE, 0);
        lm_update(h, o, nprob, m, n, w, dx, dfvec);
        hipLaunchKernelGGL(k_count_active, dim3(1), dim3(256), 0, h->stream, nprob, w.st, d_active);
        HIPCHK(h, hipMemcpyAsync(h_active, d_active, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        if (echo) HIPCHK(h, hipMemcpyAsync(h_state, w.st, sizeof(LmState), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        // a single solve with print_status set: the reference's status block at the end of every outer iteration that
        // goes on (:372-374), printed from the state that came back with the count
        if (echo && h_state[0].stage == ST_NEED_JAC && h_state[0].iter != last_printed_iter) {
            print_status(h_state[0].iter, h_state[0].neval, h_state[0].njac, h_state[0].xnorm, h_state[0].fnorm);
            last_printed_iter = h_state[0].iter;
        }
        if (*h_active == 0) break;
        nact = *h_active;
    }
    HIPCHK(h, hipMemcpyAsync(h_state, w.st, sizeof(LmState) * (size_t)nprob, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    for (int p = 0; p < nprob; ++p) {
        if (ib) fill_ib(h_state[p], &ib[p]);
        if (status) status[p] = (h_state[p].flag != 0 || h_state[p].stage != ST_DONE) ? NLH_CONVERGENCE_ERROR : 0;  // :388-390
    }
    return 0;
}

// Several kernels of the lock-step drivers carry the problem index in gridDim.y / .z (at most 65535): a larger batch is
// solved in slices of NLH_MAX_LOCKSTEP problems, one after the other (independent problems: the same bits).
static const int32_t NLH_MAX_LOCKSTEP = 65535;
static int lockstep_slices(int32_t nprob, const std::function<int(int32_t, int32_t)> &run)         // run(first, count)
{
    for (int32_t p0 = 0; p0 < nprob; p0 += NLH_MAX_LOCKSTEP) {
        const int rc = run(p0, std::min<int32_t>(NLH_MAX_LOCKSTEP, nprob - p0));
        if (rc) return rc;
    }
    return 0;
}

// Several sub-batches in flight.  A batch is a lock-step state machine whose rounds contain latency-bound stages (pivot /
// NORM2 chains of the exact lmfactor, Cholesky, lmpar's iteration for the few problems that need it, straggler rounds,
// the status read-back): with the batch dealt to S host threads, each driving its own stream and workspace, those stages
// of one sub-batch run under the streaming kernels of the others.  Problems are independent and a problem's arithmetic
// does not depend on its neighbours, so x, fvec and all counts are the same bits for any S.
static int lm_sub_batches(const nlh_options *o, int nprob, int m, int n)
{
    int S = o->sub_batches;
    if (S <= 0)                                     // the environment only fills in for "automatic", never overrides a caller
        if (const char *e = getenv("NLH_SUB_BATCHES")) S = atoi(e);
    if (S <= 0) {                                   // auto: >= 128 problems per sub-batch, at most 3 in flight (measured
        S = nprob / 128;                            // on 512 x 4096x256 exact: 1 / 2 / 3 / 4 -> 1032 / 959 / 928 / 1036 ms)
        if (S > 3) S = 3;
        if (o->factor_policy != NLH_FACTOR_EXACT) S = 1;   // the normal-equations pipeline has no long latency-bound
    }                                                      // stage to hide (1 / 2 / 4 -> 40.0 / 40.6 / 41.7 ms)
    if (S > nprob) S = nprob;
    return S < 1 ? 1 : S;
}

int nlh_dq_lm_solve_batch(nlh_handle *h, const nlh_options *o, int32_t nprob, int32_t m, int32_t n,
                          const double *dA, const double *db, double gamma, double *dx, double *dfvec,
                          nlh_iteration_behavior *ib, int32_t *status)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (nprob <= 0) return 0;
    int rc = check_opts_lm(o, m, n);
    if (rc) return rc;
    if (nprob > NLH_MAX_LOCKSTEP)
        return lockstep_slices(nprob, [&](int32_t p0, int32_t cnt) {
            return nlh_dq_lm_solve_batch(h, o, cnt, m, n, dA + (size_t)p0 * m * n, db + (size_t)p0 * m, gamma, dx + (size_t)p0 * n,
                                         dfvec + (size_t)p0 * m, ib ? ib + p0 : nullptr, status ? status + p0 : nullptr);
        });
    const int S = lm_sub_batches(o, nprob, m, n);
    if (S == 1) return lm_solve_range(h, o, nprob, m, n, dA, db, gamma, dx, dfvec, ib, status);
    if ((rc = ensure_workers(h, S))) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));                 // inputs written on the caller's stream are complete
    std::vector<int> rcs(S, 0);
    std::vector<std::thread> pool;
    const size_t mn = (size_t)m * n;
    for (int t = 0; t < S; ++t) {
        const int p0 = (int)((long)nprob * t / S), p1 = (int)((long)nprob * (t + 1) / S);
        pool.emplace_back([&, t, p0, p1]() {
            nlh_handle *wk = h->workers[t];
            wk->timing = h->timing;
            rcs[t] = lm_solve_range(wk, o, p1 - p0, m, n, dA + (size_t)p0 * mn, db + (size_t)p0 * m, gamma,
                                    dx + (size_t)p0 * n, dfvec + (size_t)p0 * m, ib ? ib + p0 : nullptr,
                                    status ? status + p0 : nullptr);
        });
    }
    for (auto &th : pool) th.join();
    for (int t = 0; t < S; ++t) {
        nlh_handle *wk = h->workers[t];
        if (h->timing) {                                        // fold the workers' kernel timers into the caller's
            timing_flush(wk);
            for (int k = 0; k < NLH_K_COUNT; ++k) { h->ms[k] += wk->ms[k]; h->launches[k] += wk->launches[k]; wk->ms[k] = 0; wk->launches[k] = 0; }
        }
        if (rcs[t]) { h->err = wk->err; return rcs[t]; }
    }
    return 0;
}

// ===========================================================================
// Host-callback LM: the same kernels with nprob = 1; residuals come from fcn.
// ===========================================================================
int nlh_lm_solve(nlh_handle *h, const nlh_options *o, int32_t m, int32_t n, nlh_vecfcn fcn,
                 nlh_jacfcn jacfcn, void *ctx, double *x, double *fvec, nlh_iteration_behavior *ib)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (ib) memset(ib, 0, sizeof *ib);                          // :177-185
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;              // :188
    int rc = check_opts_lm(o, m, n);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    LmWs w;
    if ((rc = lm_workspace(h, 1, m, n, w, true))) return rc;
    if ((rc = ensure(h, h->xdev, sizeof(double) * n))) return rc;
    if ((rc = ensure(h, h->fdev, sizeof(double) * m))) return rc;
    const size_t pin_bytes = 256 + sizeof(double) * ((size_t)m * n + 2 * (size_t)m + 2 * (size_t)n);
    if ((rc = ensure_pinned(h, pin_bytes))) return rc;
    LmState *hs = (LmState *)h->pinned;
    double *hP = (double *)((char *)h->pinned + 256);   // m*n panel / Jacobian staging
    double *hf = hP + (size_t)m * n;                     // m
    double *hx = hf + m;                                 // n
    double *dx = (double *)h->xdev.p, *dfvec = (double *)h->fdev.p;
    hipStream_t s = h->stream;

    fcn(ctx, n, x, m, fvec);                                    // :211
    HIPCHK(h, hipMemcpyAsync(dx, x, sizeof(double) * n, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(dfvec, fvec, sizeof(double) * m, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_sumsq_part<RB>, dim3(w.nblk, 1), dim3(RB), 0, s, m, n, dfvec, w.part);
    if (o->factor_policy == NLH_FACTOR_EXACT)
        hipLaunchKernelGGL(k_lm_init_exact, dim3(1), dim3(256), 0, s, m, dfvec, w.st, (int)ST_NEED_JAC);
    else
        hipLaunchKernelGGL(k_lm_init, dim3(1), dim3(64), 0, s, 1, w.nblk, w.part, w.st, (int)ST_NEED_JAC);
    HIPCHK(h, hipStreamSynchronize(s));

    const int max_rounds = o->max_evals + 8;
    int last_printed_iter = -1;
    for (int round = 0; round < max_rounds; ++round) {
        HIPCHK(h, hipMemcpyAsync(hs, w.st, sizeof(LmState), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        if (hs->stage == ST_DONE) break;
        if (hs->stage == ST_NEED_JAC) {
            if (round > 0 && o->print_status && hs->iter != last_printed_iter) {   // :372-374
                print_status(hs->iter, hs->neval, hs->njac, hs->xnorm, hs->fnorm);
                last_printed_iter = hs->iter;
            }
            // vfh_jac_fcn (:221).  x and fvec on the host are kept equal to the device copies.
            if (jacfcn) {
                jacfcn(ctx, n, x, m, hP);
                HIPCHK(h, hipMemcpyAsync(w.J, hP, sizeof(double) * (size_t)m * n, hipMemcpyHostToDevice, s));
            } else {
                for (int j = 0; j < n; ++j) {                   // src/nonlin_multi_eqn_mult_var.f90:267-273
                    const double temp = x[j];
                    double hh = NLH_SQRT_EPS * fabs(temp);
                    if (hh == 0.0) hh = NLH_SQRT_EPS;
                    x[j] = temp + hh;
                    fcn(ctx, n, x, m, hP + (size_t)j * m);
                    x[j] = temp;
                }
                HIPCHK(h, hipMemcpyAsync(w.P, hP, sizeof(double) * (size_t)m * n, hipMemcpyHostToDevice, s));
                launch_fd(h, 1, m, n, w.P, dfvec, dx, w.J, nullptr, -1);      // :274
            }
            hipLaunchKernelGGL(k_stage_advance, dim3(1), dim3(64), 0, s, 1, w.st, (int)ST_NEED_JAC,
                               o->factor_policy != NLH_FACTOR_AUTO ? (int)ST_NEED_QR : (int)ST_HAVE_JAC, 1);
        }
        if ((rc = lm_factor_and_step(h, o, 1, m, n, w, dx, dfvec))) return rc;
        HIPCHK(h, hipMemcpyAsync(hs, w.st, sizeof(LmState), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(hx, w.v.wa2, sizeof(double) * n, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        if (hs->stage == ST_DONE) break;                       // gradient convergence (:270-273)
        if (hs->stage != ST_TRIAL_READY) { h->err = "lm: unexpected stage"; return NLH_ERR_HIP; }
        fcn(ctx, n, hx, m, hf);                                 // :297
        HIPCHK(h, hipMemcpyAsync(w.wa4, hf, sizeof(double) * m, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_sumsq_part<RB>, dim3(w.nblk, 1), dim3(RB), 0, s, m, n, w.wa4, w.part);
        hipLaunchKernelGGL(k_stage_advance, dim3(1), dim3(64), 0, s, 1, w.st, (int)ST_TRIAL_READY, (int)ST_TRIAL_DONE, 0);
        const int iter_before = hs->iter;
        lm_update(h, o, 1, m, n, w, dx, dfvec);
        HIPCHK(h, hipMemcpyAsync(hs, w.st, sizeof(LmState), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        if (hs->iter != iter_before) {                          // accepted: mirror x, fvec on the host (:341-345)
            memcpy(x, hx, sizeof(double) * n);
            memcpy(fvec, hf, sizeof(double) * m);
        }
    }
    HIPCHK(h, hipMemcpyAsync(hs, w.st, sizeof(LmState), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipGetLastError());
    if (ib) fill_ib(*hs, ib);
    return (hs->flag != 0 || hs->stage != ST_DONE) ? NLH_CONVERGENCE_ERROR : 0;
}

// ===========================================================================
// vecfcn_helper%jacobian for host callbacks.
// ===========================================================================
int nlh_fd_jacobian(nlh_handle *h, int32_t m, int32_t n, nlh_vecfcn fcn, nlh_jacfcn jacfcn, void *ctx,
                    double *x, const double *fv, double *jac)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;              // :240
    if (jacfcn) { jacfcn(ctx, n, x, m, jac); return 0; }       // :241-243
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    const size_t mn = (size_t)m * n;
    if ((rc = ensure(h, h->J, sizeof(double) * mn))) return rc;
    if ((rc = ensure(h, h->P, sizeof(double) * mn))) return rc;
    if ((rc = ensure(h, h->xdev, sizeof(double) * n))) return rc;
    if ((rc = ensure(h, h->fdev, sizeof(double) * m))) return rc;
    if ((rc = ensure_pinned(h, sizeof(double) * (mn + m)))) return rc;
    double *hP = (double *)h->pinned, *hf0 = hP + mn;
    if (fv) memcpy(hf0, fv, sizeof(double) * m);
    else fcn(ctx, n, x, m, hf0);                                // :257-259
    for (int j = 0; j < n; ++j) {                               // :267-273
        const double temp = x[j];
        double hh = NLH_SQRT_EPS * fabs(temp);
        if (hh == 0.0) hh = NLH_SQRT_EPS;
        x[j] = temp + hh;
        fcn(ctx, n, x, m, hP + (size_t)j * m);
        x[j] = temp;
    }
    hipStream_t s = h->stream;
    HIPCHK(h, hipMemcpyAsync(h->P.p, hP, sizeof(double) * mn, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(h->fdev.p, hf0, sizeof(double) * m, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(h->xdev.p, x, sizeof(double) * n, hipMemcpyHostToDevice, s));
    launch_fd(h, 1, m, n, (const double *)h->P.p, (const double *)h->fdev.p, (const double *)h->xdev.p,
              (double *)h->J.p, nullptr, -1);
    HIPCHK(h, hipMemcpyAsync(jac, h->J.p, sizeof(double) * mn, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipGetLastError());
    return 0;
}

}  // extern "C"

// ===========================================================================
// Newton: ns_solve as a host loop; Jacobian, gradient, LU on the device.
// The O(n) line-search and convergence arithmetic stays on the host in the
// reference's exact order (sequential dot products), so given the same search
// direction every accept/backtrack decision matches the CPU path.
// ===========================================================================
struct NewtonEval {
    // evaluate F at host x -> host f (device copy of f kept in dfvec when keep_dev)
    std::function<int(const double *x, double *f)> fcn;
    // Jacobian at host x (device copy of f0 in dfvec) -> device J (n x n)
    std::function<int(double *x, const double *f0_host, double *dJ)> jac;
};

static double h_dot(int n, const double *a, const double *b)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s = s + a[i] * b[i];
    return s;
}

// NORM2 as the flang runtime evaluates it (processor-dependent intrinsic); the host-side
// Newton logic uses it only for stpmax and limit_search_vector.
static double h_norm2(int n, const double *x)
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
static double min_backtrack_search(int mode, double f0, double f, double f1, double alam, double alam1, double slope)
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

// ls_search_mimo, src/nonlin_linesearch.f90:152-326
static int line_search(const nlh_options *o, NewtonEval &ev, int n, const double *xold, const double *grad,
                       const double *dir, double *x, double *fvec, double fold, double *fx, int *fcn_count)
{
    const double tolx = 2.0 * DBL_EPSILON, alpha = o->ls_alpha, lambdamin = o->ls_factor;
    const int maxeval = o->ls_max_evals;
    int neval = 0, niter = 0, flag = 0, rc = 0;
    double alam, alam1 = 0.0, alamin, f1 = 0.0, slope, test, tmplam = 0.0, f = 0.0;
    *fcn_count = 0;
    slope = h_dot(n, grad, dir);                                // :249-253
    if (slope >= 0.0) return NLH_DIVERGENT_BEHAVIOR_ERROR;
    test = 0.0;                                                 // :256-262
    for (int i = 0; i < n; ++i) {
        const double t = fabs(dir[i]) / fmax(fabs(xold[i]), 1.0);
        if (t > test) test = t;
    }
    alamin = tolx / test;
    alam = 1.0;
    for (;;) {                                                  // :266-310
        for (int i = 0; i < n; ++i) x[i] = xold[i] + alam * dir[i];
        if ((rc = ev.fcn(x, fvec))) return rc;
        f = 0.5 * h_dot(n, fvec, fvec);
        neval += 1;
        niter += 1;
        if (alam < alamin) {                                    // :275-287
            double sq = 0.0;
            for (int i = 0; i < n; ++i) { const double d = x[i] - xold[i]; sq = sq + d * d; }
            if (sqrt(sq) == 0.0) { rc = NLH_CONVERGENCE_ERROR; break; }
            for (int i = 0; i < n; ++i) x[i] = xold[i];
            break;
        } else if (f <= fold + alpha * alam * slope) {          // :288-291
            break;
        } else {
            tmplam = min_backtrack_search(niter, fold, f, f1, alam, alam1, slope);
        }
        alam1 = alam;                                           // :300-302
        f1 = f;
        alam = fmax(tmplam, lambdamin * alam);
        if (neval >= maxeval) { flag = 1; break; }              // :305-309
    }
    *fx = f;
    *fcn_count = neval;
    if (rc) return rc;
    return flag ? NLH_CONVERGENCE_ERROR : 0;
}

static int newton_core(nlh_handle *h, const nlh_options *o, int n, NewtonEval &ev, double *x, double *fvec,
                       nlh_iteration_behavior *ib)
{
    int rc;
    const size_t nn = (size_t)n * n;
    if ((rc = ensure(h, h->J, sizeof(double) * nn))) return rc;
    if ((rc = ensure(h, h->lu, sizeof(double) * nn))) return rc;
    if ((rc = ensure(h, h->gvec, sizeof(double) * 2 * n))) return rc;
    if ((rc = ensure(h, h->ipvt, sizeof(int32_t) * n))) return rc;
    if ((rc = ensure(h, h->fdev, sizeof(double) * n))) return rc;
    double *dJ = (double *)h->J.p, *dLU = (double *)h->lu.p, *dgrad = (double *)h->gvec.p, *drhs = dgrad + n;
    double *dfvec = (double *)h->fdev.p;
    int32_t *dipvt = (int32_t *)h->ipvt.p;
    hipStream_t s = h->stream;
    std::vector<double> dir(n), grad(n), xold(n), rhs(n);
    int xcnvrg = 0, fcnvrg = 0, gcnvrg = 0, neval = 0, iter = 0, njac = 0, flag = 0;
    double f, fold, stpmax, xnorm = 0, fnorm = 0, test;
    rc = 0;

    // :535  Jacobian requested before fvec is defined; result discarded, not counted.
    if ((rc = ev.jac(x, fvec, dJ))) return rc;

    if ((rc = ev.fcn(x, fvec))) return rc;                      // :538-547
    f = 0.5 * h_dot(n, fvec, fvec);
    neval += 1;
    test = 0.0;
    for (int i = 0; i < n; ++i) test = fmax(fabs(fvec[i]), test);
    if (test < o->ftol) fcnvrg = 1;

    if (!fcnvrg) {
        stpmax = 100.0 * fmax(h_norm2(n, x), (double)n);        // :553
        for (;;) {                                              // :556-620
            iter += 1;
            if ((rc = ev.jac(x, fvec, dJ))) break;              // :561-562
            njac += 1;
            // grad(i) = dot(jac(:,i), fvec)  (:565-567), rows ascending as in the reference
            HIPCHK(h, hipMemcpyAsync(dfvec, fvec, sizeof(double) * n, hipMemcpyHostToDevice, s));
            {
                Timed t(h, NLH_K_JTF);
                hipLaunchKernelGGL(k_jtf_exact, dim3((n + 255) / 256, 1), dim3(256), 0, s, n, n, (const double *)dJ, (const double *)dfvec, dgrad,
                                   (const LmState *)nullptr, -1);
            }
            // LU of a copy (:570) and solve for -fvec (:577)
            HIPCHK(h, hipMemcpyAsync(dLU, dJ, sizeof(double) * nn, hipMemcpyDeviceToDevice, s));
            launch_lu_factor(h, 1, n, dLU, dipvt, (int32_t *)nullptr);
            for (int i = 0; i < n; ++i) rhs[i] = -fvec[i];
            HIPCHK(h, hipMemcpyAsync(drhs, rhs.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_lu_solve, dim3(1), dim3(n >= 96 ? 1024 : 256), sizeof(double) * n, s, n, (const double *)dLU, (const int32_t *)dipvt, drhs,
                               (const LmState *)nullptr, -1);
            HIPCHK(h, hipMemcpyAsync(dir.data(), drhs, sizeof(double) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipMemcpyAsync(grad.data(), dgrad, sizeof(double) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipStreamSynchronize(s));

            memcpy(xold.data(), x, sizeof(double) * n);         // :573-574
            fold = f;

            if (o->use_line_search) {                           // :580-589
                const double temp = h_dot(n, dir.data(), dir.data());
                if (temp > stpmax) {
                    const double sc = stpmax / temp;
                    for (int i = 0; i < n; ++i) dir[i] = dir[i] * sc;
                }
                const double mag = h_norm2(n, dir.data());      // limit_search_vector, linesearch.f90:554-572
                if (mag != 0.0 && mag > stpmax) {
                    const double sc = stpmax / mag;
                    for (int i = 0; i < n; ++i) dir[i] = sc * dir[i];
                }
                int lcount = 0;
                rc = line_search(o, ev, n, xold.data(), grad.data(), dir.data(), x, fvec, fold, &f, &lcount);
                neval += lcount;
                if (rc) break;
            } else {                                            // :591-595
                for (int i = 0; i < n; ++i) x[i] = x[i] + dir[i];
                if ((rc = ev.fcn(x, fvec))) break;
                f = 0.5 * h_dot(n, fvec, fvec);
                neval += 1;
            }

            // test_convergence, src/nonlin_helper.f90:36-124
            int check = 0;
            xcnvrg = fcnvrg = gcnvrg = 0;
            {
                const double fc = 0.5 * h_dot(n, fvec, fvec);
                fnorm = 0.0; xnorm = 0.0;
                for (int i = 0; i < n; ++i) fnorm = fmax(fabs(fvec[i]), fnorm);
                if (fnorm < o->ftol) { fcnvrg = 1; check = 1; }
                else {
                    for (int i = 0; i < n; ++i) {
                        const double t = fabs(x[i] - xold[i]) / fmax(fabs(x[i]), 1.0);
                        xnorm = fmax(t, xnorm);
                    }
                    if (xnorm < o->xtol) { xcnvrg = 1; check = 1; }
                    else {
                        double tg = 0.0;
                        const double den = fmax(fc, 0.5 * (double)n);
                        for (int i = 0; i < n; ++i) tg = fmax(tg, fabs(grad[i]) * fmax(fabs(x[i]), 1.0) / den);
                        if (tg < o->gtol) gcnvrg = 1;
                    }
                }
            }
            if (check) break;
            if (gcnvrg) { rc = NLH_SPURIOUS_CONVERGENCE_ERROR; break; }     // :604-608
            if (o->print_status) print_status(iter, neval, njac, xnorm, fnorm);   // :611-613
            if (neval >= o->max_evals) { flag = 1; break; }     // :616-619
        }
    }
    if (ib) {                                                   // :624-632
        ib->iter_count = iter; ib->fcn_count = neval; ib->jacobian_count = njac; ib->gradient_count = 0;
        ib->converge_on_fcn = fcnvrg; ib->converge_on_chng = xcnvrg; ib->converge_on_zero_diff = gcnvrg;
    }
    if (rc) return rc;
    return flag ? NLH_CONVERGENCE_ERROR : 0;
}


// ===========================================================================
// Quasi-Newton (Broyden): qns_solve as a host loop; B, Q, R live on the device.
// Same division of labour as Newton: O(n) vector logic on the host in the reference's order,
// every O(n^2)/O(n^3) operation in the kernels of nlh_kernels_broyden.h.
// ===========================================================================
static const int QN_MAX_N = 4096;      // k_qn_retri: 4 columns per thread at most
static const int QN_LDS_ROWS = 18000;  // up to here k_qn_house_dot keeps the reflector (rows doubles) in LDS; beyond: in global memory

// B (column-major) -> Q, R: Householder QR with Q formed (qr_factor(b, q = q, r = r), :289)
// Householder steps on the row-major work array [A | E] (rows x ncA | rows x ncE); vbuf slot 0 must hold column 0 of A.
static void launch_house_steps(nlh_handle *h, int nprob, int rows, int ncA, int ncE, double *dA, double *dE,
                               double *vbuf, double *wbuf, double *st, const LmState *gst = nullptr, int gwant = -1)
{
    // vbuf: [nprob][2][rows]; wbuf: room for [nprob][2][ncA + ncE]; st: room for [nprob][2][4]
    hipStream_t s = h->stream;
    const int steps = std::min(ncA, rows - 1), nc = ncA + ncE;
    if (steps < 1) return;
    if (rows <= QN_FUSED_MAXROWS) {
        // one pass per step: the update of step j-1 rides along with the sums of step j
        const size_t sh3 = sizeof(double) * (2 * (size_t)rows + 2 * QN_DOT2_TR * QN_DOT2_CG);
        const bool skinny = (long)nc * nprob < 1536;     // few columns in total: 4 per workgroup so that the chip has work
        for (int j = 0; j < steps; ++j) {
            if (skinny)
                hipLaunchKernelGGL(k_qn_house_fused<4>, dim3((nc + 3) / 4, nprob), dim3(256), sh3, s,
                                   rows, ncA, ncE, j, dA, dE, vbuf, wbuf, st, gst, gwant);
            else
                hipLaunchKernelGGL(k_qn_house_fused<16>, dim3((nc + 15) / 16, nprob), dim3(256), sh3, s,
                                   rows, ncA, ncE, j, dA, dE, vbuf, wbuf, st, gst, gwant);
        }
        const int jl = steps - 1, slot = jl & 1;
        hipLaunchKernelGGL(k_qn_house_apply, dim3((nc + 255) / 256, (rows - jl + QN_RC - 1) / QN_RC, nprob), dim3(256), 0, s,
                           rows, ncA, ncE, jl, dA, dE, vbuf, wbuf + (size_t)slot * nc, st + (size_t)slot * 4, 2 * nc, 8, 1, gst, gwant);
        return;
    }
    const bool wide = rows <= QN_DOT2_MAXROWS;          // workgroup-wide loads (reflector + two product tiles in LDS)
    const size_t sh2 = sizeof(double) * ((size_t)rows + 2 * QN_DOT2_TR * QN_DOT2_CG);
    for (int j = 0; j < steps; ++j) {
        if (wide)
            hipLaunchKernelGGL(k_qn_house_dot2, dim3((nc + QN_DOT2_CG - 1) / QN_DOT2_CG, nprob), dim3(256), sh2, s,
                               rows, ncA, ncE, j, dA, dE, vbuf, wbuf, st, gst, gwant);
        else if (rows <= QN_LDS_ROWS)
            hipLaunchKernelGGL(k_qn_house_dot<false>, dim3((nc + QN_DOT_BS - 1) / QN_DOT_BS, nprob), dim3(QN_DOT_BS),
                               sizeof(double) * rows, s, rows, ncA, ncE, j, dA, dE, vbuf, wbuf, st, gst, gwant);
        else                                                // the reflector does not fit LDS: it stays in global memory
            hipLaunchKernelGGL(k_qn_house_dot<true>, dim3((nc + QN_DOT_BS - 1) / QN_DOT_BS, nprob), dim3(QN_DOT_BS),
                               0, s, rows, ncA, ncE, j, dA, dE, vbuf, wbuf, st, gst, gwant);
        hipLaunchKernelGGL(k_qn_house_apply, dim3((nc + 255) / 256, (rows - j + QN_RC - 1) / QN_RC, nprob), dim3(256), 0, s,
                           rows, ncA, ncE, j, dA, dE, vbuf, wbuf, st, nc, 4, 0, gst, gwant);
    }
}

static void launch_qn_qr(nlh_handle *h, int nprob, int n, const double *dB, double *dQ, double *dRt, double *dvb,
                         const LmState *gst = nullptr, int gwant = -1)
{
    // dvb: per problem 2n (reflector column, two slots) + 2 x 2n (w, two slots) + 2 x 4 (tau, scal, beta)
    hipStream_t s = h->stream;
    double *vbuf = dvb, *wbuf = dvb + (size_t)nprob * 2 * n, *st = wbuf + (size_t)nprob * 4 * n;
    {
        dim3 grid((n + 31) / 32, (n + 31) / 32, nprob);
        hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, s, n, n, dB, dRt, n, gst, gwant);
    }
    hipLaunchKernelGGL(k_qn_qr_init, dim3(std::min(1024, (n * n + 255) / 256), nprob), dim3(256), 0, s, n, dRt, dQ, vbuf, gst, gwant);
    launch_house_steps(h, nprob, n, n, n, dRt, dQ, vbuf, wbuf, st, gst, gwant);
}

// Q1 R1 = Q R + u v^T (qr_rank1_update(q, r, s, dx), :307).  dwcs: 3n doubles per problem of scratch.
static void launch_qn_update(nlh_handle *h, int nprob, int n, double *dQ, double *dRt, const double *du,
                             const double *dv, double *dwcs, const LmState *gst = nullptr, int gwant = -1)
{
    hipStream_t s = h->stream;
    double *dw = dwcs, *dc = dwcs + (size_t)nprob * n, *dsn = dc + (size_t)nprob * n;
    hipLaunchKernelGGL(k_qn_colsdot, dim3((n + 15) / 16, nprob), dim3(256), 0, s, n, n, dQ, du, dw, 1.0, gst, gwant);
    hipLaunchKernelGGL(k_qn_fold, dim3(nprob), dim3(64), 0, s, n, dw, dc, dsn, gst, gwant);
    const dim3 g1((n + 255) / 256, nprob);
    hipLaunchKernelGGL(k_qn_rot_q, g1, dim3(256), sizeof(double) * 2 * n, s, n, dQ, dc, dsn, 1, gst, gwant);
    hipLaunchKernelGGL(k_qn_hess_r, g1, dim3(256), sizeof(double) * 2 * n, s, n, dRt, dc, dsn, dw, dv, gst, gwant);
    if (n <= 1024) {
        const int bs = std::min(1024, ((n + 63) / 64) * 64);
        hipLaunchKernelGGL(k_qn_retri<1>, dim3(nprob), dim3(bs), sizeof(double) * 2 * n, s, n, dRt, dc, dsn, gst, gwant);
    } else {
        hipLaunchKernelGGL(k_qn_retri<4>, dim3(nprob), dim3(1024), sizeof(double) * 2 * n, s, n, dRt, dc, dsn, gst, gwant);
    }
    hipLaunchKernelGGL(k_qn_rot_q, g1, dim3(256), sizeof(double) * 2 * n, s, n, dQ, dc, dsn, 0, gst, gwant);
}

static int quasi_newton_core(nlh_handle *h, const nlh_options *o, int jdelta, int n, NewtonEval &ev, double *x,
                             double *fvec, nlh_iteration_behavior *ib)
{
    int rc;
    const size_t nn = (size_t)n * n;
    if (n > QN_MAX_N) return NLH_ARRAY_SIZE_ERROR;
    if ((rc = ensure(h, h->J, sizeof(double) * nn))) return rc;
    if ((rc = ensure(h, h->qnQ, sizeof(double) * nn))) return rc;
    if ((rc = ensure(h, h->qnR, sizeof(double) * nn))) return rc;
    if ((rc = ensure(h, h->qnV, sizeof(double) * ((size_t)16 * n + 8)))) return rc;
    double *dB = (double *)h->J.p, *dQ = (double *)h->qnQ.p, *dRt = (double *)h->qnR.p;
    double *dv = (double *)h->qnV.p;
    double *ddx = dv, *ddf = dv + n, *dsv = dv + 2 * n, *dwcs = dv + 3 * n /* 3n */, *dgrad = dv + 6 * n,
           *dstep = dv + 7 * n, *dfv = dv + 8 * n, *dvb = dv + 9 * n /* 4n + 4 */;
    hipStream_t s = h->stream;
    std::vector<double> dx(n), df(n), fvold(n), xold(n);
    int restart = 1, xcnvrg = 0, fcnvrg = 0, gcnvrg = 0, neval = 0, iter = 0, njac = 0, flag = 0, jcount = 0;
    int ls_zero_diff = 0;                                       // lib%converge_on_zero_diff: .false. after every search (:318 of linesearch)
    double f, fold, stpmax, xnorm = 0, fnorm = 0, test;
    rc = 0;

    if ((rc = ev.fcn(x, fvec))) return rc;                      // :261-270
    f = 0.5 * h_dot(n, fvec, fvec);
    neval += 1;
    test = 0.0;
    for (int i = 0; i < n; ++i) test = fmax(fabs(fvec[i]), test);
    if (test < o->ftol) fcnvrg = 1;

    if (!fcnvrg) {
        stpmax = 100.0 * fmax(h_norm2(n, x), (double)n);        // :276
        for (;;) {                                              // :279-411
            iter += 1;
            if (restart) {                                      // :284-292
                if ((rc = ev.jac(x, fvec, dB))) break;
                njac += 1;
                launch_qn_qr(h, 1, n, dB, dQ, dRt, dvb);
                jcount = 0;
            } else {                                            // :294-310
                for (int i = 0; i < n; ++i) df[i] = fvec[i] - fvold[i];
                for (int i = 0; i < n; ++i) dx[i] = x[i] - xold[i];
                const double x2 = h_dot(n, dx.data(), dx.data());
                HIPCHK(h, hipMemcpyAsync(ddx, dx.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
                HIPCHK(h, hipMemcpyAsync(ddf, df.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(k_qn_resid, dim3((n + 255) / 256, 1), dim3(256), sizeof(double) * n, s, n, dB, ddx, ddf, x2, (const double *)nullptr, dsv, (const LmState *)nullptr, -1);
                hipLaunchKernelGGL(k_qn_rank1, dim3((n + 255) / 256, n, 1), dim3(256), 0, s, n, dB, dsv, ddx, (const LmState *)nullptr, -1);
                launch_qn_update(h, 1, n, dQ, dRt, dsv, ddx, dwcs);
                jcount += 1;
            }
            // grad = B^T f (:313), step = -R^-1 Q^T f (:322-328)
            HIPCHK(h, hipMemcpyAsync(dfv, fvec, sizeof(double) * n, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_qn_colsdot, dim3((n + 15) / 16, 1), dim3(256), 0, s, n, n, dB, dfv, dgrad, 1.0, (const LmState *)nullptr, -1);
            hipLaunchKernelGGL(k_qn_colsdot, dim3((n + 15) / 16, 1), dim3(256), 0, s, n, n, dQ, dfv, dstep, -1.0, (const LmState *)nullptr, -1);
            hipLaunchKernelGGL(k_qn_solve_upper, dim3(1), dim3(std::min(1024, ((n + 63) / 64) * 64)), sizeof(double) * n, s, n, dRt, dstep, (size_t)n * n, (size_t)n, (const LmState *)nullptr, -1);
            HIPCHK(h, hipMemcpyAsync(dx.data(), dgrad, sizeof(double) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipMemcpyAsync(df.data(), dstep, sizeof(double) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipStreamSynchronize(s));

            memcpy(xold.data(), x, sizeof(double) * n);         // :316-318
            memcpy(fvold.data(), fvec, sizeof(double) * n);
            fold = f;

            double temp = h_dot(n, dx.data(), df.data());       // :332-339
            if (temp >= 0.0) {
                restart = 1;
                if (o->print_status) print_status(iter, neval, njac, xnorm, fnorm);
                if (iter > 10 * o->max_evals + 100) { flag = 1; break; }    // the reference would spin here
                continue;
            }

            if (o->use_line_search) {                           // :342-351
                temp = h_dot(n, df.data(), df.data());
                if (temp > stpmax) {
                    const double sc = stpmax / temp;
                    for (int i = 0; i < n; ++i) df[i] = df[i] * sc;
                }
                const double mag = h_norm2(n, df.data());       // limit_search_vector
                if (mag != 0.0 && mag > stpmax) {
                    const double sc = stpmax / mag;
                    for (int i = 0; i < n; ++i) df[i] = sc * df[i];
                }
                int lcount = 0;
                rc = line_search(o, ev, n, xold.data(), dx.data(), df.data(), x, fvec, fold, &f, &lcount);
                neval += lcount;
                ls_zero_diff = 0;
                if (rc) break;
            } else {                                            // :353-357
                for (int i = 0; i < n; ++i) x[i] = x[i] + df[i];
                if ((rc = ev.fcn(x, fvec))) break;
                f = 0.5 * h_dot(n, fvec, fvec);
                neval += 1;
            }

            // test_convergence (:360-367); the gradient test runs only if the search reported a zero slope
            int check = 0;
            xcnvrg = fcnvrg = gcnvrg = 0;
            {
                const double fc = 0.5 * h_dot(n, fvec, fvec);
                fnorm = 0.0; xnorm = 0.0;
                for (int i = 0; i < n; ++i) fnorm = fmax(fabs(fvec[i]), fnorm);
                if (fnorm < o->ftol) { fcnvrg = 1; check = 1; }
                else {
                    for (int i = 0; i < n; ++i) {
                        const double t = fabs(x[i] - xold[i]) / fmax(fabs(x[i]), 1.0);
                        xnorm = fmax(t, xnorm);
                    }
                    if (xnorm < o->xtol) { xcnvrg = 1; check = 1; }
                    else if (ls_zero_diff && o->use_line_search) {
                        double tg = 0.0;
                        const double den = fmax(fc, 0.5 * (double)n);
                        for (int i = 0; i < n; ++i) tg = fmax(tg, fabs(dx[i]) * fmax(fabs(x[i]), 1.0) / den);
                        if (tg < o->gtol) gcnvrg = 1;
                    }
                }
            }
            if (!check) {                                       // :368-391
                if (gcnvrg) {
                    if (restart) { rc = NLH_SPURIOUS_CONVERGENCE_ERROR; break; }
                    restart = 1;
                } else {
                    restart = jcount >= jdelta ? 1 : 0;
                }
            } else {
                break;
            }
            if (o->print_status) print_status(iter, neval, njac, xnorm, fnorm);   // :398-400
            if (neval >= o->max_evals) { flag = 1; break; }     // :403-406
        }
    }
    if (ib) {                                                   // :414-422
        ib->iter_count = iter; ib->fcn_count = neval; ib->jacobian_count = njac; ib->gradient_count = 0;
        ib->converge_on_fcn = fcnvrg; ib->converge_on_chng = xcnvrg; ib->converge_on_zero_diff = gcnvrg;
    }
    if (rc) return rc;
    return flag ? NLH_CONVERGENCE_ERROR : 0;
}


// ===========================================================================
// Constrained least squares: cls_solve (src/nonlin_least_squares.f90:938-1176) as a host loop.
// Device: FD / analytic Jacobian, Householder QR of J with the reflectors applied to f
// (qr_factor :1047 + solve_qr :1334), the triangular solve, J^T f and the two J v products of dogleg
// (:1331, :1341, :1398).  Host, in the reference's order: Coleman-Li scaling, the dog-leg logic,
// alpha_box, the reduction ratio, the trust-region update and the Armijo fallback.
// ===========================================================================
struct ClsEval {
    std::function<int(const double *x, double *f)> fcn;                          // F at host x -> host f
    std::function<int(double *x, const double *f0_host, double *dJ)> jac;        // Jacobian at host x -> device J (m x n)
};

static double cls_scaled_norm(int n, const double *x, const double *s, double *tmp)   // :1263-1273
{
    for (int i = 0; i < n; ++i) tmp[i] = x[i] * s[i];
    return h_norm2(n, tmp);
}

static bool cls_is_finite(int n, const double *x)               // :1276-1298
{
    for (int i = 0; i < n; ++i) {
        if (!(x[i] == x[i])) return false;
        if (fabs(x[i]) == DBL_MAX) return false;
    }
    return true;
}

static void cls_apply_limits(int n, const double *xl, const double *xu, double *x)   // :858-883
{
    for (int i = 0; i < n; ++i) if (x[i] < xl[i]) x[i] = xl[i];
    for (int i = 0; i < n; ++i) if (x[i] > xu[i]) x[i] = xu[i];
}

static double cls_alpha_box(int n, const double *x, const double *p, const double *xl, const double *xu)   // :1181-1219
{
    double rst = DBL_MAX;
    for (int i = 0; i < n; ++i) {
        if (p[i] > 0.0) {
            if (xu[i] < x[i]) return 0.0;
            const double a = (xu[i] - x[i]) / p[i];
            if (a < rst) rst = a;
        } else if (p[i] < 0.0) {
            if (xl[i] > x[i]) return 0.0;
            const double a = (xl[i] - x[i]) / p[i];
            if (a < rst) rst = a;
        }
    }
    if (rst < 0.0) rst = 0.0;
    return rst;
}

static int cls_core(nlh_handle *h, const nlh_options *o, double delta0, double stepscale0, const double *xl_in,
                    const double *xu_in, int m, int n, ClsEval &ev, double *x, double *fvec, nlh_iteration_behavior *ib)
{
    int rc;
    const size_t mn = (size_t)m * n;
    if ((rc = ensure(h, h->J, sizeof(double) * mn))) return rc;
    if ((rc = ensure(h, h->W2, sizeof(double) * mn))) return rc;
    if ((rc = ensure(h, h->qnV, sizeof(double) * ((size_t)6 * m + 6 * n + 16)))) return rc;
    double *dJ = (double *)h->J.p, *dW = (double *)h->W2.p, *dv = (double *)h->qnV.p;
    double *dE = dv, *dfv = dv + m, *dJv = dv + 2 * m, *vbuf = dv + 3 * m /* 2m */, *dgv = dv + 5 * m, *dvec = dgv + n,
           *dstep = dvec + n, *wbuf = dstep + n /* 2 (n + 1) */, *st = wbuf + 2 * (n + 1);
    hipStream_t s = h->stream;
    std::vector<double> xl(n), xu(n), sc(n), g(n), p(n), xnew(n), tmp(n), pgn(n), psd(n), u(n), v(n), Jg(m), Jp(m), fnew(m);
    int converged = 0, xcnvrg = 0, fcnvrg = 0, gcnvrg = 0, neval = 0, iter = 0, njac = 0;
    double xnorm, fnorm, gnorm, fnewnorm, actred = 0, prered = 0, rho = 0, delta, stepscale, dderiv;
    for (int i = 0; i < n; ++i) {                               // :999-1009
        xl[i] = xl_in ? xl_in[i] : -DBL_MAX;
        xu[i] = xu_in ? xu_in[i] : DBL_MAX;
    }
    cls_apply_limits(n, xl.data(), xu.data(), x);               // :1023-1031
    if ((rc = ev.fcn(x, fvec))) return rc;
    neval = 1;
    fnorm = h_norm2(m, fvec);
    xnorm = h_norm2(n, x);
    if (!cls_is_finite(n, x) || !cls_is_finite(m, fvec)) {
        if (ib) { ib->iter_count = 0; ib->fcn_count = 0; ib->jacobian_count = 0; }
        return 0;                                               // the reference returns silently here
    }
    auto matvec = [&](const double *vec, double *out) -> int {  // out = J vec (dgemv 'N')
        HIPCHK(h, hipMemcpyAsync(dvec, vec, sizeof(double) * n, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_matvec_cm, dim3((m + 255) / 256, 1), dim3(256), sizeof(double) * n, s, m, n, dJ, dvec, dJv, (const LmState *)nullptr, -1);
        HIPCHK(h, hipMemcpyAsync(out, dJv, sizeof(double) * m, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        return 0;
    };

    delta = delta0;                                             // :1034
    iter = 1;
    rc = 0;
    for (;;) {                                                  // :1036-1160
        if ((rc = ev.jac(x, fvec, dJ))) break;                  // :1038
        njac += 1;
        if (o->print_status) print_status(iter, neval, njac, xnorm, fnorm);

        // QR of J with the reflectors applied to f (:1047, :1334), g = J^T f (:1331), u = R^-1 (Q^T f)(1:n)
        HIPCHK(h, hipMemcpyAsync(dfv, fvec, sizeof(double) * m, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(dE, dfv, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
        {
            dim3 grid((m + 31) / 32, (n + 31) / 32, 1);
            hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, s, m, n, dJ, dW, n, (const LmState *)nullptr, -1);
        }
        hipLaunchKernelGGL(k_qn_col0, dim3((m + 255) / 256, 1), dim3(256), 0, s, m, n, dW, vbuf);
        launch_house_steps(h, 1, m, n, 1, dW, dE, vbuf, wbuf, st);
        HIPCHK(h, hipMemcpyAsync(dstep, dE, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_qn_solve_upper, dim3(1), dim3(std::min(1024, ((n + 63) / 64) * 64)), sizeof(double) * n, s, n, dW, dstep, (size_t)m * n, (size_t)n, (const LmState *)nullptr, -1);
        hipLaunchKernelGGL(k_qn_colsdot, dim3((n + 15) / 16, 1), dim3(256), 0, s, m, n, dJ, dfv, dgv, 1.0, (const LmState *)nullptr, -1);
        HIPCHK(h, hipMemcpyAsync(u.data(), dstep, sizeof(double) * n, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(g.data(), dgv, sizeof(double) * n, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));

        // coleman_li_scaling, :1222-1260
        for (int i = 0; i < n; ++i) {
            const double big = DBL_MAX;
            double di;
            if (xl[i] > -big && xu[i] < big) di = fmin(x[i] - xl[i], xu[i] - x[i]);
            else if (xl[i] > -big) di = x[i] - xl[i];
            else if (xu[i] < big) di = xu[i] - x[i];
            else di = 1.0;
            di = fmax(di, 1.0e-8);
            sc[i] = 1.0 / di;
            if (sc[i] > 1.0e8) sc[i] = 1.0e8;
        }
        // dogleg, :1301-1403
        for (int i = 0; i < n; ++i) pgn[i] = -u[i];
        const double pgnnorm = cls_scaled_norm(n, pgn.data(), sc.data(), tmp.data());
        if (pgnnorm > delta) {
            if ((rc = matvec(g.data(), Jg.data()))) break;
            const double c1 = h_dot(n, g.data(), g.data()), c2 = h_dot(m, Jg.data(), Jg.data());
            const double alpha = (c2 > 0.0 && c1 > 0.0) ? c1 / c2 : 0.0;
            for (int i = 0; i < n; ++i) psd[i] = -alpha * g[i];
            const double psdnorm = cls_scaled_norm(n, psd.data(), sc.data(), tmp.data());
            if (psdnorm >= delta && psdnorm > 0.0) {
                const double f1 = delta / psdnorm;
                for (int i = 0; i < n; ++i) p[i] = f1 * psd[i];
            } else {
                for (int i = 0; i < n; ++i) u[i] = pgn[i] - psd[i];
                for (int i = 0; i < n; ++i) u[i] = sc[i] * u[i];
                for (int i = 0; i < n; ++i) v[i] = sc[i] * psd[i];
                const double a = h_dot(n, u.data(), u.data());
                const double b = 2.0 * h_dot(n, u.data(), v.data());
                const double c = h_dot(n, v.data(), v.data()) - delta * delta;
                if (a <= 0.0) {
                    for (int i = 0; i < n; ++i) p[i] = psd[i];
                } else {
                    const double arg = fmax(0.0, b * b - 4.0 * a * c);
                    double t;
                    if (arg == 0.0) {
                        t = -b / (2.0 * a);
                    } else {
                        t = (-b + sqrt(arg)) / (2.0 * a);
                        if (t < 0.0 || t > 1.0) t = (-b - sqrt(arg)) / (2.0 * a);
                    }
                    t = fmax(0.0, fmin(1.0, t));
                    for (int i = 0; i < n; ++i) p[i] = psd[i] + t * u[i];
                }
            }
        } else {
            for (int i = 0; i < n; ++i) p[i] = pgn[i];
        }
        {
            const double alpha = cls_alpha_box(n, x, p.data(), xl.data(), xu.data());   // :1392-1395
            if (alpha < 1.0)
                for (int i = 0; i < n; ++i) p[i] = alpha * p[i];
        }
        if ((rc = matvec(p.data(), Jp.data()))) break;          // :1398
        prered = -h_dot(n, g.data(), p.data()) - 0.5 * h_dot(m, Jp.data(), Jp.data());

        xnorm = cls_scaled_norm(n, p.data(), sc.data(), tmp.data());   // :1055-1057
        gnorm = h_norm2(n, g.data());
        for (int i = 0; i < n; ++i) xnew[i] = x[i] + p[i];
        if ((rc = ev.fcn(xnew.data(), fnew.data()))) break;     // :1060-1062
        fnewnorm = h_norm2(m, fnew.data());
        neval += 1;

        actred = 0.5 * (fnorm * fnorm - fnewnorm * fnewnorm);   // :1065-1070
        rho = (prered > 0.0 && actred >= 0.0) ? actred / prered : 0.0;
        if (rho < 0.25) delta = fmax(0.25, 1.0e-12);            // :1073-1077 (constant 0.25: as in the reference)
        else if (rho > 0.75 && fabs(xnorm - delta) < 1.0e-12 * delta) delta = fmin(2.0 * delta, 1.0e3);

        if (rho > 0.1 && fnewnorm <= fnorm) {                   // :1080-1086
            memcpy(x, xnew.data(), sizeof(double) * n);
            cls_apply_limits(n, xl.data(), xu.data(), x);
            memcpy(fvec, fnew.data(), sizeof(double) * m);
            fnorm = fnewnorm;
            iter += 1;
        } else {                                                // :1088-1123
            dderiv = h_dot(n, g.data(), p.data());
            if (dderiv >= 0.0) {
                delta = fmax(0.5 * delta, 1.0e-12);
            } else {
                stepscale = stepscale0;
                int k;
                for (k = 1; k <= 10; ++k) {
                    for (int i = 0; i < n; ++i) xnew[i] = x[i] + stepscale * p[i];
                    cls_apply_limits(n, xl.data(), xu.data(), xnew.data());
                    if ((rc = ev.fcn(xnew.data(), fnew.data()))) break;
                    neval += 1;
                    fnewnorm = h_norm2(m, fnew.data());
                    if (fnewnorm <= fnorm + 1.0e-4 * stepscale * dderiv) {
                        memcpy(x, xnew.data(), sizeof(double) * n);
                        memcpy(fvec, fnew.data(), sizeof(double) * m);
                        fnorm = fnewnorm;
                        iter += 1;
                        delta = fmax(stepscale * xnorm, 1.0e-12);
                        break;
                    }
                    stepscale = stepscale * 0.5;
                }
                if (rc) break;
                if (k > 10) delta = fmax(0.5 * delta, 1.0e-12);
            }
        }
        if (!cls_is_finite(n, x) || !cls_is_finite(m, fvec)) break;     // :1125-1127
        if (xnorm <= o->xtol) { converged = 1; xcnvrg = 1; break; }     // :1130-1149
        if (fabs(actred) <= o->ftol && fabs(prered) <= o->ftol && 0.5 * rho <= 1.0) { converged = 1; fcnvrg = 1; break; }
        if (gnorm <= o->gtol) { converged = 1; gcnvrg = 1; break; }
        if (neval >= o->max_evals) break;
    }
    if (ib) {                                                   // :1163-1170
        ib->iter_count = iter; ib->fcn_count = neval; ib->jacobian_count = njac; ib->gradient_count = 0;
        ib->converge_on_fcn = fcnvrg; ib->converge_on_chng = xcnvrg; ib->converge_on_zero_diff = gcnvrg;
    }
    if (rc) return rc;
    return converged ? 0 : NLH_CONVERGENCE_ERROR;               // :1173-1175
}


// ===========================================================================
// BFGS: bfgs_solve (src/nonlin_optimize.f90:557-770) as a host loop; the Hessian factor R (row-major) and
// B = R^T R live on the device.  Host: ls_search_miso, the convergence tests and the O(n) vector algebra in the
// reference's order.
// ===========================================================================
struct BfgsEval {
    std::function<int(const double *x, double *f)> fcn;                 // objective at host x
    std::function<int(double *x, double fv, double *g)> grad;           // gradient at host x (fv = f(x)) -> host g
};

// ls_search_miso, src/nonlin_linesearch.f90:329-492
static int line_search_scalar(const nlh_options *o, BfgsEval &ev, int n, const double *xold, const double *grad,
                              const double *dir, double *x, double fold, double *fx, int *fcn_count)
{
    const double tolx = 2.0 * DBL_EPSILON, alpha = o->ls_alpha, lambdamin = o->ls_factor;
    const int maxeval = o->ls_max_evals;
    int neval = 0, niter = 0, flag = 0, rc = 0;
    double alam, alam1 = 0.0, alamin, f1 = 0.0, slope, test, tmplam = 0.0, f = 0.0;
    *fcn_count = 0;
    slope = h_dot(n, grad, dir);
    if (slope >= 0.0) return NLH_DIVERGENT_BEHAVIOR_ERROR;
    test = 0.0;
    for (int i = 0; i < n; ++i) {
        const double t = fabs(dir[i]) / fmax(fabs(xold[i]), 1.0);
        if (t > test) test = t;
    }
    alamin = tolx / test;
    alam = 1.0;
    for (;;) {
        for (int i = 0; i < n; ++i) x[i] = xold[i] + alam * dir[i];
        if ((rc = ev.fcn(x, &f))) return rc;
        neval += 1;
        niter += 1;
        if (alam < alamin) {
            double sq = 0.0;
            for (int i = 0; i < n; ++i) { const double d = x[i] - xold[i]; sq = sq + d * d; }
            if (sqrt(sq) == 0.0) { rc = NLH_CONVERGENCE_ERROR; break; }
            for (int i = 0; i < n; ++i) x[i] = xold[i];
            break;
        } else if (f <= fold + alpha * alam * slope) {
            break;
        } else {
            tmplam = min_backtrack_search(niter, fold, f, f1, alam, alam1, slope);
        }
        alam1 = alam;
        f1 = f;
        alam = fmax(tmplam, lambdamin * alam);
        if (neval >= maxeval) { flag = 1; break; }
    }
    *fx = f;
    *fcn_count = neval;
    if (rc) return rc;
    return flag ? NLH_CONVERGENCE_ERROR : 0;
}

static int bfgs_core(nlh_handle *h, const nlh_options *o, int n, BfgsEval &ev, double *x, double *fout,
                     nlh_iteration_behavior *ib)
{
    int rc;
    if (n > QN_MAX_N) return NLH_ARRAY_SIZE_ERROR;
    const size_t nn = (size_t)n * n;
    if ((rc = ensure(h, h->bfB, sizeof(double) * nn))) return rc;
    if ((rc = ensure(h, h->bfR, sizeof(double) * nn))) return rc;
    if ((rc = ensure(h, h->bfV, sizeof(double) * ((size_t)6 * n + 8)))) return rc;
    double *dB = (double *)h->bfB.p, *dR = (double *)h->bfR.p, *dv = (double *)h->bfV.p;
    double *dvec = dv, *dout = dv + n, *du = dv + 2 * n, *dc = dv + 3 * n;
    int *dinfo = (int *)(dv + 4 * n);
    hipStream_t s = h->stream;
    std::vector<double> g(n), dx(n), u(n), v(n), y(n), bdx(n), gold(n), xnew(n);
    int xcnvrg = 0, gcnvrg = 0, neval = 0, ngrad = 0, flag = 0, iter = 0, hinfo = 0;
    double fp, stpmax = 0.0, fret = 0.0, xtest = 0.0, gtest, temp, ydx;
    const int bs1 = std::min(1024, ((n + 63) / 64) * 64);
    rc = 0;

    if ((rc = ev.fcn(x, &fp))) return rc;                       // :633-636
    if ((rc = ev.grad(x, fp, g.data()))) return rc;
    neval = 1;
    ngrad = 1;
    gtest = h_norm2(n, g.data());                               // :639-642
    if (gtest < o->gtol) gcnvrg = 1;

    if (!gcnvrg) {
        for (;;) {                                              // :647-748
            iter += 1;
            if (iter == 1) {                                    // :653-656
                for (int i = 0; i < n; ++i) dx[i] = -g[i];
                stpmax = 100.0 * fmax(h_norm2(n, x), (double)n);
            }
            if (o->use_line_search) {                           // :659-669
                const double mag = h_norm2(n, dx.data());       // limit_search_vector
                if (mag != 0.0 && mag > stpmax) {
                    const double sc = stpmax / mag;
                    for (int i = 0; i < n; ++i) dx[i] = sc * dx[i];
                }
                int lcount = 0;
                rc = line_search_scalar(o, ev, n, x, g.data(), dx.data(), xnew.data(), fp, &fret, &lcount);
                neval += lcount;
                if (rc) break;
                fp = fret;
            } else {
                for (int i = 0; i < n; ++i) xnew[i] = x[i] + dx[i];
                if ((rc = ev.fcn(xnew.data(), &fp))) break;
                neval += 1;
            }
            for (int i = 0; i < n; ++i) {                       // :672-678
                dx[i] = xnew[i] - x[i];
                x[i] = xnew[i];
                gold[i] = g[i];
            }
            if ((rc = ev.grad(x, fp, g.data()))) break;
            ngrad += 1;

            xtest = 0.0;                                        // :681-689
            for (int i = 0; i < n; ++i) {
                temp = fabs(dx[i]) / fmax(fabs(x[i]), 1.0);
                xtest = fmax(temp, xtest);
            }
            if (xtest < o->xtol) { xcnvrg = 1; break; }
            gtest = h_norm2(n, g.data());                       // :692-696
            if (gtest < o->gtol) { gcnvrg = 1; break; }

            for (int i = 0; i < n; ++i) y[i] = g[i] - gold[i];  // :699-700
            ydx = h_dot(n, y.data(), dx.data());
            if (iter == 1) {                                    // :703-706
                temp = sqrt(h_dot(n, y.data(), y.data()) / ydx);
                hipLaunchKernelGGL(k_bf_scaled_identity, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, s, n, temp, dR, (const double *)nullptr, (size_t)0, (const int32_t *)nullptr, (size_t)0, (const LmState *)nullptr, -1);
            }
            // B = R^T R (:709), bdx = B dx (:712)
            hipLaunchKernelGGL(k_bf_rtr, dim3((n + 255) / 256, n), dim3(256), 0, s, n, dR, dB, (const LmState *)nullptr, -1);
            HIPCHK(h, hipMemcpyAsync(dvec, dx.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_matvec_cm, dim3((n + 255) / 256, 1), dim3(256), sizeof(double) * n, s, n, n, dB, dvec, dout, (const LmState *)nullptr, -1);
            HIPCHK(h, hipMemcpyAsync(bdx.data(), dout, sizeof(double) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipStreamSynchronize(s));
            if (ydx > 1.0e-10 && iter > 1) {                    // :715-724
                const double s1 = sqrt(ydx), s2 = sqrt(h_dot(n, dx.data(), bdx.data()));
                for (int i = 0; i < n; ++i) u[i] = y[i] / s1;
                for (int i = 0; i < n; ++i) v[i] = bdx[i] / s2;
                HIPCHK(h, hipMemcpyAsync(du, u.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
                if (n <= 1024) hipLaunchKernelGGL(k_bf_chol_update<1>, dim3(1), dim3(bs1), sizeof(double) * 2 * n, s, n, dR, du, (const LmState *)nullptr, -1);
                else hipLaunchKernelGGL(k_bf_chol_update<4>, dim3(1), dim3(1024), sizeof(double) * 2 * n, s, n, dR, du, (const LmState *)nullptr, -1);
                HIPCHK(h, hipMemcpyAsync(du, v.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(k_bf_solve_upper_t, dim3(1), dim3(bs1), sizeof(double) * n, s, n, dR, du, (const LmState *)nullptr, -1);
                hipLaunchKernelGGL(k_bf_downdate_rot, dim3(1), dim3(64), 0, s, n, du, dc, dinfo, (const LmState *)nullptr, -1);
                hipLaunchKernelGGL(k_bf_downdate_apply, dim3((n + 255) / 256), dim3(256), sizeof(double) * 2 * n, s, n, dR, dc, du, dinfo, (const LmState *)nullptr, -1);
            } else {
                if (n <= 1024) hipLaunchKernelGGL(k_bf_chol_factor<1>, dim3(1), dim3(bs1), sizeof(double) * n, s, n, dB, dR, dinfo, (const LmState *)nullptr, -1);
                else hipLaunchKernelGGL(k_bf_chol_factor<4>, dim3(1), dim3(1024), sizeof(double) * n, s, n, dB, dR, dinfo, (const LmState *)nullptr, -1);
            }
            // dx = -(R^T R)^-1 g (:727)
            for (int i = 0; i < n; ++i) u[i] = -g[i];
            HIPCHK(h, hipMemcpyAsync(dout, u.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_bf_solve_upper_t, dim3(1), dim3(bs1), sizeof(double) * n, s, n, dR, dout, (const LmState *)nullptr, -1);
            hipLaunchKernelGGL(k_qn_solve_upper, dim3(1), dim3(bs1), sizeof(double) * n, s, n, dR, dout, (size_t)n * n, (size_t)n, (const LmState *)nullptr, -1);
            HIPCHK(h, hipMemcpyAsync(dx.data(), dout, sizeof(double) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipMemcpyAsync(&hinfo, dinfo, sizeof(int), hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipStreamSynchronize(s));
            if (hinfo) { rc = NLH_INVALID_OPERATION_ERROR; break; }     // linalg: matrix not positive definite

            if (o->print_status) {                              // :730-737
                printf(" \n");
                printf("Iteration: %d\n", iter);
                printf("Function Evaluations: %d\n", neval);
                char e1[16], e2[16], e3[16];
                format_e10_3(fp, e1); format_e10_3(xtest, e2); format_e10_3(gtest, e3);
                printf("Function Value: %s\nChange in Variable: %s\nGradient: %s\n", e1, e2, e3);
            }
            if (neval >= o->max_evals) { flag = 1; break; }     // :740-743
        }
    }
    if (ib) {                                                   // :751-759
        ib->iter_count = iter; ib->fcn_count = neval; ib->jacobian_count = 0; ib->gradient_count = ngrad;
        ib->converge_on_fcn = 0; ib->converge_on_chng = xcnvrg; ib->converge_on_zero_diff = gcnvrg;
    }
    if (fout) *fout = fp;                                       // :762
    if (rc) return rc;
    return flag ? NLH_CONVERGENCE_ERROR : 0;                    // :765-767
}


// Host-loop solvers (Newton, quasi-Newton, constrained least squares, bfgs) over a batch of independent problems:
// the problems are dealt to a few host threads, each with a private handle (own HIP stream and workspace), so the
// latency-bound kernels of different problems overlap on the device.  NLH_WORKERS sets the thread count (default 8).
// Private handles (own non-blocking stream + workspace) for work the caller's handle deals out to host threads.
static int ensure_workers(nlh_handle *h, int T)
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

static int run_problems(nlh_handle *h, int nprob, const std::function<int(nlh_handle *, int)> &solve_one)
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

extern "C" {

int nlh_newton_solve(nlh_handle *h, const nlh_options *o, int32_t n, nlh_vecfcn fcn, nlh_jacfcn jacfcn,
                     void *ctx, double *x, double *fvec, nlh_iteration_behavior *ib)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (ib) memset(ib, 0, sizeof *ib);
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;              // :518
    if (!o || n < 1) return NLH_INVALID_INPUT_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    const size_t nn = (size_t)n * n;
    if ((rc = ensure(h, h->P, sizeof(double) * nn))) return rc;
    if ((rc = ensure(h, h->xdev, sizeof(double) * n))) return rc;
    if ((rc = ensure(h, h->wa4, sizeof(double) * n))) return rc;
    if ((rc = ensure_pinned(h, sizeof(double) * (nn + n)))) return rc;
    double *hP = (double *)h->pinned;
    hipStream_t s = h->stream;
    NewtonEval ev;
    ev.fcn = [&](const double *xx, double *ff) -> int { fcn(ctx, n, xx, n, ff); return 0; };
    ev.jac = [&](double *xx, const double *f0, double *dJ) -> int {
        if (jacfcn) {
            jacfcn(ctx, n, xx, n, hP);
            HIPCHK(h, hipMemcpyAsync(dJ, hP, sizeof(double) * nn, hipMemcpyHostToDevice, s));
            return 0;
        }
        for (int j = 0; j < n; ++j) {                           // vfh_jac_fcn :267-273
            const double temp = xx[j];
            double hh = NLH_SQRT_EPS * fabs(temp);
            if (hh == 0.0) hh = NLH_SQRT_EPS;
            xx[j] = temp + hh;
            fcn(ctx, n, xx, n, hP + (size_t)j * n);
            xx[j] = temp;
        }
        HIPCHK(h, hipMemcpyAsync(h->P.p, hP, sizeof(double) * nn, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(h->wa4.p, f0, sizeof(double) * n, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(h->xdev.p, xx, sizeof(double) * n, hipMemcpyHostToDevice, s));
        launch_fd(h, 1, n, n, (const double *)h->P.p, (const double *)h->wa4.p, (const double *)h->xdev.p, dJ, nullptr, -1);
        HIPCHK(h, hipStreamSynchronize(s));   // hP is reused by the next call
        return 0;
    };
    rc = newton_core(h, o, n, ev, x, fvec, ib);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return NLH_ERR_HIP; }
    return rc;
}

// newton_solver%solve / quasi_newton_solver%solve for a batch of device-model problems: the lock-step state machine of
// nlh_kernels_newton.h.  A round serves every problem in whatever stage it is.  Newton: the ones that want a Jacobian
// get J (analytic or forward differences), grad = J^T F in the reference's row order, the LU of J in place (the
// reference factors a copy; J is not read again in the iteration), the direction and the set-up of the line search.
// Quasi-Newton (broyden): an iteration starts either from a fresh Jacobian B and its QR factors with Q formed (:284-292)
// or from Broyden's rank-one update of B, Q and R (:294-310); then grad = B^T F, step = -R^-1 Q^T F (:313-328).  The ones
// with a trial point get F(x) and one turn of the search loop / the convergence test.  One 12-byte read-back per round.
static int square_lockstep(nlh_handle *h, const nlh_options *o, bool broyden, int jdelta, int32_t nprob, int32_t n,
                           const double *dA, const double *db, double gamma, int32_t analytic, double *dx, double *dfvec,
                           nlh_iteration_behavior *ib, int32_t *status)
{
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if (broyden && n > QN_MAX_N) return NLH_ARRAY_SIZE_ERROR;
    const size_t nn = (size_t)n * n, np = (size_t)nprob;
    if ((rc = ensure(h, h->J, sizeof(double) * nn * np))) return rc;
    if (!analytic && (rc = ensure(h, h->P, sizeof(double) * nn * np))) return rc;
    if ((rc = ensure(h, h->vecs, sizeof(double) * (broyden ? 17 : 3) * n * np + sizeof(double) * 16 * np))) return rc;
    if ((rc = ensure(h, h->ipvt, sizeof(int32_t) * n * np))) return rc;
    if ((rc = ensure(h, h->state, sizeof(LmState) * np))) return rc;
    if ((rc = ensure(h, h->misc, sizeof(NtState) * np + 64))) return rc;
    if ((rc = ensure_pinned(h, sizeof(NtState) * np + 64))) return rc;
    if (broyden) {
        if ((rc = ensure(h, h->qnQ, sizeof(double) * nn * np))) return rc;
        if ((rc = ensure(h, h->qnR, sizeof(double) * nn * np))) return rc;
    }
    double *dJ = (double *)h->J.p, *dP = (double *)h->P.p;       // J: the Jacobian / its LU; quasi-Newton: B
    double *dxold = (double *)h->vecs.p, *ddir = dxold + (size_t)n * np, *dgrad = ddir + (size_t)n * np;
    // quasi-Newton only: F(xold), dx, df, s, the update's scratch (3n), the QR's reflector / w slots (6n + 8), x2
    double *dfvold = dgrad + (size_t)n * np, *dddx = dfvold + (size_t)n * np, *dddf = dddx + (size_t)n * np;
    double *dsv = dddf + (size_t)n * np, *dwcs = dsv + (size_t)n * np, *dvb = dwcs + 3 * (size_t)n * np;
    double *dx2 = dvb + (6 * (size_t)n + 8) * np;
    double *dQ = (double *)h->qnQ.p, *dRt = (double *)h->qnR.p;
    int32_t *dipvt = (int32_t *)h->ipvt.p;
    LmState *st = (LmState *)h->state.p;
    int32_t *dcounts = (int32_t *)h->misc.p;
    NtState *ns = (NtState *)((char *)h->misc.p + 64);
    int32_t *hcounts = (int32_t *)h->pinned;
    NtState *hns = (NtState *)((char *)h->pinned + 64);
    hipStream_t s = h->stream;
    NtOpts no;
    no.ftol = o->ftol; no.xtol = o->xtol; no.gtol = o->gtol; no.ls_alpha = o->ls_alpha; no.ls_factor = o->ls_factor;
    no.max_evals = o->max_evals; no.ls_max_evals = o->ls_max_evals; no.use_line_search = o->use_line_search ? 1 : 0;
    no.broyden = broyden ? 1 : 0; no.jdelta = jdelta; no.pad = 0;
    const int pb = (nprob + 255) / 256;
    const bool echo = o->print_status && nprob == 1;             // the status block is a single solve's (:611-613)
    auto jacobian = [&]() {                                      // for the problems in stage NT_NEED_JAC
        if (analytic) {
            Timed t(h, NLH_K_DQ_JACOBIAN);
            hipLaunchKernelGGL(k_dq_jacobian<RB>, dim3((n + RB - 1) / RB, nprob), dim3(RB), sizeof(double) * n, s,
                               n, n, dA, gamma, (const double *)dx, dJ, (const LmState *)st, (int)NT_NEED_JAC);
        } else {                                                 // vfh_jac_fcn: n perturbed evaluations, (f1 - f0) / h
            launch_dq_panel(h, nprob, n, n, dA, db, gamma, dx, dP, st, NT_NEED_JAC);
            launch_fd(h, nprob, n, n, dP, dfvec, dx, dJ, st, NT_NEED_JAC);
        }
    };

    // (ns_solve :535 asks for a Jacobian before fvec is defined and discards it: nothing observable for a device model.)
    hipLaunchKernelGGL(k_nt_reset, dim3(pb), dim3(256), 0, s, nprob, st, ns);
    launch_dq_residual(h, nprob, n, n, dA, db, gamma, dx, dfvec, nullptr, st, NT_START);       // :538 / :261
    hipLaunchKernelGGL(k_nt_start, dim3(nprob), dim3(256), 0, s, n, no, (const double *)dx, (const double *)dfvec, st, ns);
    int need_jac = nprob, update = 0;                            // upper bounds until the first read-back
    // a round advances every live problem by one evaluation at least (or, quasi-Newton, turns an iteration without a
    // descent direction into a restart: bounded by 10 max_evals + 100 iterations as in the host loop)
    const long max_rounds = broyden ? 11L * o->max_evals + (long)o->ls_max_evals + 128 : (long)o->max_evals + (long)o->ls_max_evals + 8;
    for (long round = 0; round < max_rounds; ++round) {
        if (!broyden && need_jac > 0) {
            jacobian();
            {
                Timed t(h, NLH_K_JTF);                           // :565-567
                hipLaunchKernelGGL(k_jtf_exact, dim3((n + 255) / 256, nprob), dim3(256), 0, s, n, n, (const double *)dJ,
                                   (const double *)dfvec, dgrad, (const LmState *)st, (int)NT_NEED_JAC);
            }
            hipLaunchKernelGGL(k_nt_rhs, dim3((n + 255) / 256, nprob), dim3(256), 0, s, n, (const double *)dfvec, ddir,
                               (const LmState *)st);
            launch_lu_factor(h, nprob, n, dJ, dipvt, nullptr, st, NT_NEED_JAC);          // :570
            hipLaunchKernelGGL(k_lu_solve, dim3(nprob), dim3(n >= 96 ? 1024 : 256), sizeof(double) * n, s, n, (const double *)dJ,
                               (const int32_t *)dipvt, ddir, (const LmState *)st, (int)NT_NEED_JAC);   // :577
            hipLaunchKernelGGL(k_nt_step_begin, dim3(nprob), dim3(256), 0, s, n, no, (int)NT_NEED_JAC, dx, dxold, ddir, (const double *)dgrad,
                               (const double *)dfvec, (double *)nullptr, st, ns);
        }
        if (broyden && (need_jac > 0 || update > 0)) {
            if (need_jac > 0) {                                  // :284-292: B = J(x), QR with Q formed
                jacobian();
                launch_qn_qr(h, nprob, n, dJ, dQ, dRt, dvb, st, NT_NEED_JAC);
                hipLaunchKernelGGL(k_nt_advance, dim3(pb), dim3(256), 0, s, nprob, st, (int)NT_NEED_JAC, (int)NT_DIR);
            }
            if (update > 0) {                                    // :294-310: B += s dx^T, Q R <- Q R + s dx^T
                hipLaunchKernelGGL(k_qn_prep, dim3(nprob), dim3(256), 0, s, n, (const double *)dx, (const double *)dxold,
                                   (const double *)dfvec, (const double *)dfvold, dddx, dddf, dx2, (const LmState *)st);
                hipLaunchKernelGGL(k_qn_resid, dim3((n + 255) / 256, nprob), dim3(256), sizeof(double) * n, s, n, (const double *)dJ,
                                   (const double *)dddx, (const double *)dddf, 0.0, (const double *)dx2, dsv, (const LmState *)st, (int)NT_UPDATE);
                hipLaunchKernelGGL(k_qn_rank1, dim3((n + 255) / 256, n, nprob), dim3(256), 0, s, n, dJ, (const double *)dsv,
                                   (const double *)dddx, (const LmState *)st, (int)NT_UPDATE);
                launch_qn_update(h, nprob, n, dQ, dRt, dsv, dddx, dwcs, st, NT_UPDATE);
                hipLaunchKernelGGL(k_nt_advance, dim3(pb), dim3(256), 0, s, nprob, st, (int)NT_UPDATE, (int)NT_DIR);
            }
            // grad = B^T F (:313), step = -R^-1 Q^T F (:322-328)
            hipLaunchKernelGGL(k_qn_colsdot, dim3((n + 15) / 16, nprob), dim3(256), 0, s, n, n, (const double *)dJ, (const double *)dfvec,
                               dgrad, 1.0, (const LmState *)st, (int)NT_DIR);
            hipLaunchKernelGGL(k_qn_colsdot, dim3((n + 15) / 16, nprob), dim3(256), 0, s, n, n, (const double *)dQ, (const double *)dfvec,
                               ddir, -1.0, (const LmState *)st, (int)NT_DIR);
            hipLaunchKernelGGL(k_qn_solve_upper, dim3(nprob), dim3(std::min(1024, ((n + 63) / 64) * 64)), sizeof(double) * n, s, n,
                               (const double *)dRt, ddir, nn, (size_t)n, (const LmState *)st, (int)NT_DIR);
            hipLaunchKernelGGL(k_nt_step_begin, dim3(nprob), dim3(256), 0, s, n, no, (int)NT_DIR, dx, dxold, ddir, (const double *)dgrad,
                               (const double *)dfvec, dfvold, st, ns);
        }
        launch_dq_residual(h, nprob, n, n, dA, db, gamma, dx, dfvec, nullptr, st, NT_TRIAL);
        hipLaunchKernelGGL(k_nt_trial, dim3(nprob), dim3(256), 0, s, n, no, dx, (const double *)dxold, (const double *)ddir,
                           (const double *)dgrad, (const double *)dfvec, st, ns);
        hipLaunchKernelGGL(k_nt_count, dim3(1), dim3(256), 0, s, nprob, (const LmState *)st, dcounts);
        HIPCHK(h, hipMemcpyAsync(hcounts, dcounts, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        if (echo) HIPCHK(h, hipMemcpyAsync(hns, ns, sizeof(NtState), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        if (echo && hns[0].print_due) print_status(hns[0].iter, hns[0].neval, hns[0].njac, hns[0].xnorm, hns[0].fnorm);
        need_jac = hcounts[0]; update = hcounts[2];
        if (need_jac == 0 && update == 0 && hcounts[1] == 0) break;
    }
    HIPCHK(h, hipMemcpyAsync(hns, ns, sizeof(NtState) * np, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipGetLastError());
    for (int p = 0; p < nprob; ++p) {
        const NtState &q = hns[p];
        if (ib) {                                                // :624-632 / :414-422
            ib[p].iter_count = q.iter; ib[p].fcn_count = q.neval; ib[p].jacobian_count = q.njac; ib[p].gradient_count = 0;
            ib[p].converge_on_fcn = q.fcnvrg; ib[p].converge_on_chng = q.xcnvrg; ib[p].converge_on_zero_diff = q.gcnvrg;
        }
        const bool finished = q.rc || q.flag || q.fcnvrg || q.xcnvrg;
        if (status) status[p] = q.rc ? q.rc : ((q.flag || !finished) ? NLH_CONVERGENCE_ERROR : 0);       // :635-637 / :425-427
    }
    return 0;
}

int nlh_dq_newton_solve_batch(nlh_handle *h, const nlh_options *o, int32_t nprob, int32_t n, const double *dA,
                              const double *db, double gamma, int32_t analytic, double *dx, double *dfvec,
                              nlh_iteration_behavior *ib, int32_t *status)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (!o || n < 1 || nprob < 1) return NLH_INVALID_INPUT_ERROR;
    return lockstep_slices(nprob, [&](int32_t p0, int32_t cnt) {
        return square_lockstep(h, o, false, 0, cnt, n, dA + (size_t)p0 * n * n, db + (size_t)p0 * n, gamma, analytic, dx + (size_t)p0 * n,
                               dfvec + (size_t)p0 * n, ib ? ib + p0 : nullptr, status ? status + p0 : nullptr);
    });
}

// quasi_newton_solver%solve -- qns_solve, src/nonlin_solve.f90:156-427
int nlh_quasi_newton_solve(nlh_handle *h, const nlh_options *o, int32_t jdelta, int32_t n, nlh_vecfcn fcn,
                           nlh_jacfcn jacfcn, void *ctx, double *x, double *fvec, nlh_iteration_behavior *ib)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (ib) memset(ib, 0, sizeof *ib);
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;              // :240
    if (!o || n < 1) return NLH_INVALID_INPUT_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    const size_t nn = (size_t)n * n;
    if ((rc = ensure(h, h->P, sizeof(double) * nn))) return rc;
    if ((rc = ensure(h, h->xdev, sizeof(double) * n))) return rc;
    if ((rc = ensure(h, h->wa4, sizeof(double) * n))) return rc;
    if ((rc = ensure_pinned(h, sizeof(double) * (nn + n)))) return rc;
    double *hP = (double *)h->pinned;
    hipStream_t s = h->stream;
    NewtonEval ev;
    ev.fcn = [&](const double *xx, double *ff) -> int { fcn(ctx, n, xx, n, ff); return 0; };
    ev.jac = [&](double *xx, const double *f0, double *dJ) -> int {
        if (jacfcn) {
            jacfcn(ctx, n, xx, n, hP);
            HIPCHK(h, hipMemcpyAsync(dJ, hP, sizeof(double) * nn, hipMemcpyHostToDevice, s));
            HIPCHK(h, hipStreamSynchronize(s));
            return 0;
        }
        for (int j = 0; j < n; ++j) {                           // vfh_jac_fcn :267-273
            const double temp = xx[j];
            double hh = NLH_SQRT_EPS * fabs(temp);
            if (hh == 0.0) hh = NLH_SQRT_EPS;
            xx[j] = temp + hh;
            fcn(ctx, n, xx, n, hP + (size_t)j * n);
            xx[j] = temp;
        }
        HIPCHK(h, hipMemcpyAsync(h->P.p, hP, sizeof(double) * nn, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(h->wa4.p, f0, sizeof(double) * n, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(h->xdev.p, xx, sizeof(double) * n, hipMemcpyHostToDevice, s));
        launch_fd(h, 1, n, n, (const double *)h->P.p, (const double *)h->wa4.p, (const double *)h->xdev.p, dJ, nullptr, -1);
        HIPCHK(h, hipStreamSynchronize(s));
        return 0;
    };
    rc = quasi_newton_core(h, o, jdelta, n, ev, x, fvec, ib);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return NLH_ERR_HIP; }
    return rc;
}

int nlh_dq_quasi_newton_solve_batch(nlh_handle *h, const nlh_options *o, int32_t jdelta, int32_t nprob, int32_t n,
                                    const double *dA, const double *db, double gamma, int32_t analytic, double *dx,
                                    double *dfvec, nlh_iteration_behavior *ib, int32_t *status)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (!o || n < 1 || nprob < 1) return NLH_INVALID_INPUT_ERROR;
    return lockstep_slices(nprob, [&](int32_t p0, int32_t cnt) {                                           // the same state machine
        return square_lockstep(h, o, true, jdelta, cnt, n, dA + (size_t)p0 * n * n, db + (size_t)p0 * n, gamma, analytic, dx + (size_t)p0 * n,
                               dfvec + (size_t)p0 * n, ib ? ib + p0 : nullptr, status ? status + p0 : nullptr);
    });
}

// constrained_least_squares_solver%solve -- cls_solve, src/nonlin_least_squares.f90:938-1176
int nlh_cls_solve(nlh_handle *h, const nlh_options *o, double delta0, double stepscale0, const double *xl,
                  const double *xu, int32_t m, int32_t n, nlh_vecfcn fcn, nlh_jacfcn jacfcn, void *ctx, double *x,
                  double *fvec, nlh_iteration_behavior *ib)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (ib) memset(ib, 0, sizeof *ib);
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;              // :988
    if (!o || n < 1 || m < 1) return NLH_INVALID_INPUT_ERROR;
    if (n > m) return NLH_UNDERDEFINED_PROBLEM_ERROR;           // :989
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    const size_t mn = (size_t)m * n;
    if ((rc = ensure(h, h->P, sizeof(double) * mn))) return rc;
    if ((rc = ensure(h, h->xdev, sizeof(double) * n))) return rc;
    if ((rc = ensure(h, h->wa4, sizeof(double) * m))) return rc;
    if ((rc = ensure_pinned(h, sizeof(double) * (mn + m)))) return rc;
    double *hP = (double *)h->pinned;
    hipStream_t s = h->stream;
    ClsEval ev;
    ev.fcn = [&](const double *xx, double *ff) -> int { fcn(ctx, n, xx, m, ff); return 0; };
    ev.jac = [&](double *xx, const double *f0, double *dJ) -> int {
        if (jacfcn) {
            jacfcn(ctx, n, xx, m, hP);
            HIPCHK(h, hipMemcpyAsync(dJ, hP, sizeof(double) * mn, hipMemcpyHostToDevice, s));
            HIPCHK(h, hipStreamSynchronize(s));
            return 0;
        }
        for (int j = 0; j < n; ++j) {                           // vfh_jac_fcn :267-273
            const double temp = xx[j];
            double hh = NLH_SQRT_EPS * fabs(temp);
            if (hh == 0.0) hh = NLH_SQRT_EPS;
            xx[j] = temp + hh;
            fcn(ctx, n, xx, m, hP + (size_t)j * m);
            xx[j] = temp;
        }
        HIPCHK(h, hipMemcpyAsync(h->P.p, hP, sizeof(double) * mn, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(h->wa4.p, f0, sizeof(double) * m, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(h->xdev.p, xx, sizeof(double) * n, hipMemcpyHostToDevice, s));
        launch_fd(h, 1, m, n, (const double *)h->P.p, (const double *)h->wa4.p, (const double *)h->xdev.p, dJ, nullptr, -1);
        HIPCHK(h, hipStreamSynchronize(s));
        return 0;
    };
    rc = cls_core(h, o, delta0, stepscale0, xl, xu, m, n, ev, x, fvec, ib);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return NLH_ERR_HIP; }
    return rc;
}

// constrained_least_squares_solver%solve for a batch of device-model problems: the lock-step state machine of
// nlh_kernels_cls.h.  A round takes every problem that wants a Jacobian through J (forward differences, fused into the
// panel kernel), the QR of J with the reflectors applied to f, the Gauss-Newton step, the gradient and the dog-leg (with
// J g and J p where the step needs them) to its trial point, evaluates F there and runs the ratio test; problems inside
// the projected backtracking get one more trial point evaluated.  One 8-byte read-back per round.
static int cls_lockstep(nlh_handle *h, const nlh_options *o, double delta0, double stepscale0, const double *xl_in,
                        const double *xu_in, int32_t nprob, int32_t m, int32_t n, const double *dA, const double *db,
                        double gamma, double *dx, double *dfvec, nlh_iteration_behavior *ib, int32_t *status)
{
    int rc;
    const size_t mn = (size_t)m * n, np = (size_t)nprob;
    if ((rc = ensure(h, h->J, sizeof(double) * mn * np))) return rc;
    if ((rc = ensure(h, h->W2, sizeof(double) * mn * np))) return rc;
    const size_t per = 5 * (size_t)m + 7 * (size_t)n + 2 * ((size_t)n + 1) + 8;
    if ((rc = ensure(h, h->qnV, sizeof(double) * (per * np + 2 * (size_t)n)))) return rc;
    if ((rc = ensure(h, h->state, sizeof(LmState) * np))) return rc;
    if ((rc = ensure(h, h->misc, sizeof(ClState) * np + 64))) return rc;
    if ((rc = ensure_pinned(h, sizeof(ClState) * np + 64 + sizeof(double) * 2 * (size_t)n))) return rc;
    double *dJ = (double *)h->J.p, *dW = (double *)h->W2.p, *q = (double *)h->qnV.p;
    double *dE = q; q += (size_t)m * np;                         // the QR's extra column, then u in its head
    double *dJv = q; q += (size_t)m * np;                        // J g, later J p
    double *dfnew = q; q += (size_t)m * np;
    double *vbuf = q; q += 2 * (size_t)m * np;
    double *dg = q; q += (size_t)n * np;
    double *dsc = q; q += (size_t)n * np;
    double *dpgn = q; q += (size_t)n * np;
    double *dpsd = q; q += (size_t)n * np;
    double *du = q; q += (size_t)n * np;
    double *dp = q; q += (size_t)n * np;
    double *dxnew = q; q += (size_t)n * np;
    double *wbuf = q; q += 2 * ((size_t)n + 1) * np;
    double *st2 = q; q += 8 * np;
    double *dxl = q, *dxu = q + n;
    LmState *st = (LmState *)h->state.p;
    int32_t *dcounts = (int32_t *)h->misc.p;
    ClState *cs = (ClState *)((char *)h->misc.p + 64);
    int32_t *hcounts = (int32_t *)h->pinned;
    ClState *hcs = (ClState *)((char *)h->pinned + 64);
    double *hb = (double *)((char *)h->pinned + 64 + sizeof(ClState) * np);
    hipStream_t s = h->stream;
    for (int i = 0; i < n; ++i) {                                // :999-1009
        hb[i] = xl_in ? xl_in[i] : -DBL_MAX;
        hb[n + i] = xu_in ? xu_in[i] : DBL_MAX;
    }
    HIPCHK(h, hipMemcpyAsync(dxl, hb, sizeof(double) * 2 * n, hipMemcpyHostToDevice, s));
    ClOpts co;
    co.ftol = o->ftol; co.xtol = o->xtol; co.gtol = o->gtol; co.delta0 = delta0; co.stepscale0 = stepscale0;
    co.max_evals = o->max_evals; co.pad = 0;
    const int pb = (nprob + 255) / 256;
    const bool echo = o->print_status && nprob == 1;
    const int bsn = std::min(1024, ((n + 63) / 64) * 64);

    hipLaunchKernelGGL(k_cls_reset, dim3(pb), dim3(256), 0, s, nprob, delta0, st, cs);
    hipLaunchKernelGGL(k_cls_limits, dim3((n + 255) / 256, nprob), dim3(256), 0, s, n, (const double *)dxl, (const double *)dxu, dx);   // :1023
    launch_dq_residual(h, nprob, m, n, dA, db, gamma, dx, dfvec, nullptr, st, CL_START);
    hipLaunchKernelGGL(k_cls_start, dim3(nprob), dim3(256), 0, s, m, n, (const double *)dx, (const double *)dfvec, st, cs);
    int need_jac = nprob;                                        // upper bound until the first read-back
    // a round is an iteration or one backtracking trial: every one of them costs its problem an evaluation
    const long max_rounds = (long)o->max_evals + 16;
    for (long round = 0; round < max_rounds; ++round) {
        if (need_jac > 0) {
            launch_dq_panel(h, nprob, m, n, dA, db, gamma, dx, dJ, st, CL_NEED_JAC, dfvec);     // :1038, fused FD column write
            hipLaunchKernelGGL(k_cls_qr_prep, dim3((m + 255) / 256, nprob), dim3(256), 0, s, m, (const double *)dfvec, dE, (const LmState *)st);
            {
                dim3 grid((m + 31) / 32, (n + 31) / 32, nprob);
                hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, s, m, n, (const double *)dJ, dW, n, (const LmState *)st, (int)CL_NEED_JAC);
            }
            hipLaunchKernelGGL(k_qn_col0, dim3((m + 255) / 256, nprob), dim3(256), 0, s, m, n, (const double *)dW, vbuf);
            launch_house_steps(h, nprob, m, n, 1, dW, dE, vbuf, wbuf, st2, st, CL_NEED_JAC);
            hipLaunchKernelGGL(k_qn_solve_upper, dim3(nprob), dim3(bsn), sizeof(double) * n, s, n, (const double *)dW, dE, mn, (size_t)m,
                               (const LmState *)st, (int)CL_NEED_JAC);
            hipLaunchKernelGGL(k_qn_colsdot, dim3((n + 15) / 16, nprob), dim3(256), 0, s, m, n, (const double *)dJ, (const double *)dfvec, dg, 1.0,
                               (const LmState *)st, (int)CL_NEED_JAC);
            hipLaunchKernelGGL(k_cls_dog1, dim3(nprob), dim3(256), 0, s, m, n, (const double *)dx, (const double *)dxl, (const double *)dxu,
                               (const double *)dE, dsc, dpgn, dp, st, cs);
            hipLaunchKernelGGL(k_matvec_cm, dim3((m + 255) / 256, nprob), dim3(256), sizeof(double) * n, s, m, n, (const double *)dJ,
                               (const double *)dg, dJv, (const LmState *)st, (int)CL_DOG_SD);
            hipLaunchKernelGGL(k_cls_dog2, dim3(nprob), dim3(256), 0, s, m, n, (const double *)dx, (const double *)dxl, (const double *)dxu,
                               (const double *)dg, (const double *)dJv, (const double *)dsc, (const double *)dpgn, dpsd, du, dp, st, cs);
            hipLaunchKernelGGL(k_matvec_cm, dim3((m + 255) / 256, nprob), dim3(256), sizeof(double) * n, s, m, n, (const double *)dJ,
                               (const double *)dp, dJv, (const LmState *)st, (int)CL_PRED);
            hipLaunchKernelGGL(k_cls_pred, dim3(nprob), dim3(256), 0, s, m, n, (const double *)dx, (const double *)dg, (const double *)dp,
                               (const double *)dJv, (const double *)dsc, dxnew, st, cs);
            launch_dq_residual(h, nprob, m, n, dA, db, gamma, dxnew, dfnew, nullptr, st, CL_TRIAL);
            hipLaunchKernelGGL(k_cls_judge, dim3(nprob), dim3(256), 0, s, m, n, co, dx, dxnew, dfvec, (const double *)dfnew, (const double *)dg,
                               (const double *)dp, (const double *)dxl, (const double *)dxu, st, cs);
        }
        // (a problem that has just entered the backtracking gets its first point evaluated in the same round)
        launch_dq_residual(h, nprob, m, n, dA, db, gamma, dxnew, dfnew, nullptr, st, CL_BT);
        hipLaunchKernelGGL(k_cls_bt, dim3(nprob), dim3(256), 0, s, m, n, co, dx, dxnew, dfvec, (const double *)dfnew, (const double *)dp,
                           (const double *)dxl, (const double *)dxu, st, cs);
        hipLaunchKernelGGL(k_cls_count, dim3(1), dim3(256), 0, s, nprob, (const LmState *)st, dcounts);
        HIPCHK(h, hipMemcpyAsync(hcounts, dcounts, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        if (echo) HIPCHK(h, hipMemcpyAsync(hcs, cs, sizeof(ClState), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        if (echo && need_jac > 0 && hcs[0].print_due) print_status(hcs[0].pr_iter, hcs[0].pr_neval, hcs[0].pr_njac, hcs[0].pr_xnorm, hcs[0].pr_fnorm);
        need_jac = hcounts[0];
        if (need_jac == 0 && hcounts[1] == 0) break;
    }
    HIPCHK(h, hipMemcpyAsync(hcs, cs, sizeof(ClState) * np, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipGetLastError());
    for (int p = 0; p < nprob; ++p) {
        const ClState &c = hcs[p];
        if (ib) {                                                // :1163-1170 (a non-finite start leaves them zero)
            memset(&ib[p], 0, sizeof ib[p]);
            if (!c.silent) {
                ib[p].iter_count = c.iter; ib[p].fcn_count = c.neval; ib[p].jacobian_count = c.njac;
                ib[p].converge_on_fcn = c.fcnvrg; ib[p].converge_on_chng = c.xcnvrg; ib[p].converge_on_zero_diff = c.gcnvrg;
            }
        }
        if (status) status[p] = (c.silent || c.converged) ? 0 : NLH_CONVERGENCE_ERROR;      // :1173-1175
    }
    return 0;
}

int nlh_dq_cls_solve_batch(nlh_handle *h, const nlh_options *o, double delta0, double stepscale0, const double *xl,
                           const double *xu, int32_t nprob, int32_t m, int32_t n, const double *dA, const double *db,
                           double gamma, double *dx, double *dfvec, nlh_iteration_behavior *ib, int32_t *status)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (!o || n < 1 || m < 1) return NLH_INVALID_INPUT_ERROR;
    if (n > m) return NLH_UNDERDEFINED_PROBLEM_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    static const int cls_host = [] { const char *e = getenv("NLH_CLS_HOSTLOOP"); return e ? atoi(e) : 0; }();
    if (!cls_host)
        return lockstep_slices(nprob, [&](int32_t p0, int32_t cnt) {
            return cls_lockstep(h, o, delta0, stepscale0, xl, xu, cnt, m, n, dA + (size_t)p0 * m * n, db + (size_t)p0 * m, gamma,
                                dx + (size_t)p0 * n, dfvec + (size_t)p0 * m, ib ? ib + p0 : nullptr, status ? status + p0 : nullptr);
        });
    // one problem per call; run_problems deals the problems to worker threads with private handles
    auto solve_one = [&](nlh_handle *h, int p) -> int {
        int rc;
        const size_t mn = (size_t)m * n;
        if ((rc = ensure(h, h->P, sizeof(double) * mn))) return rc;
        if ((rc = ensure(h, h->xdev, sizeof(double) * n))) return rc;
        if ((rc = ensure(h, h->wa4, sizeof(double) * m))) return rc;
        hipStream_t s = h->stream;
        std::vector<double> x(n), f(m);
        const double *A = dA + (size_t)p * mn, *b = db + (size_t)p * m;
        double *dxp = dx + (size_t)p * n, *dfp = dfvec + (size_t)p * m;
        double *dxs = (double *)h->xdev.p, *dfs = (double *)h->wa4.p;
        HIPCHK(h, hipMemcpyAsync(x.data(), dxp, sizeof(double) * n, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        ClsEval ev;
        ev.fcn = [&](const double *xx, double *ff) -> int {
            HIPCHK(h, hipMemcpyAsync(dxs, xx, sizeof(double) * n, hipMemcpyHostToDevice, s));
            launch_dq_residual(h, 1, m, n, A, b, gamma, dxs, dfs, nullptr, nullptr, -1);
            HIPCHK(h, hipMemcpyAsync(ff, dfs, sizeof(double) * m, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipStreamSynchronize(s));
            return 0;
        };
        ev.jac = [&](double *xx, const double *f0, double *dJ) -> int {
            HIPCHK(h, hipMemcpyAsync(dxs, xx, sizeof(double) * n, hipMemcpyHostToDevice, s));
            HIPCHK(h, hipMemcpyAsync(dfs, f0, sizeof(double) * m, hipMemcpyHostToDevice, s));
            launch_dq_panel(h, 1, m, n, A, b, gamma, dxs, dJ, nullptr, -1, dfs);      // fused FD column write
            return 0;
        };
        nlh_iteration_behavior lib;
        memset(&lib, 0, sizeof lib);
        rc = cls_core(h, o, delta0, stepscale0, xl, xu, m, n, ev, x.data(), f.data(), &lib);
        if (rc < 0) return rc;
        if (ib) ib[p] = lib;
        if (status) status[p] = rc;
        HIPCHK(h, hipMemcpyAsync(dxp, x.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(dfp, f.data(), sizeof(double) * m, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipStreamSynchronize(s));
        return 0;
    };
    const int rcb = run_problems(h, nprob, solve_one);
    if (rcb) return rcb;
    HIPCHK(h, hipGetLastError());
    return 0;
}

// fcnnvar_helper%gradient -- fnh_grad_fcn, src/nonlin_multi_var.f90:182-246: the user's gradient routine when there is
// one, otherwise forward differences with h_j = sqrt(eps) |x_j| (sqrt(eps) at x_j = 0), one evaluation per variable in
// ascending order on the calling thread, true division.  The work is n + 1 calls of a host function: nothing here for
// the device; it lives behind the C ABI so that the Fortran shim and nlh_bfgs_solve share one implementation.
int nlh_fd_gradient(int32_t n, nlh_fcnnvar fcn, nlh_gradfcn gradfcn, void *ctx, double *x, const double *fv, double *g)
{
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;
    if (n < 1 || !x || !g) return NLH_INVALID_INPUT_ERROR;
    if (gradfcn) { gradfcn(ctx, n, x, g); return 0; }
    const double f0 = fv ? *fv : fcn(ctx, n, x);
    for (int j = 0; j < n; ++j) {
        const double xj = x[j];
        double step = NLH_SQRT_EPS * fabs(xj);
        if (step == 0.0) step = NLH_SQRT_EPS;
        x[j] = xj + step;
        const double fj = fcn(ctx, n, x);
        x[j] = xj;
        g[j] = (fj - f0) / step;
    }
    return 0;
}

// bfgs%solve -- bfgs_solve, src/nonlin_optimize.f90:557-770; fcnnvar / gradientfcn callbacks flattened to C
int nlh_bfgs_solve(nlh_handle *h, const nlh_options *o, int32_t n, nlh_fcnnvar fcn, nlh_gradfcn gradfcn, void *ctx,
                   double *x, double *fout, nlh_iteration_behavior *ib)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (ib) memset(ib, 0, sizeof *ib);
    if (!fcn) return NLH_UNDEFINED_FUNCTION_ERROR;              // :614
    if (!o || n < 1) return NLH_INVALID_INPUT_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    BfgsEval ev;
    ev.fcn = [&](const double *xx, double *f) -> int { *f = fcn(ctx, n, xx); return 0; };
    ev.grad = [&](double *xx, double fv, double *g) -> int { return nlh_fd_gradient(n, fcn, gradfcn, ctx, xx, &fv, g); };
    int rc = bfgs_core(h, o, n, ev, x, fout, ib);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return NLH_ERR_HIP; }
    return rc;
}

// Device model: minimise f(x) = 0.5 * sum_i r_i(x)^2 of the dense-quadratic residual with bfgs; the
// forward-difference gradient (fnh_grad_fcn) is the residual panel kernel + k_bf_fd_gradient.
// bfgs%solve for a batch of device-model problems (objective 0.5 ||F(x)||^2, forward-difference gradient): the lock-step
// state machine of nlh_kernels_bfgs_batch.h.  A round takes every problem that wants a gradient through the residual
// panel and the differences, the tests and the secant pair, the update of the Cholesky factor (rank-one update +
// downdate, or a refactorisation) and the two triangular solves for the direction to the first trial point of its line
// search; every problem with a trial point gets F evaluated there and one turn of the search.  One 8-byte read-back
// per round.
static int bfgs_lockstep(nlh_handle *h, const nlh_options *o, int32_t nprob, int32_t m, int32_t n, const double *dA,
                         const double *db, double gamma, double *dx, double *hfout, nlh_iteration_behavior *ib,
                         int32_t *status)
{
    int rc;
    if (n > QN_MAX_N) return NLH_ARRAY_SIZE_ERROR;
    const size_t mn = (size_t)m * n, nn = (size_t)n * n, np = (size_t)nprob;
    if ((rc = ensure(h, h->P, sizeof(double) * mn * np))) return rc;
    if ((rc = ensure(h, h->bfB, sizeof(double) * nn * np))) return rc;
    if ((rc = ensure(h, h->bfR, sizeof(double) * nn * np))) return rc;
    if ((rc = ensure(h, h->bfV, sizeof(double) * ((size_t)m + 10 * (size_t)n) * np + sizeof(int32_t) * np + 64))) return rc;
    if ((rc = ensure(h, h->state, sizeof(LmState) * np))) return rc;
    if ((rc = ensure(h, h->misc, sizeof(BfState) * np + 64))) return rc;
    if ((rc = ensure_pinned(h, sizeof(BfState) * np + 64))) return rc;
    double *dP = (double *)h->P.p, *dB = (double *)h->bfB.p, *dR = (double *)h->bfR.p, *q = (double *)h->bfV.p;
    double *dfv = q; q += (size_t)m * np;                        // F at the point evaluated last
    double *dg = q; q += (size_t)n * np;
    double *dgold = q; q += (size_t)n * np;
    double *ddx = q; q += (size_t)n * np;                        // the direction, then the step taken
    double *dy = q; q += (size_t)n * np;
    double *dbdx = q; q += (size_t)n * np;
    double *du = q; q += (size_t)n * np;
    double *dv = q; q += (size_t)n * np;
    double *dc = q; q += (size_t)n * np;
    double *dw = q; q += (size_t)n * np;
    double *dxnew = q; q += (size_t)n * np;
    int32_t *dinfo = (int32_t *)q;
    LmState *st = (LmState *)h->state.p;
    int32_t *dcounts = (int32_t *)h->misc.p;
    BfState *bs = (BfState *)((char *)h->misc.p + 64);
    int32_t *hcounts = (int32_t *)h->pinned;
    BfState *hbs = (BfState *)((char *)h->pinned + 64);
    hipStream_t s = h->stream;
    BfOpts bo;
    bo.xtol = o->xtol; bo.gtol = o->gtol; bo.ls_alpha = o->ls_alpha; bo.ls_factor = o->ls_factor;
    bo.max_evals = o->max_evals; bo.ls_max_evals = o->ls_max_evals; bo.use_line_search = o->use_line_search ? 1 : 0;
    bo.rc_divergent = NLH_DIVERGENT_BEHAVIOR_ERROR; bo.rc_convergence = NLH_CONVERGENCE_ERROR; bo.rc_invalid_op = NLH_INVALID_OPERATION_ERROR;
    bo.pad0 = bo.pad1 = 0;
    const int pb = (nprob + 255) / 256;
    const bool echo = o->print_status && nprob == 1;
    const int bs1 = std::min(1024, ((n + 63) / 64) * 64);
    const size_t bstride = sizeof(BfState) / sizeof(double), istride = sizeof(BfState) / sizeof(int32_t);
    static_assert(sizeof(BfState) % sizeof(double) == 0, "BfState is read through strided double / int pointers");
    const double *fp_all = &bs[0].fp, *temp_all = &bs[0].temp;
    const int32_t *iter_all = &bs[0].iter;
    const LmState *cst = st;

    hipLaunchKernelGGL(k_bfl_reset, dim3(pb), dim3(256), 0, s, nprob, st, bs, dinfo);
    launch_dq_residual(h, nprob, m, n, dA, db, gamma, dx, dfv, nullptr, st, BF_START);           // :633
    hipLaunchKernelGGL(k_bfl_start, dim3(nprob), dim3(256), 0, s, m, (const double *)dfv, st, bs);
    int need_grad = nprob;                                       // upper bound until the first read-back
    // a round costs every live problem an evaluation at least (a trial point, or an iteration's first one)
    const long max_rounds = (long)o->max_evals + (long)o->ls_max_evals + 16;
    for (long round = 0; round < max_rounds; ++round) {
        if (need_grad > 0) {
            // fnh_grad_fcn: n perturbed evaluations, (f_j - f) / h_j (src/nonlin_multi_var.f90:182-246)
            launch_dq_panel(h, nprob, m, n, dA, db, gamma, dx, dP, st, BF_GRAD);
            hipLaunchKernelGGL(k_bf_fd_gradient, dim3((n + 63) / 64, nprob), dim3(64), 0, s, m, n, (const double *)dP, (const double *)dx, 0.0, dg,
                               fp_all, bstride, cst, (int)BF_GRAD);
            hipLaunchKernelGGL(k_bfl_after_grad, dim3(nprob), dim3(256), 0, s, n, bo, (const double *)dx, (const double *)dg, (const double *)dgold,
                               ddx, dy, dxnew, st, bs);
            // :703-712: R = temp I in the first iteration, B = R^T R, B dx
            hipLaunchKernelGGL(k_bf_scaled_identity, dim3((unsigned)((nn + 255) / 256), nprob), dim3(256), 0, s, n, 0.0, dR, temp_all, bstride,
                               iter_all, istride, cst, (int)BF_UPD_A);
            hipLaunchKernelGGL(k_bf_rtr, dim3((n + 255) / 256, n, nprob), dim3(256), 0, s, n, (const double *)dR, dB, cst, (int)BF_UPD_A);
            hipLaunchKernelGGL(k_matvec_cm, dim3((n + 255) / 256, nprob), dim3(256), sizeof(double) * n, s, n, n, (const double *)dB,
                               (const double *)ddx, dbdx, cst, (int)BF_UPD_A);
            hipLaunchKernelGGL(k_bfl_split, dim3(nprob), dim3(256), 0, s, n, (const double *)ddx, (const double *)dbdx, (const double *)dy, du, dv, st, bs);
            // :716-722: R^T R += u u^T, then -= v v^T
            if (n <= 1024) hipLaunchKernelGGL(k_bf_chol_update<1>, dim3(nprob), dim3(bs1), sizeof(double) * 2 * n, s, n, dR, (const double *)du, cst, (int)BF_UPD_RANK);
            else hipLaunchKernelGGL(k_bf_chol_update<4>, dim3(nprob), dim3(1024), sizeof(double) * 2 * n, s, n, dR, (const double *)du, cst, (int)BF_UPD_RANK);
            hipLaunchKernelGGL(k_bf_solve_upper_t, dim3(nprob), dim3(bs1), sizeof(double) * n, s, n, (const double *)dR, dv, cst, (int)BF_UPD_RANK);
            hipLaunchKernelGGL(k_bf_downdate_rot, dim3(nprob), dim3(64), 0, s, n, dv, dc, dinfo, cst, (int)BF_UPD_RANK);
            hipLaunchKernelGGL(k_bf_downdate_apply, dim3((n + 255) / 256, nprob), dim3(256), sizeof(double) * 2 * n, s, n, dR, (const double *)dc,
                               (const double *)dv, (const int *)dinfo, cst, (int)BF_UPD_RANK);
            hipLaunchKernelGGL(k_nt_advance, dim3(pb), dim3(256), 0, s, nprob, st, (int)BF_UPD_RANK, (int)BF_DIR);
            // :724: R = chol(B)
            if (n <= 1024) hipLaunchKernelGGL(k_bf_chol_factor<1>, dim3(nprob), dim3(bs1), sizeof(double) * n, s, n, (const double *)dB, dR, dinfo, cst, (int)BF_UPD_FACTOR);
            else hipLaunchKernelGGL(k_bf_chol_factor<4>, dim3(nprob), dim3(1024), sizeof(double) * n, s, n, (const double *)dB, dR, dinfo, cst, (int)BF_UPD_FACTOR);
            hipLaunchKernelGGL(k_nt_advance, dim3(pb), dim3(256), 0, s, nprob, st, (int)BF_UPD_FACTOR, (int)BF_DIR);
            // :727: dx = -(R^T R)^-1 g
            hipLaunchKernelGGL(k_bfl_neg, dim3((n + 255) / 256, nprob), dim3(256), 0, s, n, (const double *)dg, dw, cst);
            hipLaunchKernelGGL(k_bf_solve_upper_t, dim3(nprob), dim3(bs1), sizeof(double) * n, s, n, (const double *)dR, dw, cst, (int)BF_DIR);
            hipLaunchKernelGGL(k_qn_solve_upper, dim3(nprob), dim3(bs1), sizeof(double) * n, s, n, (const double *)dR, dw, nn, (size_t)n, cst, (int)BF_DIR);
            hipLaunchKernelGGL(k_bfl_dir_done, dim3(nprob), dim3(256), 0, s, n, bo, (const double *)dx, (const double *)dg, ddx, (const double *)dw,
                               dxnew, (const int32_t *)dinfo, st, bs);
        }
        launch_dq_residual(h, nprob, m, n, dA, db, gamma, dxnew, dfv, nullptr, st, BF_TRIAL);
        hipLaunchKernelGGL(k_bfl_trial, dim3(nprob), dim3(256), 0, s, m, n, bo, dx, dxnew, ddx, (const double *)dg, dgold, (const double *)dfv, st, bs);
        hipLaunchKernelGGL(k_bfl_count, dim3(1), dim3(256), 0, s, nprob, cst, dcounts);
        HIPCHK(h, hipMemcpyAsync(hcounts, dcounts, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        if (echo) HIPCHK(h, hipMemcpyAsync(hbs, bs, sizeof(BfState), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        if (echo && need_grad > 0 && hbs[0].print_due) {         // :730-737
            printf(" \n");
            printf("Iteration: %d\n", hbs[0].pr_iter);
            printf("Function Evaluations: %d\n", hbs[0].pr_neval);
            char e1[16], e2[16], e3[16];
            format_e10_3(hbs[0].pr_fp, e1); format_e10_3(hbs[0].pr_xtest, e2); format_e10_3(hbs[0].pr_gtest, e3);
            printf("Function Value: %s\nChange in Variable: %s\nGradient: %s\n", e1, e2, e3);
        }
        need_grad = hcounts[0];
        if (need_grad == 0 && hcounts[1] == 0) break;
    }
    HIPCHK(h, hipMemcpyAsync(hbs, bs, sizeof(BfState) * np, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipGetLastError());
    for (int p = 0; p < nprob; ++p) {
        const BfState &c = hbs[p];
        if (ib) {                                                // :751-759
            ib[p].iter_count = c.iter; ib[p].fcn_count = c.neval; ib[p].jacobian_count = 0; ib[p].gradient_count = c.ngrad;
            ib[p].converge_on_fcn = 0; ib[p].converge_on_chng = c.xcnvrg; ib[p].converge_on_zero_diff = c.gcnvrg;
        }
        const bool finished = c.rc || c.flag || c.xcnvrg || c.gcnvrg;
        if (status) status[p] = c.rc ? c.rc : ((c.flag || !finished) ? NLH_CONVERGENCE_ERROR : 0);   // :765-767
        if (hfout) hfout[p] = c.fp;                              // :762
    }
    return 0;
}

int nlh_dq_bfgs_solve_batch(nlh_handle *h, const nlh_options *o, int32_t nprob, int32_t m, int32_t n, const double *dA,
                            const double *db, double gamma, double *dx, double *hfout, nlh_iteration_behavior *ib,
                            int32_t *status)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (!o || n < 1 || m < 1) return NLH_INVALID_INPUT_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    static const int bfgs_host = [] { const char *e = getenv("NLH_BFGS_HOSTLOOP"); return e ? atoi(e) : 0; }();
    if (!bfgs_host)
        return lockstep_slices(nprob, [&](int32_t p0, int32_t cnt) {
            return bfgs_lockstep(h, o, cnt, m, n, dA + (size_t)p0 * m * n, db + (size_t)p0 * m, gamma, dx + (size_t)p0 * n,
                                 hfout ? hfout + p0 : nullptr, ib ? ib + p0 : nullptr, status ? status + p0 : nullptr);
        });
    // one problem per call; run_problems deals the problems to worker threads with private handles
    auto solve_one = [&](nlh_handle *h, int p) -> int {
        int rc;
        const size_t mn = (size_t)m * n;
        if ((rc = ensure(h, h->P, sizeof(double) * mn))) return rc;
        if ((rc = ensure(h, h->xdev, sizeof(double) * 2 * n))) return rc;
        if ((rc = ensure(h, h->wa4, sizeof(double) * m))) return rc;
        hipStream_t s = h->stream;
        std::vector<double> x(n), f(m);
        const double *A = dA + (size_t)p * mn, *b = db + (size_t)p * m;
        double *dxp = dx + (size_t)p * n;
        double *dxs = (double *)h->xdev.p, *dgs = dxs + n, *dfs = (double *)h->wa4.p;
        HIPCHK(h, hipMemcpyAsync(x.data(), dxp, sizeof(double) * n, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        BfgsEval ev;
        ev.fcn = [&](const double *xx, double *fv) -> int {
            HIPCHK(h, hipMemcpyAsync(dxs, xx, sizeof(double) * n, hipMemcpyHostToDevice, s));
            launch_dq_residual(h, 1, m, n, A, b, gamma, dxs, dfs, nullptr, nullptr, -1);
            HIPCHK(h, hipMemcpyAsync(f.data(), dfs, sizeof(double) * m, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipStreamSynchronize(s));
            *fv = 0.5 * h_dot(m, f.data(), f.data());
            return 0;
        };
        ev.grad = [&](double *xx, double fv, double *g) -> int {
            HIPCHK(h, hipMemcpyAsync(dxs, xx, sizeof(double) * n, hipMemcpyHostToDevice, s));
            launch_dq_panel(h, 1, m, n, A, b, gamma, dxs, (double *)h->P.p, nullptr, -1);
            hipLaunchKernelGGL(k_bf_fd_gradient, dim3((n + 63) / 64), dim3(64), 0, s, m, n, (const double *)h->P.p, dxs, fv, dgs, (const double *)nullptr, (size_t)0, (const LmState *)nullptr, -1);
            HIPCHK(h, hipMemcpyAsync(g, dgs, sizeof(double) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipStreamSynchronize(s));
            return 0;
        };
        nlh_iteration_behavior lib;
        memset(&lib, 0, sizeof lib);
        double fo = 0.0;
        rc = bfgs_core(h, o, n, ev, x.data(), &fo, &lib);
        if (rc < 0) return rc;
        if (ib) ib[p] = lib;
        if (status) status[p] = rc;
        if (hfout) hfout[p] = fo;
        HIPCHK(h, hipMemcpyAsync(dxp, x.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipStreamSynchronize(s));
        return 0;
    };
    const int rcb = run_problems(h, nprob, solve_one);
    if (rcb) return rcb;
    HIPCHK(h, hipGetLastError());
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
};

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
    return model_run(h, md, const_cast<double *>(x), false, f, nullptr, nullptr,
                     [&](nlh_handle *ph, const DqPart &pt, nlh_iteration_behavior *, int32_t *) -> int {
                         launch_dq_residual(ph, pt.cnt, md->m, md->n, pt.dA, pt.db, md->gamma, pt.dx, pt.df, nullptr, nullptr, -1);
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
    return model_run(h, md, x, true, fvec, ib, status,
                     [&](nlh_handle *ph, const DqPart &pt, nlh_iteration_behavior *pib, int32_t *pst) -> int {
                         std::vector<double> fo(pt.cnt, 0.0);
                         const int rc = nlh_dq_bfgs_solve_batch(ph, o, pt.cnt, md->m, md->n, pt.dA, pt.db, md->gamma, pt.dx, fo.data(), pib, pst);
                         if (rc) return rc;
                         if (fout)
                             for (int i = 0; i < pt.cnt; ++i) fout[(size_t)pt.first + (size_t)i * pt.stride] = fo[i];
                         launch_dq_residual(ph, pt.cnt, md->m, md->n, pt.dA, pt.db, md->gamma, pt.dx, pt.df, nullptr, nullptr, -1);
                         return 0;
                     });
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

int nlh_gram(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, const double *dJ, const double *df, double *dG,
             double *dg)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    HIPCHK(h, hipSetDevice(h->device));
    int rc = launch_gram(h, nprob, m, n, dJ, df, dG, df ? dg : nullptr, nullptr, -1);
    if (rc) return rc;
    HIPCHK(h, hipGetLastError());
    return 0;
}

int nlh_chol_factor(nlh_handle *h, int32_t nprob, int32_t n, double *dG, const double *dg, int32_t *dipvt,
                    double *dacnorm, double *dqtf, int32_t *dinfo)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    HIPCHK(h, hipSetDevice(h->device));
    LmVecs v;
    memset(&v, 0, sizeof v);
    v.ipvt = dipvt; v.acnorm = dacnorm; v.qtf = dqtf;
    nlh_options o;
    nlh_default_options(&o);
    {
        Timed t(h, NLH_K_CHOL);
        size_t sh = sizeof(double) * (size_t)(3 * n + 64);
        hipLaunchKernelGGL(k_chol_factor, dim3(nprob), dim3(factor_threads(n)), sh, h->stream, n, dG,
                           (const double *)nullptr, dg, v, (const double *)nullptr, (LmState *)nullptr, dinfo,
                           o.factor, o.gtol, 0.0, 1, -1);
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}

int nlh_qr_factor(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, double *dJ, const double *df,
                  int32_t *dipvt, double *drdiag, double *dacnorm, double *dqtf, double *dwa4)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, h->G, sizeof(double) * (size_t)nprob * n * n))) return rc;
    LmVecs v;
    memset(&v, 0, sizeof v);
    v.ipvt = dipvt; v.acnorm = dacnorm; v.qtf = dqtf; v.rdiag = drdiag;
    {
        Timed t(h, NLH_K_QR);
        size_t sh = sizeof(double) * (size_t)(3 * n + 64);
        hipLaunchKernelGGL(k_qr_factor, dim3(nprob), dim3(1024), sh, h->stream, m, n, dJ, df, (double *)h->G.p, v,
                           dwa4, dwa4, (const double *)nullptr, (LmState *)nullptr, 100.0, 0.0, 1);
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}

// lmfactor + Q^T f in the reference's operation order (the factorisation the exact LM policy runs): nlh_qrx.hip on
// caller-supplied matrices, every problem factored.  dJ: [nprob][n][m] column-major, not modified.
int nlh_lmfactor_exact(nlh_handle *h, int32_t nprob, int32_t m, int32_t n, const double *dJ, const double *df,
                       double *dR, int32_t *dipvt, double *drdiag, double *dacnorm, double *dqtf, double *dwa4)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (nprob <= 0) return 0;
    if (m < n || n < 1) return NLH_INVALID_INPUT_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, h->P, sizeof(double) * qrx_matrix_doubles(nprob, m, n)))) return rc;
    if ((rc = ensure(h, h->qxV, qrx_workspace_bytes(nprob, m, n)))) return rc;
    LmVecs v;
    memset(&v, 0, sizeof v);
    v.ipvt = dipvt; v.acnorm = dacnorm; v.qtf = dqtf; v.rdiag = drdiag;
    qrx_factor(h->stream, nprob, m, n, dJ, (double *)h->P.p, df, dR, v, dwa4, dwa4, (const double *)nullptr, (LmState *)nullptr,
               100.0, 0.0, h->qxV.p, (const QrxTimer *)nullptr, nprob);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// lmpar on caller-supplied factors (parity tests): wraps lmpar_dev.
}  // extern "C"

__global__ void __launch_bounds__(1024)
k_lmpar_standalone(int n, double *Rall, int ldr, const int32_t *ipvt_all, const double *diag_all,
                   const double *qtf_all, const double *delta_all, const double *tailsq_all, double *par_all,
                   double *x_all, double *sdiag_all, double *Wall)
{
    extern __shared__ double smem[];
    const int p = blockIdx.x, tid = threadIdx.x, BS = blockDim.x;
    double *xs = smem, *sdiag = smem + n, *wa1 = smem + 2 * n, *wa2n = smem + 3 * n, *z = smem + 4 * n;
    double *red = smem + 5 * n;
    double *rot = red + 64;
    double par = par_all[p];
    lmpar_dev<false>(n, n, Rall + (size_t)p * ldr * n, ldr, ipvt_all + (size_t)p * n, diag_all + (size_t)p * n,
                     qtf_all + (size_t)p * n, delta_all[p], &par, tailsq_all[p], nullptr, xs, sdiag, wa1, wa2n, z,
                     red, nullptr, Wall + (size_t)p * n * n, rot, 0);
    __syncthreads();
    for (int j = tid; j < n; j += BS) {
        x_all[(size_t)p * n + j] = xs[j];
        sdiag_all[(size_t)p * n + j] = sdiag[j];
    }
    if (tid == 0) par_all[p] = par;
}

extern "C" {

int nlh_lmpar(nlh_handle *h, int32_t nprob, int32_t n, double *dR, int32_t ldr, const int32_t *dipvt,
              const double *ddiag, const double *dqtf, const double *ddelta, const double *dtailsq,
              double *dpar, double *dxstep, double *dsdiag)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    HIPCHK(h, hipSetDevice(h->device));
    int rc = ensure(h, h->misc, sizeof(double) * (size_t)nprob * n * n);
    if (rc) return rc;
    {
        Timed t(h, NLH_K_LMPAR);
        size_t sh = sizeof(double) * (size_t)(6 * n + 72);
        hipLaunchKernelGGL(k_lmpar_standalone, dim3(nprob), dim3(factor_threads(n)), sh, h->stream, n, dR, ldr, dipvt,
                           ddiag, dqtf, ddelta, dtailsq, dpar, dxstep, dsdiag, (double *)h->misc.p);
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}

int nlh_lu_factor(nlh_handle *h, int32_t nprob, int32_t n, double *dA, int32_t *dipvt, int32_t *dinfo)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    HIPCHK(h, hipSetDevice(h->device));
    launch_lu_factor(h, nprob, n, dA, dipvt, dinfo);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int nlh_lu_solve(nlh_handle *h, int32_t nprob, int32_t n, const double *dLU, const int32_t *dipvt, double *db)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    HIPCHK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(k_lu_solve, dim3(nprob), dim3(n >= 96 ? 1024 : 256), sizeof(double) * n, h->stream, n, dLU, dipvt, db,
                       (const LmState *)nullptr, -1);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// qr_factor(b, q = q, r = r) / qr_rank1_update / solve_triangular_system stand-ins (call sites :289, :307, :327).
// dB, dQ column-major [nprob][n][n]; dRt is R ROW-major.
int nlh_qr_factor_full(nlh_handle *h, int32_t nprob, int32_t n, const double *dB, double *dQ, double *dRt)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (n < 1 || nprob < 1) return NLH_INVALID_INPUT_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, h->qnV, sizeof(double) * ((size_t)nprob * (16 * (size_t)n + 8))))) return rc;
    launch_qn_qr(h, nprob, n, dB, dQ, dRt, (double *)h->qnV.p);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int nlh_qr_rank1_update(nlh_handle *h, int32_t nprob, int32_t n, double *dQ, double *dRt, const double *du,
                        const double *dv)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (n < 1 || nprob < 1) return NLH_INVALID_INPUT_ERROR;
    if (n > QN_MAX_N) return NLH_ARRAY_SIZE_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, h->qnV, sizeof(double) * ((size_t)nprob * (16 * (size_t)n + 8))))) return rc;
    launch_qn_update(h, nprob, n, dQ, dRt, du, dv, (double *)h->qnV.p);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int nlh_solve_upper(nlh_handle *h, int32_t nprob, int32_t n, const double *dRt, double *dx)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (n < 1 || nprob < 1) return NLH_INVALID_INPUT_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(k_qn_solve_upper, dim3(nprob), dim3(std::min(1024, ((n + 63) / 64) * 64)), sizeof(double) * n,
                       h->stream, n, dRt, dx, (size_t)n * n, (size_t)n, (const LmState *)nullptr, -1);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// cholesky_rank1_update / cholesky_rank1_downdate stand-ins (call sites src/nonlin_optimize.f90:721-722): in place on the
// row-major upper factor dRt (n x n); du is consumed.  *hinfo = 1 if the downdate would lose positive definiteness.
int nlh_chol_rank1(nlh_handle *h, int32_t n, int32_t downdate, double *dRt, double *du, int32_t *hinfo)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (n < 1) return NLH_INVALID_INPUT_ERROR;
    if (n > QN_MAX_N) return NLH_ARRAY_SIZE_ERROR;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, h->bfV, sizeof(double) * ((size_t)6 * n + 8)))) return rc;
    double *dc = (double *)h->bfV.p;
    int *dinfo = (int *)(dc + n);
    hipStream_t s = h->stream;
    const int bs1 = std::min(1024, ((n + 63) / 64) * 64);
    int info = 0;
    if (!downdate) {
        if (n <= 1024) hipLaunchKernelGGL(k_bf_chol_update<1>, dim3(1), dim3(bs1), sizeof(double) * 2 * n, s, n, dRt, du, (const LmState *)nullptr, -1);
        else hipLaunchKernelGGL(k_bf_chol_update<4>, dim3(1), dim3(1024), sizeof(double) * 2 * n, s, n, dRt, du, (const LmState *)nullptr, -1);
    } else {
        hipLaunchKernelGGL(k_bf_solve_upper_t, dim3(1), dim3(bs1), sizeof(double) * n, s, n, dRt, du, (const LmState *)nullptr, -1);
        hipLaunchKernelGGL(k_bf_downdate_rot, dim3(1), dim3(64), 0, s, n, du, dc, dinfo, (const LmState *)nullptr, -1);
        hipLaunchKernelGGL(k_bf_downdate_apply, dim3((n + 255) / 256), dim3(256), sizeof(double) * 2 * n, s, n, dRt, dc, du, dinfo, (const LmState *)nullptr, -1);
        HIPCHK(h, hipMemcpyAsync(&info, dinfo, sizeof(int), hipMemcpyDeviceToHost, s));
    }
    HIPCHK(h, hipStreamSynchronize(s));
    if (hinfo) *hinfo = info;
    HIPCHK(h, hipGetLastError());
    return 0;
}

// polynomial%fit / fit_thru_zero (src/nonlin_polynomials.f90:146-238) for nprob independent data sets of npts
// points each: Vandermonde panel, Householder QR with y as the extra column, back substitution.
// dx, dy: [nprob][npts] device; dcoef: [nprob][order + 1] device (c0 first; c0 = 0 for thru_zero).
int nlh_poly_fit_batch(nlh_handle *h, int32_t nprob, int32_t npts, int32_t order, int32_t thru_zero, const double *dx,
                       const double *dy, double *dcoef)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (nprob < 1) return 0;
    if (order >= npts || order < 1) return 4;                   // :163-166
    HIPCHK(h, hipSetDevice(h->device));
    const int ncols = thru_zero ? order : order + 1;
    int rc;
    if ((rc = ensure(h, h->W2, sizeof(double) * (size_t)nprob * npts * ncols))) return rc;
    if ((rc = ensure(h, h->qnV, sizeof(double) * ((size_t)nprob * (3 * (size_t)npts + 2 * ncols + 16))))) return rc;
    double *dA = (double *)h->W2.p, *dv = (double *)h->qnV.p;
    double *rhs = dv, *vbuf = dv + (size_t)nprob * npts, *wbuf = vbuf + (size_t)nprob * 2 * npts,
           *st = wbuf + (size_t)nprob * 2 * (ncols + 1);
    hipStream_t s = h->stream;
    hipLaunchKernelGGL(k_vandermonde, dim3((npts + 255) / 256, nprob), dim3(256), 0, s, npts, ncols, thru_zero, dx, dy, dA, rhs);
    hipLaunchKernelGGL(k_qn_col0, dim3((npts + 255) / 256, nprob), dim3(256), 0, s, npts, ncols, dA, vbuf);
    launch_house_steps(h, nprob, npts, ncols, 1, dA, rhs, vbuf, wbuf, st);
    hipLaunchKernelGGL(k_qn_solve_upper, dim3(nprob), dim3(64), sizeof(double) * ncols, s, ncols, dA, rhs,
                       (size_t)npts * ncols, (size_t)npts, (const LmState *)nullptr, -1);
    if (thru_zero) HIPCHK(h, hipMemsetAsync(dcoef, 0, sizeof(double) * (size_t)nprob * (order + 1), s));
    HIPCHK(h, hipMemcpy2DAsync(dcoef + (thru_zero ? 1 : 0), sizeof(double) * (order + 1), rhs, sizeof(double) * npts,
                               sizeof(double) * ncols, nprob, hipMemcpyDeviceToDevice, s));
    HIPCHK(h, hipGetLastError());
    return 0;
}

// Host-array front end for one data set (what polynomial%fit marshals to).
int nlh_poly_fit(nlh_handle *h, int32_t npts, int32_t order, int32_t thru_zero, const double *x, const double *y, double *coef)
{
    if (!h) return NLH_ERR_BAD_HANDLE;
    if (order >= npts || order < 1) return 4;
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, h->xdev, sizeof(double) * ((size_t)2 * npts + order + 1)))) return rc;
    double *dxv = (double *)h->xdev.p, *dyv = dxv + npts, *dc = dyv + npts;
    hipStream_t s = h->stream;
    HIPCHK(h, hipMemcpyAsync(dxv, x, sizeof(double) * npts, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(dyv, y, sizeof(double) * npts, hipMemcpyHostToDevice, s));
    if ((rc = nlh_poly_fit_batch(h, 1, npts, order, thru_zero, dxv, dyv, dc))) return rc;
    HIPCHK(h, hipMemcpyAsync(coef, dc, sizeof(double) * (order + 1), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    return 0;
}

}  // extern "C"
