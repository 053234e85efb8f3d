// nlh_kernels_newton.h -- ns_solve (src/nonlin_solve.f90:452-638) as a LOCK-STEP BATCH: every problem carries a stage, every
// kernel of a round is launched over all problems and returns at once for problems in another stage (as the batched
// Levenberg-Marquardt driver does), and the only host synchronisation is one 8-byte read-back per round (how many problems
// want a Jacobian, how many a trial evaluation).  The O(n) logic of the reference -- ls_search_mimo
// (src/nonlin_linesearch.f90:152-326), min_backtrack_search (:495-551), limit_search_vector (:554-572), test_convergence
// (src/nonlin_helper.f90:36-124) -- runs here, one workgroup per problem, in the reference's operation order: every dot
// product is ONE ordered chain of adds (it runs down the lanes of a wave, sixteen terms per lane), NORM2 is the flang
// runtime's algorithm, maxima are exact in any order.  Given bit-identical J, LU and residual kernels, every accept /
// backtrack decision and every iterate is bit-identical to the host loop (newton_core) and to the CPU path.
#pragma once
#include "nlh_common.h"

enum NtStage : int32_t {
    NT_START = 20,      // F(x0) evaluated, start-up logic due (:538-553)
    NT_NEED_JAC = 21,   // iteration head: Jacobian, gradient, LU, direction, then the step set-up (:556-589);
                        // quasi-Newton: an iteration that starts from a fresh Jacobian and its QR factors (:284-292)
    NT_TRIAL = 22,      // x holds a trial point: F(x) due, then the search / convergence logic
    NT_UPDATE = 23,     // quasi-Newton: an iteration that starts with Broyden's rank-one update of B, Q, R (:294-310)
    NT_DIR = 24,        // quasi-Newton: factors current, gradient / step / search set-up due (:313-351)
    NT_DONE = ST_DONE
};

struct NtState {
    double f, fold, stpmax, xnorm, fnorm;
    double alam, alam1, f1, slope, alamin;
    int32_t iter, neval, njac;
    int32_t ls_iter, ls_neval;        // ls_search_mimo's own counters (reset by every search)
    int32_t fcnvrg, xcnvrg, gcnvrg;
    int32_t flag;                     // max_evals reached (:616-619): reported as NL_CONVERGENCE_ERROR
    int32_t rc;                       // code of an `error stop` inside the iteration (0: none)
    int32_t print_due;                // the reference would print its status block now (:611-613)
    int32_t restart;                  // quasi-Newton: this iteration started from a fresh Jacobian
    int32_t jcount;                   // quasi-Newton: updates since the last fresh Jacobian
    int32_t pad;
};

struct NtOpts {
    double ftol, xtol, gtol, ls_alpha, ls_factor;
    int32_t max_evals, ls_max_evals, use_line_search;
    int32_t broyden;                  // 0: ns_solve; 1: qns_solve (src/nonlin_solve.f90:156-427) on the same search / test kernels
    int32_t jdelta, pad;              // quasi-Newton: iterations between fresh Jacobians
};

// min_backtrack_search, src/nonlin_linesearch.f90:495-551 (host and device: the same expressions)
__host__ __device__ inline double nlh_min_backtrack_search(int mode, double f0, double f, double f1, double alam, double alam1,
                                                         double slope)
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

#define NT_CHUNK 1024                  // terms per pass of the ordered sum (16 per lane of one wave)

// s = sum_i term(i), i ascending, one add per term starting from +0.0: the chain a sequential loop executes.  All
// threads of the workgroup form the terms (buf: NT_CHUNK doubles of LDS), wave 0 adds them: lane l holds terms
// 16 l .. 16 l + 15, the running sum is handed down the lanes (one DPP shift per lane) and each lane adds its own terms
// in order; slots past the last term hold +0.0 (s + 0.0 == s: s is never -0.0, the chain starts from +0.0).
template <typename Term>
__device__ double nt_ordered_sum(int n, Term term, double *buf, double *xch)
{
    const int tid = threadIdx.x, BS = blockDim.x, lane = tid & 63, wid = tid >> 6;
    double s = 0.0;
    for (int base = 0; base < n; base += NT_CHUNK) {
        const int cl = min(NT_CHUNK, n - base);
        __syncthreads();
        for (int i = tid; i < NT_CHUNK; i += BS) buf[i] = (i < cl) ? term(base + i) : 0.0;
        __syncthreads();
        if (wid == 0) {
            double d[16];
            const double2 *mine = reinterpret_cast<const double2 *>(buf + lane * 16);
#pragma unroll
            for (int u = 0; u < 8; ++u) { const double2 v2 = mine[u]; d[2 * u] = v2.x; d[2 * u + 1] = v2.y; }
            const int nl = (cl + 15) >> 4;
            double t = s;
#pragma unroll 1
            for (int l = 0; l < nl; ++l) {
                if (l > 0) t = nlh_wave_shr1(t);
#pragma unroll
                for (int u = 0; u < 16; ++u) t = t + d[u];
            }
            const int lo = __builtin_amdgcn_readlane(__double2loint(t), nl - 1);
            const int hi = __builtin_amdgcn_readlane(__double2hiint(t), nl - 1);
            if (lane == 0) xch[0] = __hiloint2double(hi, lo);
        }
        __syncthreads();
        s = xch[0];
    }
    return s;
}

// max_i term(i) (exact in any order), broadcast
template <typename Term>
__device__ double nt_block_max(int n, Term term, double *red)
{
    const int tid = threadIdx.x, BS = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = (BS + 63) >> 6;
    double v = 0.0;
    for (int i = tid; i < n; i += BS) v = fmax(term(i), v);
    v = wave_reduce_max(v);
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    double r = red[0];
    for (int w = 1; w < nw; ++w) r = fmax(r, red[w]);
    __syncthreads();
    return r;
}

static __global__ void __launch_bounds__(256)
k_nt_reset(int nprob, LmState *__restrict__ st, NtState *__restrict__ ns)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nprob) return;
    NtState z;
    z.f = z.fold = z.stpmax = z.xnorm = z.fnorm = 0.0;
    z.alam = z.alam1 = z.f1 = z.slope = z.alamin = 0.0;
    z.iter = z.neval = z.njac = z.ls_iter = z.ls_neval = 0;
    z.fcnvrg = z.xcnvrg = z.gcnvrg = z.flag = z.rc = z.print_due = z.pad = 0;
    z.restart = 1; z.jcount = 0;
    ns[p] = z;
    st[p].stage = NT_START;
}

// :538-553 after F(x0): f = 0.5 F.F, the start-point convergence test, stpmax
static __global__ void __launch_bounds__(256)
k_nt_start(int n, NtOpts o, const double *__restrict__ xall, const double *__restrict__ fall, LmState *__restrict__ st,
           NtState *__restrict__ ns)
{
    __shared__ __attribute__((aligned(16))) double buf[NT_CHUNK];
    __shared__ double xch[2], red[8];
    __shared__ double scratch[3 * NLH_NCH + 8];
    const int p = blockIdx.x;
    if (st[p].stage != NT_START) return;
    const double *x = xall + (size_t)p * n, *fv = fall + (size_t)p * n;
    const double f = 0.5 * nt_ordered_sum(n, [&](int i) { return fv[i] * fv[i]; }, buf, xch);
    const double test = nt_block_max(n, [&](int i) { return fabs(fv[i]); }, red);
    const double xn = norm2_flang_block([&](int i) { return x[i]; }, n, scratch);
    if (threadIdx.x == 0) {
        NtState *s = ns + p;
        s->f = f;
        s->neval = 1;
        if (test < o.ftol) {
            s->fcnvrg = 1;
            st[p].stage = NT_DONE;
        } else {
            s->stpmax = 100.0 * fmax(xn, (double)n);
            st[p].stage = NT_NEED_JAC;
        }
    }
}

// rhs = -F(x) (:577), elementwise
static __global__ void __launch_bounds__(256)
k_nt_rhs(int n, const double *__restrict__ fall, double *__restrict__ rall, const LmState *__restrict__ st)
{
    const int p = blockIdx.y;
    if (st[p].stage != NT_NEED_JAC) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rall[(size_t)p * n + i] = -fall[(size_t)p * n + i];
}

// quasi-Newton, an update iteration's head (:294-296): df = F(x) - F(xold), dx = x - xold, x2 = dx.dx (ordered)
static __global__ void __launch_bounds__(256)
k_qn_prep(int n, const double *__restrict__ xall, const double *__restrict__ xoldall, const double *__restrict__ fall,
          const double *__restrict__ fvoldall, double *__restrict__ ddxall, double *__restrict__ ddfall,
          double *__restrict__ x2all, const LmState *__restrict__ st)
{
    __shared__ __attribute__((aligned(16))) double buf[NT_CHUNK];
    __shared__ double xch[2];
    const int p = blockIdx.x;
    if (st[p].stage != NT_UPDATE) return;
    const int tid = threadIdx.x, BS = blockDim.x;
    const double *x = xall + (size_t)p * n, *xold = xoldall + (size_t)p * n, *fv = fall + (size_t)p * n, *fvold = fvoldall + (size_t)p * n;
    double *ddx = ddxall + (size_t)p * n, *ddf = ddfall + (size_t)p * n;
    for (int i = tid; i < n; i += BS) { ddf[i] = fv[i] - fvold[i]; ddx[i] = x[i] - xold[i]; }
    const double x2 = nt_ordered_sum(n, [&](int i) { return ddx[i] * ddx[i]; }, buf, xch);
    if (tid == 0) x2all[p] = x2;
}

static __global__ void __launch_bounds__(256)
k_nt_advance(int nprob, LmState *__restrict__ st, int from, int to)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nprob && st[p].stage == from) st[p].stage = to;
}

// :573-589 after the LU solve (quasi-Newton: :316-351 after grad = B^T F and step = -R^-1 Q^T F): xold, fold, the
// step-length guards, the set-up of ls_search_mimo (:249-265) and the first trial point
static __global__ void __launch_bounds__(256)
k_nt_step_begin(int n, NtOpts o, int want, double *__restrict__ xall, double *__restrict__ xoldall, double *__restrict__ dirall,
                const double *__restrict__ gradall, const double *__restrict__ fall, double *__restrict__ fvoldall,
                LmState *__restrict__ st, NtState *__restrict__ ns)
{
    __shared__ __attribute__((aligned(16))) double buf[NT_CHUNK];
    __shared__ double xch[2], red[8];
    __shared__ double scratch[3 * NLH_NCH + 8];
    const int p = blockIdx.x;
    if (st[p].stage != want) return;
    const int tid = threadIdx.x, BS = blockDim.x;
    double *x = xall + (size_t)p * n, *xold = xoldall + (size_t)p * n, *dir = dirall + (size_t)p * n;
    const double *grad = gradall + (size_t)p * n;
    NtState *s = ns + p;
    const double stpmax = s->stpmax;
    // the counters of this iteration's head: iter (loop top); Newton: a Jacobian every iteration; quasi-Newton: a
    // Jacobian when the iteration restarted (:284-292), one more update since the last one otherwise (:310)
    const int njac_inc = o.broyden ? (s->restart ? 1 : 0) : 1;
    const int jcount = o.broyden ? (s->restart ? 0 : s->jcount + 1) : 0;
    for (int i = tid; i < n; i += BS) xold[i] = x[i];                 // :573-574
    if (o.broyden) {
        const double *fv = fall + (size_t)p * n;
        double *fvold = fvoldall + (size_t)p * n;
        for (int i = tid; i < n; i += BS) fvold[i] = fv[i];           // :317
        const double temp = nt_ordered_sum(n, [&](int i) { return grad[i] * dir[i]; }, buf, xch);   // :332
        if (temp >= 0.0) {                                            // :333-339: not a descent direction: start over
            if (tid == 0) {
                s->iter += 1; s->njac += njac_inc; s->jcount = jcount; s->fold = s->f;
                s->restart = 1; s->print_due = 1;
                if (s->iter > 10 * o.max_evals + 100) { s->flag = 1; st[p].stage = NT_DONE; }     // the reference would spin here
                else st[p].stage = NT_NEED_JAC;
            }
            return;
        }
    }
    if (o.use_line_search) {
        const double temp = nt_ordered_sum(n, [&](int i) { return dir[i] * dir[i]; }, buf, xch);   // :581 (squared length, kept)
        if (temp > stpmax) {
            const double sc = stpmax / temp;
            __syncthreads();
            for (int i = tid; i < n; i += BS) dir[i] = dir[i] * sc;
            __syncthreads();
        }
        const double mag = norm2_flang_block([&](int i) { return dir[i]; }, n, scratch);            // limit_search_vector
        if (mag != 0.0 && mag > stpmax) {
            const double sc = stpmax / mag;
            __syncthreads();
            for (int i = tid; i < n; i += BS) dir[i] = sc * dir[i];
            __syncthreads();
        }
        const double slope = nt_ordered_sum(n, [&](int i) { return grad[i] * dir[i]; }, buf, xch); // linesearch :249
        if (slope >= 0.0) {                                           // :250-253: error stop
            if (tid == 0) { s->iter += 1; s->njac += njac_inc; s->jcount = jcount; s->fold = s->f; s->rc = 206; st[p].stage = NT_DONE; }
            return;
        }
        const double test = nt_block_max(n, [&](int i) { return fabs(dir[i]) / fmax(fabs(xold[i]), 1.0); }, red);
        const double alam = 1.0;
        for (int i = tid; i < n; i += BS) x[i] = xold[i] + alam * dir[i];
        if (tid == 0) {
            s->iter += 1; s->njac += njac_inc; s->jcount = jcount; s->fold = s->f; s->print_due = 0;
            s->slope = slope;
            s->alamin = (2.0 * NLH_EPS) / test;
            s->alam = alam; s->alam1 = 0.0; s->f1 = 0.0;
            s->ls_iter = 0; s->ls_neval = 0;
            st[p].stage = NT_TRIAL;
        }
    } else {                                                          // :591-595
        for (int i = tid; i < n; i += BS) x[i] = x[i] + dir[i];
        if (tid == 0) { s->iter += 1; s->njac += njac_inc; s->jcount = jcount; s->fold = s->f; s->print_due = 0; st[p].stage = NT_TRIAL; }
    }
}

// F(x) at the trial point is in fvec: one turn of ls_search_mimo's loop (:266-310); on acceptance test_convergence and the
// end of the outer iteration (:599-619)
static __global__ void __launch_bounds__(256)
k_nt_trial(int n, NtOpts o, double *__restrict__ xall, const double *__restrict__ xoldall, const double *__restrict__ dirall,
           const double *__restrict__ gradall, const double *__restrict__ fall, LmState *__restrict__ st,
           NtState *__restrict__ ns)
{
    __shared__ __attribute__((aligned(16))) double buf[NT_CHUNK];
    __shared__ double xch[2], red[8];
    const int p = blockIdx.x;
    if (st[p].stage != NT_TRIAL) return;
    const int tid = threadIdx.x, BS = blockDim.x;
    double *x = xall + (size_t)p * n;
    const double *xold = xoldall + (size_t)p * n, *dir = dirall + (size_t)p * n, *grad = gradall + (size_t)p * n;
    const double *fv = fall + (size_t)p * n;
    NtState *s = ns + p;
    // (every thread takes its copy of the search state before the first barrier: thread 0 rewrites it below, and a wave
    // that is behind must not see the new values)
    const NtState q = *s;
    const double f = 0.5 * nt_ordered_sum(n, [&](int i) { return fv[i] * fv[i]; }, buf, xch);
    int neval = q.neval;
    if (o.use_line_search) {
        const int lsn = q.ls_neval + 1, lsi = q.ls_iter + 1;
        const double alam = q.alam, fold = q.fold, slope = q.slope;
        bool accept = false;
        if (alam < q.alamin) {                                        // :275-287
            const double sq = nt_ordered_sum(n, [&](int i) { const double d = x[i] - xold[i]; return d * d; }, buf, xch);
            if (sqrt(sq) == 0.0) {
                if (tid == 0) { s->f = f; s->neval = neval + lsn; s->ls_neval = lsn; s->ls_iter = lsi; s->rc = 106; st[p].stage = NT_DONE; }
                return;
            }
            __syncthreads();
            for (int i = tid; i < n; i += BS) x[i] = xold[i];
            __syncthreads();
            accept = true;
        } else if (f <= fold + o.ls_alpha * alam * slope) {           // :288-291
            accept = true;
        }
        if (!accept) {
            const double tmplam = nlh_min_backtrack_search(lsi, fold, f, q.f1, alam, q.alam1, slope);
            const double nalam = fmax(tmplam, o.ls_factor * alam);    // :300-302
            if (lsn >= o.ls_max_evals) {                              // :305-309: error stop
                if (tid == 0) {
                    s->f = f; s->neval = neval + lsn; s->ls_neval = lsn; s->ls_iter = lsi;
                    s->alam1 = alam; s->f1 = f; s->alam = nalam;
                    s->rc = 106; st[p].stage = NT_DONE;
                }
                return;
            }
            for (int i = tid; i < n; i += BS) x[i] = xold[i] + nalam * dir[i];
            if (tid == 0) { s->alam1 = alam; s->f1 = f; s->alam = nalam; s->ls_neval = lsn; s->ls_iter = lsi; }
            return;                                                   // stays NT_TRIAL
        }
        neval += lsn;
        if (tid == 0) { s->ls_neval = lsn; s->ls_iter = lsi; }
    } else {
        neval += 1;
    }
    // test_convergence (src/nonlin_helper.f90:36-124), early returns in the reference's order
    int fc = 0, xc = 0, gc = 0, check = 0;
    double xnorm = 0.0;
    const double fnorm = nt_block_max(n, [&](int i) { return fabs(fv[i]); }, red);
    if (fnorm < o.ftol) {
        fc = 1; check = 1;
    } else {
        xnorm = nt_block_max(n, [&](int i) { return fabs(x[i] - xold[i]) / fmax(fabs(x[i]), 1.0); }, red);
        if (xnorm < o.xtol) {
            xc = 1; check = 1;
        } else if (!o.broyden) {
            const double den = fmax(f, 0.5 * (double)n);
            const double tg = nt_block_max(n, [&](int i) { return fabs(grad[i]) * fmax(fabs(x[i]), 1.0) / den; }, red);
            if (tg < o.gtol) gc = 1;
        }                                                             // (qns_solve tests the gradient only when the search reports a
    }                                                                 //  zero slope, which ls_search_mimo never does: :360-367)
    if (tid == 0) {
        s->f = f; s->neval = neval;
        s->fcnvrg = fc; s->xcnvrg = xc; s->gcnvrg = gc;
        s->xnorm = xnorm; s->fnorm = fnorm;
        if (check) {
            st[p].stage = NT_DONE;
        } else if (gc) {
            s->rc = 207; st[p].stage = NT_DONE;                       // :604-608
        } else {
            s->print_due = 1;                                         // :611-613 / :398-400
            const int restart = o.broyden ? (q.jcount >= o.jdelta ? 1 : 0) : 1;      // :368-391
            s->restart = restart;
            if (neval >= o.max_evals) { s->flag = 1; st[p].stage = NT_DONE; }   // :616-619 / :403-406
            else st[p].stage = restart ? NT_NEED_JAC : NT_UPDATE;
        }
    }
}

static __global__ void __launch_bounds__(256)
k_nt_count(int nprob, const LmState *__restrict__ st, int32_t *__restrict__ counts)
{
    __shared__ int c[3];
    if (threadIdx.x < 3) c[threadIdx.x] = 0;
    __syncthreads();
    int a = 0, b = 0, u = 0;
    for (int p = threadIdx.x; p < nprob; p += blockDim.x) {
        const int sg = st[p].stage;
        a += (sg == NT_NEED_JAC);
        b += (sg == NT_TRIAL);
        u += (sg == NT_UPDATE);
    }
    if (a) atomicAdd(&c[0], a);
    if (b) atomicAdd(&c[1], b);
    if (u) atomicAdd(&c[2], u);
    __syncthreads();
    if (threadIdx.x < 3) counts[threadIdx.x] = c[threadIdx.x];
}
