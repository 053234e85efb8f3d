// nlh_kernels_bfgs_batch.h -- bfgs_solve (src/nonlin_optimize.f90:557-770) as a LOCK-STEP BATCH, built like
// nlh_kernels_newton.h: every problem carries a stage, every kernel of a round is launched over all problems and returns
// at once for problems in another stage, one small read-back per round.  The O(n) logic of the reference -- the start
// (:633-642), limit_search_vector and ls_search_miso (src/nonlin_linesearch.f90:329-492, :554-572), the step bookkeeping
// and the convergence tests (:672-696), the secant pair and the choice between rank-one update / downdate and
// refactorisation (:699-724) -- runs here, one workgroup per problem, in the reference's operation order: every dot
// product is ONE ordered chain of adds (nt_ordered_sum), NORM2 is the flang runtime's algorithm, maxima are exact in any
// order.  The dense kernels (R^T R, B dx, the Cholesky update / downdate / factorisation, the two triangular solves) are
// the ones of nlh_kernels_bfgs.h with a problem index and a stage gate.  Given those, every decision and every iterate
// is bit-identical to the host loop (bfgs_core) and to the CPU path.
#pragma once
#include <cfloat>
#include "nlh_kernels_newton.h"

enum BfStage : int32_t {
    BF_START = 40,       // F(x0) evaluated: the objective value due (:633)
    BF_GRAD = 41,        // x and f(x) current: the gradient due (residual panel + differences), then the start-up (:634-642)
                         // or the step bookkeeping and the convergence tests (:672-706)
    BF_UPD_A = 42,       // secant pair ready: R (first iteration: a scaled identity), B = R^T R, B dx due, then the choice (:703-715)
    BF_UPD_RANK = 43,    // rank-one update with u, downdate with v (:716-722)
    BF_UPD_FACTOR = 44,  // R = chol(B) (:724)
    BF_DIR = 45,         // dx = -(R^T R)^-1 g due (:727), then the end of the iteration and the set-up of the next search
    BF_TRIAL = 46,       // xnew holds a trial point: F(xnew) due, then one turn of ls_search_miso (or the plain step)
    BF_DONE = ST_DONE
};

struct BfState {
    double fp;                        // f at x (first member: the gradient kernel reads it through a strided pointer)
    double temp;                      // the first iteration's scale of the identity (:704)
    double stpmax, xtest, gtest, ydx;
    double alam, alam1, f1, slope, alamin;
    double pr_fp, pr_xtest, pr_gtest; // the status block (:730-737), printed by the host for a lone solve
    int32_t iter;                     // (second int-aligned group; iter is read through a strided pointer too)
    int32_t neval, ngrad;
    int32_t ls_iter, ls_neval;
    int32_t xcnvrg, gcnvrg, flag, rc;
    int32_t print_due, pr_iter, pr_neval;
};

struct BfOpts {
    double xtol, gtol, ls_alpha, ls_factor;
    int32_t max_evals, ls_max_evals, use_line_search;
    int32_t rc_divergent, rc_convergence, rc_invalid_op, pad0, pad1;      // the C ABI's codes for the three error stops
};

static __global__ void __launch_bounds__(256)
k_bfl_reset(int nprob, LmState *__restrict__ st, BfState *__restrict__ bs, int32_t *__restrict__ info)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nprob) return;
    BfState z;
    z.fp = z.temp = z.stpmax = z.xtest = z.gtest = z.ydx = 0.0;
    z.alam = z.alam1 = z.f1 = z.slope = z.alamin = 0.0;
    z.pr_fp = z.pr_xtest = z.pr_gtest = 0.0;
    z.iter = z.neval = z.ngrad = z.ls_iter = z.ls_neval = 0;
    z.xcnvrg = z.gcnvrg = z.flag = z.rc = z.print_due = z.pr_iter = z.pr_neval = 0;
    bs[p] = z;
    info[p] = 0;
    st[p].stage = BF_START;
}

// the objective at the current point of every problem, contiguous (f0 of the forward differences of a user's fcnnvar)
static __global__ void __launch_bounds__(256)
k_bfl_gather_fp(int nprob, const BfState *__restrict__ bs, double *__restrict__ out)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < nprob) out[p] = bs[p].fp;
}

// :633 after F(x0): f = 0.5 F.F
static __global__ void __launch_bounds__(256)
k_bfl_start(int m, const double *__restrict__ fall, LmState *__restrict__ st, BfState *__restrict__ bs, int scalar)
{
    __shared__ __attribute__((aligned(16))) double buf[NT_CHUNK];
    __shared__ double xch[2];
    const int p = blockIdx.x;
    if (st[p].stage != BF_START) return;
    const double *fv = fall + (size_t)p * m;
    // scalar: the user's fcnnvar itself (a launcher with m = 1), not 0.5 F.F of a residual family
    const double f = scalar ? fv[0] : 0.5 * nt_ordered_sum(m, [&](int i) { return fv[i] * fv[i]; }, buf, xch);
    if (threadIdx.x == 0) { bs[p].fp = f; bs[p].neval = 1; st[p].stage = BF_GRAD; }
}

// The head of an iteration from its direction on (:659-669): limit_search_vector, then the set-up of ls_search_miso
// (:380-400) and its first trial point -- or, without a line search, the full step.  Called by every thread of the
// problem's workgroup; dir is the problem's dx.
__device__ void bfl_search_begin(int n, const BfOpts &o, const double *x, const double *g, double *dir, double *xnew,
                                 LmState *sg, BfState *s, double *buf, double *xch, double *red, double *scratch)
{
    const int tid = threadIdx.x, BS = blockDim.x;
    if (!o.use_line_search) {
        for (int i = tid; i < n; i += BS) xnew[i] = x[i] + dir[i];
        if (tid == 0) sg->stage = BF_TRIAL;
        return;
    }
    const double stpmax = s->stpmax;
    const double mag = norm2_flang_block([&](int i) { return dir[i]; }, n, scratch);    // limit_search_vector
    if (mag != 0.0 && mag > stpmax) {
        const double sc = stpmax / mag;
        __syncthreads();
        for (int i = tid; i < n; i += BS) dir[i] = sc * dir[i];
        __syncthreads();
    }
    const double slope = nt_ordered_sum(n, [&](int i) { return g[i] * dir[i]; }, buf, xch);
    if (slope >= 0.0) {                                                 // not a descent direction: error stop
        if (tid == 0) { s->rc = o.rc_divergent; sg->stage = BF_DONE; }
        return;
    }
    const double test = nt_block_max(n, [&](int i) { return fabs(dir[i]) / fmax(fabs(x[i]), 1.0); }, red);
    for (int i = tid; i < n; i += BS) xnew[i] = x[i] + 1.0 * dir[i];
    if (tid == 0) {
        s->slope = slope; s->alamin = (2.0 * DBL_EPSILON) / test; s->alam = 1.0; s->alam1 = 0.0; s->f1 = 0.0;
        s->ls_iter = 0; s->ls_neval = 0;
        sg->stage = BF_TRIAL;
    }
}

// After the gradient: the start-up (:639-656) the first time, afterwards the tests and the secant pair (:681-706).
// dxs: the step just taken (x - xold), y: g - gold.
static __global__ void __launch_bounds__(256)
k_bfl_after_grad(int n, BfOpts o, const double *__restrict__ xall, const double *__restrict__ gall, const double *__restrict__ goldall,
                 double *__restrict__ dxall, double *__restrict__ yall, double *__restrict__ xnewall, LmState *__restrict__ st,
                 BfState *__restrict__ bs)
{
    __shared__ double scratch[3 * NLH_NCH + 8];
    __shared__ __attribute__((aligned(16))) double buf[NT_CHUNK];
    __shared__ double xch[2], red[16];
    const int p = blockIdx.x, tid = threadIdx.x, BS = blockDim.x;
    if (st[p].stage != BF_GRAD) return;
    BfState *s = bs + p;
    const double *x = xall + (size_t)p * n, *g = gall + (size_t)p * n, *gold = goldall + (size_t)p * n;
    double *dx = dxall + (size_t)p * n, *y = yall + (size_t)p * n, *xnew = xnewall + (size_t)p * n;
    const int ngrad = s->ngrad;
    const double gtest = norm2_flang_block([&](int i) { return g[i]; }, n, scratch);
    if (ngrad == 0) {                                                   // :634-656
        if (gtest < o.gtol) {
            if (tid == 0) { s->ngrad = 1; s->gtest = gtest; s->gcnvrg = 1; st[p].stage = BF_DONE; }
            return;
        }
        const double xn = norm2_flang_block([&](int i) { return x[i]; }, n, scratch);
        for (int i = tid; i < n; i += BS) dx[i] = -g[i];
        __syncthreads();
        if (tid == 0) { s->ngrad = 1; s->gtest = gtest; s->iter = 1; s->stpmax = 100.0 * fmax(xn, (double)n); }
        __syncthreads();
        bfl_search_begin(n, o, x, g, dx, xnew, st + p, s, buf, xch, red, scratch);
        return;
    }
    const double xtest = nt_block_max(n, [&](int i) { return fabs(dx[i]) / fmax(fabs(x[i]), 1.0); }, red);   // :681-689
    if (xtest < o.xtol) {
        if (tid == 0) { s->ngrad = ngrad + 1; s->xtest = xtest; s->xcnvrg = 1; st[p].stage = BF_DONE; }
        return;
    }
    if (gtest < o.gtol) {                                               // :692-696
        if (tid == 0) { s->ngrad = ngrad + 1; s->xtest = xtest; s->gtest = gtest; s->gcnvrg = 1; st[p].stage = BF_DONE; }
        return;
    }
    for (int i = tid; i < n; i += BS) y[i] = g[i] - gold[i];            // :699-700
    __syncthreads();
    const double ydx = nt_ordered_sum(n, [&](int i) { return y[i] * dx[i]; }, buf, xch);
    double temp = 0.0;
    if (s->iter == 1) temp = sqrt(nt_ordered_sum(n, [&](int i) { return y[i] * y[i]; }, buf, xch) / ydx);      // :703-704
    if (tid == 0) {
        s->ngrad = ngrad + 1; s->xtest = xtest; s->gtest = gtest; s->ydx = ydx; s->temp = temp;
        st[p].stage = BF_UPD_A;
    }
}

// Given B dx: the rank-one pair u = y / sqrt(y.dx), v = B dx / sqrt(dx.B dx) (:716-720), or a refactorisation (:724).
static __global__ void __launch_bounds__(256)
k_bfl_split(int n, const double *__restrict__ dxall, const double *__restrict__ bdxall, const double *__restrict__ yall,
            double *__restrict__ uall, double *__restrict__ vall, LmState *__restrict__ st, BfState *__restrict__ bs)
{
    __shared__ __attribute__((aligned(16))) double buf[NT_CHUNK];
    __shared__ double xch[2];
    const int p = blockIdx.x, tid = threadIdx.x, BS = blockDim.x;
    if (st[p].stage != BF_UPD_A) return;
    const BfState *s = bs + p;
    if (!(s->ydx > 1.0e-10 && s->iter > 1)) {
        if (tid == 0) st[p].stage = BF_UPD_FACTOR;
        return;
    }
    const double *dx = dxall + (size_t)p * n, *bdx = bdxall + (size_t)p * n, *y = yall + (size_t)p * n;
    double *u = uall + (size_t)p * n, *v = vall + (size_t)p * n;
    const double s1 = sqrt(s->ydx), s2 = sqrt(nt_ordered_sum(n, [&](int i) { return dx[i] * bdx[i]; }, buf, xch));
    for (int i = tid; i < n; i += BS) { u[i] = y[i] / s1; v[i] = bdx[i] / s2; }
    if (tid == 0) st[p].stage = BF_UPD_RANK;
}

// right-hand side of the direction solve (:727)
static __global__ void __launch_bounds__(256)
k_bfl_neg(int n, const double *__restrict__ gall, double *__restrict__ wall, const LmState *__restrict__ st)
{
    const int p = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (st[p].stage != BF_DIR) return;
    if (i < n) wall[(size_t)p * n + i] = -gall[(size_t)p * n + i];
}

// The end of an iteration (:728-743) and the head of the next one (:650, :659-669).
static __global__ void __launch_bounds__(256)
k_bfl_dir_done(int n, BfOpts o, const double *__restrict__ xall, const double *__restrict__ gall, double *__restrict__ dxall,
               const double *__restrict__ wall, double *__restrict__ xnewall, const int32_t *__restrict__ info,
               LmState *__restrict__ st, BfState *__restrict__ bs)
{
    __shared__ double scratch[3 * NLH_NCH + 8];
    __shared__ __attribute__((aligned(16))) double buf[NT_CHUNK];
    __shared__ double xch[2], red[16];
    const int p = blockIdx.x, tid = threadIdx.x, BS = blockDim.x;
    if (st[p].stage != BF_DIR) return;
    BfState *s = bs + p;
    const double *x = xall + (size_t)p * n, *g = gall + (size_t)p * n, *w = wall + (size_t)p * n;
    double *dx = dxall + (size_t)p * n, *xnew = xnewall + (size_t)p * n;
    for (int i = tid; i < n; i += BS) dx[i] = w[i];
    __syncthreads();
    if (info[p]) {                                                      // linalg: the matrix is not positive definite
        if (tid == 0) { s->rc = o.rc_invalid_op; st[p].stage = BF_DONE; }
        return;
    }
    const int neval = s->neval;
    if (tid == 0) {
        s->print_due = 1; s->pr_iter = s->iter; s->pr_neval = neval; s->pr_fp = s->fp; s->pr_xtest = s->xtest; s->pr_gtest = s->gtest;
    }
    if (neval >= o.max_evals) {                                         // :740-743
        if (tid == 0) { s->flag = 1; st[p].stage = BF_DONE; }
        return;
    }
    if (tid == 0) s->iter += 1;                                         // :650
    __syncthreads();
    bfl_search_begin(n, o, x, g, dx, xnew, st + p, s, buf, xch, red, scratch);
}

// After F(xnew): one turn of ls_search_miso's loop (:402-468), or the plain step; an accepted point becomes x (:672-678).
static __global__ void __launch_bounds__(256)
k_bfl_trial(int m, int n, BfOpts o, double *__restrict__ xall, double *__restrict__ xnewall, double *__restrict__ dxall,
            const double *__restrict__ gall, double *__restrict__ goldall, const double *__restrict__ fall,
            LmState *__restrict__ st, BfState *__restrict__ bs, int scalar)
{
    __shared__ __attribute__((aligned(16))) double buf[NT_CHUNK];
    __shared__ double xch[2];
    const int p = blockIdx.x, tid = threadIdx.x, BS = blockDim.x;
    if (st[p].stage != BF_TRIAL) return;
    BfState *s = bs + p;
    double *x = xall + (size_t)p * n, *xnew = xnewall + (size_t)p * n, *dx = dxall + (size_t)p * n, *gold = goldall + (size_t)p * n;
    const double *g = gall + (size_t)p * n, *fv = fall + (size_t)p * m;
    // (every thread takes its copy of the search state before the first barrier: thread 0 rewrites it at the end)
    const BfState q = *s;
    const double f = scalar ? fv[0] : 0.5 * nt_ordered_sum(m, [&](int i) { return fv[i] * fv[i]; }, buf, xch);
    int add = 1;
    if (o.use_line_search) {
        const int lsn = q.ls_neval + 1, lsi = q.ls_iter + 1;
        const double alam = q.alam, fold = q.fp, slope = q.slope;
        bool accept = false;
        if (alam < q.alamin) {
            const double sq = nt_ordered_sum(n, [&](int i) { const double d = xnew[i] - x[i]; return d * d; }, buf, xch);
            if (sqrt(sq) == 0.0) {
                if (tid == 0) { s->neval += lsn; s->ls_neval = lsn; s->ls_iter = lsi; s->rc = o.rc_convergence; st[p].stage = BF_DONE; }
                return;
            }
            __syncthreads();
            for (int i = tid; i < n; i += BS) xnew[i] = x[i];
            __syncthreads();
            accept = true;
        } else if (f <= fold + o.ls_alpha * alam * slope) {
            accept = true;
        }
        if (!accept) {
            const double tmplam = nlh_min_backtrack_search(lsi, fold, f, q.f1, alam, q.alam1, slope);
            const double nalam = fmax(tmplam, o.ls_factor * alam);
            if (lsn >= o.ls_max_evals) {                                // the search gives up: reported as a convergence error
                if (tid == 0) {
                    s->neval += lsn; s->ls_neval = lsn; s->ls_iter = lsi; s->alam1 = alam; s->f1 = f; s->alam = nalam;
                    s->rc = o.rc_convergence; st[p].stage = BF_DONE;
                }
                return;
            }
            for (int i = tid; i < n; i += BS) xnew[i] = x[i] + nalam * dx[i];
            if (tid == 0) { s->alam1 = alam; s->f1 = f; s->alam = nalam; s->ls_neval = lsn; s->ls_iter = lsi; }
            return;                                                     // stays BF_TRIAL
        }
        add = lsn;
        if (tid == 0) { s->ls_neval = lsn; s->ls_iter = lsi; }
    }
    __syncthreads();
    for (int i = tid; i < n; i += BS) {                                 // :672-678
        const double xn = xnew[i];
        dx[i] = xn - x[i];
        x[i] = xn;
        gold[i] = g[i];
    }
    if (tid == 0) { s->fp = f; s->neval += add; st[p].stage = BF_GRAD; }
}

static __global__ void __launch_bounds__(256)
k_bfl_count(int nprob, const LmState *__restrict__ st, int32_t *__restrict__ counts)
{
    __shared__ int c[2];
    if (threadIdx.x < 2) c[threadIdx.x] = 0;
    __syncthreads();
    int a = 0, b = 0;
    for (int p = threadIdx.x; p < nprob; p += blockDim.x) {
        const int sg = st[p].stage;
        a += (sg == BF_GRAD);
        b += (sg == BF_TRIAL);
    }
    if (a) atomicAdd(&c[0], a);
    if (b) atomicAdd(&c[1], b);
    __syncthreads();
    if (threadIdx.x < 2) counts[threadIdx.x] = c[threadIdx.x];
}
