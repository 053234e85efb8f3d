// nlh_kernels_cls.h -- cls_solve (src/nonlin_least_squares.f90:938-1176: bounded least squares by a Coleman-Li scaled
// dog-leg) as a LOCK-STEP BATCH, built like nlh_kernels_newton.h: every problem carries a stage, every kernel of a round
// is launched over all problems and returns at once for problems in another stage, one small read-back per round.  The
// O(m + n) logic of the reference -- coleman_li_scaling (:1222-1260), the dog-leg (:1301-1403), alpha_box (:1181-1219),
// the trust-region update and the projected backtracking (:1055-1123), the convergence tests (:1125-1149) -- runs here,
// one workgroup per problem, in the reference's operation order: every dot product is ONE ordered chain of adds
// (nt_ordered_sum), NORM2 is the flang runtime's algorithm, minima / maxima are exact in any order.  Given bit-identical
// Jacobian, QR and residual kernels, every decision and every iterate is bit-identical to the host loop (cls_core) and to
// the CPU path.
#pragma once
#include <cfloat>
#include "nlh_kernels_newton.h"

enum ClStage : int32_t {
    CL_START = 30,      // x projected, F(x0) evaluated: norms and the finiteness test due (:1023-1031)
    CL_NEED_JAC = 31,   // iteration head: Jacobian, QR, Gauss-Newton step, gradient, then the first half of the dog-leg
    CL_DOG_SD = 32,     // the Gauss-Newton step leaves the region: J g due, then the steepest-descent leg (:1340-1390)
    CL_PRED = 33,       // p chosen: J p due, then the predicted reduction and the trial point (:1398-1403, :1055-1058)
    CL_TRIAL = 34,      // xnew holds x + p: F(xnew) due, then the ratio test (:1060-1090)
    CL_BT = 35,         // xnew holds a projected backtracking point: F(xnew) due (:1096-1118)
    CL_DONE = ST_DONE
};

struct ClState {
    double fnorm, xnorm, gnorm, fnewnorm, actred, prered, rho, delta, stepscale, dderiv;
    double pr_xnorm, pr_fnorm;        // the status block of this iteration (:1040-1044), printed by the host for a lone solve
    int32_t iter, neval, njac;
    int32_t xcnvrg, fcnvrg, gcnvrg, converged;
    int32_t silent;                   // non-finite start: the reference returns without a word (:1028-1031)
    int32_t bt_k;                     // backtracking trial number, 1 .. 10
    int32_t print_due, pr_iter, pr_neval, pr_njac, pad;
};

struct ClOpts {
    double ftol, xtol, gtol, delta0, stepscale0;
    int32_t max_evals, pad;
};

// min_i term(i) over the terms that are not NaN (DBL_MAX when there is none), broadcast
template <typename Term>
__device__ double cl_block_min(int n, Term term, double *red)
{
    const int tid = threadIdx.x, BS = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = (BS + 63) >> 6;
    double v = DBL_MAX;
    for (int i = tid; i < n; i += BS) v = fmin(term(i), v);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    double r = red[0];
    for (int w = 1; w < nw; ++w) r = fmin(r, red[w]);
    __syncthreads();
    return r;
}

// any_i pred(i), broadcast
template <typename Pred>
__device__ bool cl_block_any(int n, Pred pred, double *red)
{
    return nt_block_max(n, [&](int i) { return pred(i) ? 1.0 : 0.0; }, red) != 0.0;
}

__device__ __forceinline__ bool cl_bad(double v) { return !(v == v) || fabs(v) == DBL_MAX; }       // :1276-1298

// alpha_box (:1181-1219) and the shortening of p (:1392-1395).  The reference walks i upwards, returns 0 at the first
// variable that is already outside its bound in the direction of travel and keeps the smallest ratio otherwise: any such
// variable gives 0, and a minimum is exact in any order (a NaN ratio loses every `a < rst`, as it loses fmin).
__device__ void cl_box_scale(int n, const double *x, double *p, const double *xl, const double *xu, double *red)
{
    const int tid = threadIdx.x, BS = blockDim.x;
    const bool outside = cl_block_any(n, [&](int i) { return (p[i] > 0.0 && xu[i] < x[i]) || (p[i] < 0.0 && xl[i] > x[i]); }, red);
    double alpha = 0.0;
    if (!outside) {
        alpha = cl_block_min(n, [&](int i) {
            if (p[i] > 0.0) return (xu[i] - x[i]) / p[i];
            if (p[i] < 0.0) return (xl[i] - x[i]) / p[i];
            return DBL_MAX;
        }, red);
        if (alpha < 0.0) alpha = 0.0;
    }
    if (alpha < 1.0) {
        __syncthreads();
        for (int i = tid; i < n; i += BS) p[i] = alpha * p[i];
        __syncthreads();
    }
}

// apply_limits (:858-883)
__device__ __forceinline__ double cl_clamp(double v, double lo, double hi)
{
    if (v < lo) v = lo;
    if (v > hi) v = hi;
    return v;
}

// The end of an iteration (:1125-1149): finiteness, the three convergence tests, the evaluation budget.
__device__ void cl_finish_iter(int m, int n, const ClOpts &o, const double *x, const double *fv, LmState *sg, ClState *s, double *red)
{
    const bool bad = cl_block_any(n, [&](int i) { return cl_bad(x[i]); }, red) || cl_block_any(m, [&](int i) { return cl_bad(fv[i]); }, red);
    if (threadIdx.x != 0) return;
    int next = CL_NEED_JAC;
    if (bad) next = CL_DONE;
    else if (s->xnorm <= o.xtol) { s->converged = 1; s->xcnvrg = 1; next = CL_DONE; }
    else if (fabs(s->actred) <= o.ftol && fabs(s->prered) <= o.ftol && 0.5 * s->rho <= 1.0) { s->converged = 1; s->fcnvrg = 1; next = CL_DONE; }
    else if (s->gnorm <= o.gtol) { s->converged = 1; s->gcnvrg = 1; next = CL_DONE; }
    else if (s->neval >= o.max_evals) next = CL_DONE;
    sg->stage = next;
}

static __global__ void __launch_bounds__(256)
k_cls_reset(int nprob, double delta0, LmState *__restrict__ st, ClState *__restrict__ cs)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nprob) return;
    ClState z;
    z.fnorm = z.xnorm = z.gnorm = z.fnewnorm = z.actred = z.prered = z.rho = z.stepscale = z.dderiv = 0.0;
    z.pr_xnorm = z.pr_fnorm = 0.0;
    z.delta = delta0;                                                   // :1034
    z.iter = 1; z.neval = z.njac = 0;
    z.xcnvrg = z.fcnvrg = z.gcnvrg = z.converged = z.silent = z.bt_k = 0;
    z.print_due = z.pr_iter = z.pr_neval = z.pr_njac = z.pad = 0;
    cs[p] = z;
    st[p].stage = CL_START;
}

// x <- the box (:1023): every problem
static __global__ void __launch_bounds__(256)
k_cls_limits(int n, const double *__restrict__ xl, const double *__restrict__ xu, double *__restrict__ xall)
{
    const int p = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) xall[(size_t)p * n + i] = cl_clamp(xall[(size_t)p * n + i], xl[i], xu[i]);
}

// :1024-1031 after F(x0)
static __global__ void __launch_bounds__(256)
k_cls_start(int m, int n, const double *__restrict__ xall, const double *__restrict__ fall, LmState *__restrict__ st,
            ClState *__restrict__ cs)
{
    __shared__ double scratch[3 * NLH_NCH + 8];
    __shared__ double red[16];
    const int p = blockIdx.x;
    if (st[p].stage != CL_START) return;
    const double *x = xall + (size_t)p * n, *fv = fall + (size_t)p * m;
    const double fnorm = norm2_flang_block([&](int i) { return fv[i]; }, m, scratch);
    const double xnorm = norm2_flang_block([&](int i) { return x[i]; }, n, scratch);
    const bool bad = cl_block_any(n, [&](int i) { return cl_bad(x[i]); }, red) || cl_block_any(m, [&](int i) { return cl_bad(fv[i]); }, red);
    if (threadIdx.x == 0) {
        ClState *s = cs + p;
        s->neval = 1; s->fnorm = fnorm; s->xnorm = xnorm;
        if (bad) { s->silent = 1; st[p].stage = CL_DONE; }
        else st[p].stage = CL_NEED_JAC;
    }
}

// the residual joins the working array of the QR as its extra column (:1047, :1334)
static __global__ void __launch_bounds__(256)
k_cls_qr_prep(int m, const double *__restrict__ fall, double *__restrict__ Eall, const LmState *__restrict__ st)
{
    const int p = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (st[p].stage != CL_NEED_JAC) return;
    if (i < m) Eall[(size_t)p * m + i] = fall[(size_t)p * m + i];
}

// After the QR (u = R^-1 (Q^T f)(1:n) in the head of E, g = J^T f): coleman_li_scaling (:1222-1260), the Gauss-Newton
// step and its scaled length (:1336-1338); inside the region it is the step.
static __global__ void __launch_bounds__(256)
k_cls_dog1(int m, int n, const double *__restrict__ xall, const double *__restrict__ xl, const double *__restrict__ xu,
           const double *__restrict__ Eall, double *__restrict__ scall, double *__restrict__ pgnall, double *__restrict__ pall,
           LmState *__restrict__ st, ClState *__restrict__ cs)
{
    __shared__ double scratch[3 * NLH_NCH + 8];
    __shared__ double red[16];
    const int p = blockIdx.x, tid = threadIdx.x, BS = blockDim.x;
    if (st[p].stage != CL_NEED_JAC) return;
    ClState *s = cs + p;
    const double *x = xall + (size_t)p * n, *u = Eall + (size_t)p * m;
    double *sc = scall + (size_t)p * n, *pgn = pgnall + (size_t)p * n, *pp = pall + (size_t)p * n;
    for (int i = tid; i < n; i += BS) {
        const double big = DBL_MAX;
        double di;
        if (xl[i] > -big && xu[i] < big) di = fmin(x[i] - xl[i], xu[i] - x[i]);
        else if (xl[i] > -big) di = x[i] - xl[i];
        else if (xu[i] < big) di = xu[i] - x[i];
        else di = 1.0;
        di = fmax(di, 1.0e-8);
        double sv = 1.0 / di;
        if (sv > 1.0e8) sv = 1.0e8;
        sc[i] = sv;
        pgn[i] = -u[i];
    }
    __syncthreads();
    const double pgnnorm = norm2_flang_block([&](int i) { return pgn[i] * sc[i]; }, n, scratch);
    const double delta = s->delta;
    if (tid == 0) {                                                     // the status block of this iteration (:1040-1044)
        s->njac += 1;
        s->print_due = 1; s->pr_iter = s->iter; s->pr_neval = s->neval; s->pr_njac = s->njac; s->pr_xnorm = s->xnorm; s->pr_fnorm = s->fnorm;
    }
    if (pgnnorm > delta) {
        if (tid == 0) st[p].stage = CL_DOG_SD;
        return;
    }
    for (int i = tid; i < n; i += BS) pp[i] = pgn[i];
    __syncthreads();
    cl_box_scale(n, x, pp, xl, xu, red);
    if (tid == 0) st[p].stage = CL_PRED;
}

// The steepest-descent leg (:1340-1390), given J g.
static __global__ void __launch_bounds__(256)
k_cls_dog2(int m, int n, const double *__restrict__ xall, const double *__restrict__ xl, const double *__restrict__ xu,
           const double *__restrict__ gall, const double *__restrict__ Jgall, const double *__restrict__ scall,
           const double *__restrict__ pgnall, double *__restrict__ psdall, double *__restrict__ uall, double *__restrict__ pall,
           LmState *__restrict__ st, ClState *__restrict__ cs)
{
    __shared__ double scratch[3 * NLH_NCH + 8];
    __shared__ __attribute__((aligned(16))) double buf[NT_CHUNK];
    __shared__ double xch[2];
    __shared__ double red[16];
    const int p = blockIdx.x, tid = threadIdx.x, BS = blockDim.x;
    if (st[p].stage != CL_DOG_SD) return;
    ClState *s = cs + p;
    const double *x = xall + (size_t)p * n, *g = gall + (size_t)p * n, *Jg = Jgall + (size_t)p * m;
    const double *sc = scall + (size_t)p * n, *pgn = pgnall + (size_t)p * n;
    double *psd = psdall + (size_t)p * n, *u = uall + (size_t)p * n, *pp = pall + (size_t)p * n;
    const double delta = s->delta;
    const double c1 = nt_ordered_sum(n, [&](int i) { return g[i] * g[i]; }, buf, xch);
    const double c2 = nt_ordered_sum(m, [&](int i) { return Jg[i] * Jg[i]; }, buf, xch);
    const double alpha = (c2 > 0.0 && c1 > 0.0) ? c1 / c2 : 0.0;
    for (int i = tid; i < n; i += BS) psd[i] = -alpha * g[i];
    __syncthreads();
    const double psdnorm = norm2_flang_block([&](int i) { return psd[i] * sc[i]; }, n, scratch);
    if (psdnorm >= delta && psdnorm > 0.0) {
        const double f1 = delta / psdnorm;
        for (int i = tid; i < n; i += BS) pp[i] = f1 * psd[i];
    } else {
        for (int i = tid; i < n; i += BS) {
            double t = pgn[i] - psd[i];
            u[i] = sc[i] * t;
        }
        __syncthreads();
        // (v = sc * psd is formed on the fly: the same products)
        const double a = nt_ordered_sum(n, [&](int i) { return u[i] * u[i]; }, buf, xch);
        const double b = 2.0 * nt_ordered_sum(n, [&](int i) { return u[i] * (sc[i] * psd[i]); }, buf, xch);
        const double c = nt_ordered_sum(n, [&](int i) { const double v = sc[i] * psd[i]; return v * v; }, buf, xch) - delta * delta;
        if (a <= 0.0) {
            for (int i = tid; i < n; i += BS) pp[i] = psd[i];
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
            for (int i = tid; i < n; i += BS) pp[i] = psd[i] + t * u[i];
        }
    }
    __syncthreads();
    cl_box_scale(n, x, pp, xl, xu, red);
    if (tid == 0) st[p].stage = CL_PRED;
}

// Given J p: predicted reduction (:1398-1403), scaled step length and gradient norm (:1055-1057), the trial point.
static __global__ void __launch_bounds__(256)
k_cls_pred(int m, int n, const double *__restrict__ xall, const double *__restrict__ gall, const double *__restrict__ pall,
           const double *__restrict__ Jpall, const double *__restrict__ scall, double *__restrict__ xnewall,
           LmState *__restrict__ st, ClState *__restrict__ cs)
{
    __shared__ double scratch[3 * NLH_NCH + 8];
    __shared__ __attribute__((aligned(16))) double buf[NT_CHUNK];
    __shared__ double xch[2];
    const int p = blockIdx.x, tid = threadIdx.x, BS = blockDim.x;
    if (st[p].stage != CL_PRED) return;
    ClState *s = cs + p;
    const double *x = xall + (size_t)p * n, *g = gall + (size_t)p * n, *pp = pall + (size_t)p * n, *Jp = Jpall + (size_t)p * m;
    const double *sc = scall + (size_t)p * n;
    double *xnew = xnewall + (size_t)p * n;
    const double gp = nt_ordered_sum(n, [&](int i) { return g[i] * pp[i]; }, buf, xch);
    const double jj = nt_ordered_sum(m, [&](int i) { return Jp[i] * Jp[i]; }, buf, xch);
    const double prered = -gp - 0.5 * jj;
    const double xnorm = norm2_flang_block([&](int i) { return pp[i] * sc[i]; }, n, scratch);
    const double gnorm = norm2_flang_block([&](int i) { return g[i]; }, n, scratch);
    for (int i = tid; i < n; i += BS) xnew[i] = x[i] + pp[i];
    if (tid == 0) { s->prered = prered; s->xnorm = xnorm; s->gnorm = gnorm; st[p].stage = CL_TRIAL; }
}

// After F(x + p): actual reduction, ratio, trust-region radius, acceptance or the start of the backtracking (:1060-1123).
static __global__ void __launch_bounds__(256)
k_cls_judge(int m, int n, ClOpts o, double *__restrict__ xall, double *__restrict__ xnewall, double *__restrict__ fall,
            const double *__restrict__ fnewall, const double *__restrict__ gall, const double *__restrict__ pall,
            const double *__restrict__ xl, const double *__restrict__ xu, LmState *__restrict__ st, ClState *__restrict__ cs)
{
    __shared__ double scratch[3 * NLH_NCH + 8];
    __shared__ __attribute__((aligned(16))) double buf[NT_CHUNK];
    __shared__ double xch[2];
    __shared__ double red[16];
    const int p = blockIdx.x, tid = threadIdx.x, BS = blockDim.x;
    if (st[p].stage != CL_TRIAL) return;
    ClState *s = cs + p;
    double *x = xall + (size_t)p * n, *xnew = xnewall + (size_t)p * n, *fv = fall + (size_t)p * m;
    const double *fnew = fnewall + (size_t)p * m, *g = gall + (size_t)p * n, *pp = pall + (size_t)p * n;
    const double fnewnorm = norm2_flang_block([&](int i) { return fnew[i]; }, m, scratch);
    const double fnorm = s->fnorm, prered = s->prered, xnorm = s->xnorm;
    double delta = s->delta;
    const double actred = 0.5 * (fnorm * fnorm - fnewnorm * fnewnorm);  // :1065-1070
    const double rho = (prered > 0.0 && actred >= 0.0) ? actred / prered : 0.0;
    if (rho < 0.25) delta = fmax(0.25, 1.0e-12);                        // :1073-1077 (constant 0.25: as in the reference)
    else if (rho > 0.75 && fabs(xnorm - delta) < 1.0e-12 * delta) delta = fmin(2.0 * delta, 1.0e3);
    const bool accept = (rho > 0.1 && fnewnorm <= fnorm);               // :1080
    double dderiv = 0.0;
    if (!accept) dderiv = nt_ordered_sum(n, [&](int i) { return g[i] * pp[i]; }, buf, xch);
    __syncthreads();
    if (tid == 0) {
        s->neval += 1; s->fnewnorm = fnewnorm; s->actred = actred; s->rho = rho;
        if (accept) { s->fnorm = fnewnorm; s->iter += 1; }
        else if (dderiv >= 0.0) delta = fmax(0.5 * delta, 1.0e-12);
        s->delta = delta; s->dderiv = dderiv;
    }
    if (accept) {                                                       // :1081-1086
        for (int i = tid; i < n; i += BS) x[i] = cl_clamp(xnew[i], xl[i], xu[i]);
        for (int i = tid; i < m; i += BS) fv[i] = fnew[i];
    } else if (!(dderiv >= 0.0)) {                                      // :1096-1100: the first backtracking point
        const double stepscale = o.stepscale0;
        for (int i = tid; i < n; i += BS) xnew[i] = cl_clamp(x[i] + stepscale * pp[i], xl[i], xu[i]);
        if (tid == 0) { s->stepscale = stepscale; s->bt_k = 1; st[p].stage = CL_BT; }
        return;
    }
    __syncthreads();
    cl_finish_iter(m, n, o, x, fv, st + p, s, red);
}

// After F at a backtracking point (:1101-1118).
static __global__ void __launch_bounds__(256)
k_cls_bt(int m, int n, ClOpts o, double *__restrict__ xall, double *__restrict__ xnewall, double *__restrict__ fall,
         const double *__restrict__ fnewall, const double *__restrict__ pall, const double *__restrict__ xl,
         const double *__restrict__ xu, LmState *__restrict__ st, ClState *__restrict__ cs)
{
    __shared__ double scratch[3 * NLH_NCH + 8];
    __shared__ double red[16];
    const int p = blockIdx.x, tid = threadIdx.x, BS = blockDim.x;
    if (st[p].stage != CL_BT) return;
    ClState *s = cs + p;
    double *x = xall + (size_t)p * n, *xnew = xnewall + (size_t)p * n, *fv = fall + (size_t)p * m;
    const double *fnew = fnewall + (size_t)p * m, *pp = pall + (size_t)p * n;
    const double fnewnorm = norm2_flang_block([&](int i) { return fnew[i]; }, m, scratch);
    const double fnorm = s->fnorm, dderiv = s->dderiv, xnorm = s->xnorm;
    double stepscale = s->stepscale;
    const int k = s->bt_k;
    __syncthreads();
    if (fnewnorm <= fnorm + 1.0e-4 * stepscale * dderiv) {              // :1105-1112
        for (int i = tid; i < n; i += BS) x[i] = xnew[i];
        for (int i = tid; i < m; i += BS) fv[i] = fnew[i];
        if (tid == 0) {
            s->neval += 1; s->fnewnorm = fnewnorm; s->fnorm = fnewnorm; s->iter += 1;
            s->delta = fmax(stepscale * xnorm, 1.0e-12);
        }
    } else {
        stepscale = stepscale * 0.5;
        if (k + 1 <= 10) {
            for (int i = tid; i < n; i += BS) xnew[i] = cl_clamp(x[i] + stepscale * pp[i], xl[i], xu[i]);
            if (tid == 0) { s->neval += 1; s->fnewnorm = fnewnorm; s->stepscale = stepscale; s->bt_k = k + 1; }
            return;                                                     // stays in CL_BT
        }
        if (tid == 0) { s->neval += 1; s->fnewnorm = fnewnorm; s->stepscale = stepscale; s->delta = fmax(0.5 * s->delta, 1.0e-12); }   // :1120
    }
    __syncthreads();
    cl_finish_iter(m, n, o, x, fv, st + p, s, red);
}

static __global__ void __launch_bounds__(256)
k_cls_count(int nprob, const LmState *__restrict__ st, int32_t *__restrict__ counts)
{
    __shared__ int c[2];
    if (threadIdx.x < 2) c[threadIdx.x] = 0;
    __syncthreads();
    int a = 0, b = 0;
    for (int p = threadIdx.x; p < nprob; p += blockDim.x) {
        const int sg = st[p].stage;
        a += (sg == CL_NEED_JAC);
        b += (sg == CL_BT);
    }
    if (a) atomicAdd(&c[0], a);
    if (b) atomicAdd(&c[1], b);
    __syncthreads();
    if (threadIdx.x < 2) counts[threadIdx.x] = c[threadIdx.x];
}
