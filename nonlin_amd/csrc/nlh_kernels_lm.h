// nlh_kernels_lm.h -- lmpar / lmsolve and the trust-region update of lss_solve,
// one workgroup per problem (the n-by-n problem is a dependency chain; throughput comes
// from running one problem per workgroup across the whole chip).
//
//   lmsolve_dev : MINPACK qrsolv, src/nonlin_least_squares.f90:670-791
//   lmpar_dev   : MINPACK lmpar with the reference's two deviations, :394-566
//   k_lmpar     : lmpar + :286-294 (step, trial point, pnorm) + :307-313 (||R P^T p||)
//   k_lm_update : :299-365 (ratio test, trust-region update, acceptance, convergence)
//
// EXACT = true: every reduction (norms, dot products) runs in the reference's operation
// order (nlh_common.h), which makes the whole step bit-identical to the CPU path.
#pragma once
#include "nlh_common.h"
#include "nlh_kernels_factor.h"

// Faithful lmsolve on the n-by-n R (global, ld = ldr; strict lower triangle is scratch).
// diagv[l] is the diagonal of sqrt(par) D; x (indexed by original column), sdiag, wa in LDS.
template <bool EXACT>
__device__ void lmsolve_dev(int n, double *r, int ldr, const int32_t *ipvt, const double *diagv,
                            const double *qtb, double *x, double *sdiag, double *wa, double *red)
{
    const int tid = threadIdx.x, BS = blockDim.x;
    for (int j = 0; j < n; ++j) {                              // :710-714
        for (int i = j + tid; i < n; i += BS) r[(size_t)j * ldr + i] = r[(size_t)i * ldr + j];
    }
    for (int j = tid; j < n; j += BS) { x[j] = r[(size_t)j * ldr + j]; wa[j] = qtb[j]; }
    __syncthreads();

    for (int j = 0; j < n; ++j) {                              // :717-765
        const int l = ipvt[j];
        const double dl = diagv[l];
        if (dl != 0.0) {
            for (int k = j + tid; k < n; k += BS) sdiag[k] = (k == j) ? dl : 0.0;
            __syncthreads();
            double qtbpj = 0.0;
            for (int k = j; k < n; ++k) {
                const double sk = sdiag[k];
                if (sk == 0.0) continue;                       // uniform
                const double rkk = r[(size_t)k * ldr + k];
                double cs, sn;
                if (fabs(rkk) < fabs(sk)) {                    // :733-741
                    const double ctan = rkk / sk;
                    sn = 0.5 / sqrt(0.25 + 0.25 * (ctan * ctan));
                    cs = sn * ctan;
                } else {
                    const double tn = sk / rkk;
                    cs = 0.5 / sqrt(0.25 + 0.25 * (tn * tn));
                    sn = cs * tn;
                }
                const double wk = wa[k];
                const double temp = cs * wk + sn * qtbpj;
                qtbpj = -sn * wk + cs * qtbpj;
                __syncthreads();                               // all have read sdiag[k], wa[k], r(k,k)
                if (tid == 0) { r[(size_t)k * ldr + k] = cs * rkk + sn * sk; wa[k] = temp; }
                double *colk = r + (size_t)k * ldr;
                for (int i = k + 1 + tid; i < n; i += BS) {    // :753-757
                    const double rik = colk[i], si = sdiag[i];
                    colk[i] = cs * rik + sn * si;
                    sdiag[i] = -sn * rik + cs * si;
                }
                __syncthreads();
            }
        }
        __syncthreads();
        if (tid == 0) {                                        // :763-764
            sdiag[j] = r[(size_t)j * ldr + j];
            r[(size_t)j * ldr + j] = x[j];
        }
        __syncthreads();
    }

    // singular tail (:769-773) and back-substitution on S^T stored in the lower triangle (:774-784)
    int ns = n;
    for (int j = tid; j < n; j += BS)
        if (sdiag[j] == 0.0) ns = min(ns, j);
    ns = -(int)block_reduce_max((double)(-ns), red);
    __syncthreads();
    for (int j = ns + tid; j < n; j += BS) wa[j] = 0.0;
    __syncthreads();
    for (int k = 1; k <= ns; ++k) {
        const int j = ns - k;
        const double *colj = r + (size_t)j * ldr;
        const double sm = sum_block<EXACT>([&](int i) { return colj[j + 1 + i] * wa[j + 1 + i]; }, ns - j - 1, red);
        __syncthreads();
        if (tid == 0) wa[j] = (wa[j] - sm) / sdiag[j];
        __syncthreads();
    }
    for (int j = tid; j < n; j += BS) x[ipvt[j]] = wa[j];       // :787-790
    __syncthreads();
}

// lmpar.  Vectors x, sdiag, wa1, wa2n (the first n entries of the caller's wa4), z are LDS.
// Deviation A (:531) takes the norm over all m entries of the caller's wa4: EXACT reads the
// tail wa4[n..m) itself, otherwise tailsq = its sum of squares.  ne_mode: the factors come
// from the Gram matrix (row signs unknown, no Q^T f tail), so only the Gauss-Newton
// acceptance test is run; returns 1 when the iteration would be needed (=> QR fallback).
template <bool EXACT>
__device__ int lmpar_dev(int m, int n, double *r, int ldr, const int32_t *ipvt, const double *diag,
                         const double *qtb, double delta, double *par_io, double tailsq,
                         const double *wa4, double *x, double *sdiag, double *wa1, double *wa2n,
                         double *z, double *red, double *scratch, int ne_mode)
{
    const int tid = threadIdx.x, BS = blockDim.x;
    const double p1 = 0.1, p001 = 1.0e-3;
    double par = *par_io;

    // Gauss-Newton direction (:447-469)
    int nsing = n;
    for (int j = tid; j < n; j += BS)
        if (r[(size_t)j * ldr + j] == 0.0) nsing = min(nsing, j);
    nsing = -(int)block_reduce_max((double)(-nsing), red);
    for (int j = tid; j < n; j += BS) wa1[j] = (j < nsing) ? qtb[j] : 0.0;
    __syncthreads();
    for (int k = 1; k <= nsing; ++k) {
        const int j = nsing - k;
        const double temp = wa1[j] / r[(size_t)j * ldr + j];
        const double *colj = r + (size_t)j * ldr;
        for (int i = tid; i < j; i += BS) wa1[i] = wa1[i] - colj[i] * temp;
        if (tid == 0) z[j] = temp;
        __syncthreads();
    }
    for (int j = tid; j < n; j += BS) {
        const double v = (j < nsing) ? z[j] : 0.0;
        wa1[j] = v;
        x[ipvt[j]] = v;
    }
    __syncthreads();

    // :473-481
    for (int i = tid; i < n; i += BS) wa2n[i] = diag[i] * x[i];
    __syncthreads();
    double dxnorm = nrm2_block<EXACT>([&](int i) { return wa2n[i]; }, n, red, scratch);
    double fp = dxnorm - delta;
    if (fp <= p1 * delta) { *par_io = 0.0; return 0; }
    if (ne_mode) return 1;

    // lower bound parl (:486-503)
    double parl = 0.0, temp;
    if (nsing == n) {
        __syncthreads();
        for (int j = tid; j < n; j += BS) { const int l = ipvt[j]; wa1[j] = diag[l] * (wa2n[l] / dxnorm); }
        __syncthreads();
        for (int j = 0; j < n; ++j) {
            const double *colj = r + (size_t)j * ldr;
            const double sm = sum_block<EXACT>([&](int i) { return colj[i] * wa1[i]; }, j, red);
            __syncthreads();
            if (tid == 0) wa1[j] = (wa1[j] - sm) / colj[j];
            __syncthreads();
        }
        temp = nrm2_block<EXACT>([&](int j) { return wa1[j]; }, n, red, scratch);
        parl = ((fp / delta) / temp) / temp;
    }
    __syncthreads();
    // upper bound paru (:506-513): thread per column, rows ascending
    for (int j = tid; j < n; j += BS) {
        double sm = 0.0;
        const double *colj = r + (size_t)j * ldr;
        for (int i = 0; i <= j; ++i) sm = sm + colj[i] * qtb[i];
        wa1[j] = sm / diag[ipvt[j]];
    }
    __syncthreads();
    const double gnorm = nrm2_block<EXACT>([&](int j) { return wa1[j]; }, n, red, scratch);
    double paru = gnorm / delta;
    if (paru == 0.0) paru = NLH_DWARF / fmin(delta, p1);

    par = fmax(par, parl);                                     // :517-519
    par = fmin(par, paru);
    if (par == 0.0) par = gnorm / dxnorm;

    for (int iter = 1;; ++iter) {                              // :522-563
        if (par == 0.0) par = fmax(NLH_DWARF, p001 * paru);
        temp = sqrt(par);
        __syncthreads();
        for (int i = tid; i < n; i += BS) wa1[i] = temp * diag[i];
        __syncthreads();
        lmsolve_dev<EXACT>(n, r, ldr, ipvt, wa1, qtb, x, sdiag, wa2n, red);
        for (int i = tid; i < n; i += BS) wa2n[i] = diag[i] * x[i];
        __syncthreads();
        if (EXACT) {                                           // :531 deviation A: norm over all m entries
            dxnorm = norm2_flang_block([&](int i) { return i < n ? wa2n[i] : wa4[i]; }, m, scratch);
        } else {
            double sq = 0.0;
            for (int i = tid; i < n; i += BS) sq = sq + wa2n[i] * wa2n[i];
            dxnorm = sqrt(block_reduce_sum(sq, red) + tailsq);
        }
        temp = fp;
        fp = dxnorm - delta;
#ifdef NLH_DEBUG_LMPAR
        if (tid == 0 && blockIdx.x == 0)
            printf("[gpu lmpar] iter=%d par=%.17g parl=%.6g paru=%.6g dxnorm=%.17g fp=%.6g delta=%.6g\n",
                   iter, par, parl, paru, dxnorm, fp, delta);
#endif
        if (fabs(fp) <= p1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10) break;

        __syncthreads();
        for (int j = tid; j < n; j += BS) { const int l = ipvt[j]; wa1[j] = diag[l] * (wa2n[l] / dxnorm); }
        __syncthreads();
        for (int j = 0; j < n; ++j) {                          // :547-553
            const double t = wa1[j] / sdiag[j];
            __syncthreads();                                   // all have read wa1[j]
            if (j + 1 < n) {
                const double *colj = r + (size_t)j * ldr;
                for (int i = tid; i < n; i += BS) {            // :552 deviation B: every row
                    const double base = (i == j) ? t : wa1[i];
                    wa1[i] = base - colj[i] * t;
                }
            } else if (tid == 0) {
                wa1[j] = t;
            }
            __syncthreads();
        }
        temp = nrm2_block<EXACT>([&](int j) { return wa1[j]; }, n, red, scratch);
        const double parc = ((fp / delta) / temp) / temp;
        if (fp > 0.0) parl = fmax(parl, par);                  // :558-559
        if (fp < 0.0) paru = fmin(paru, par);
        par = fmax(parl, par + parc);                          // :562
    }
    *par_io = par;
    return 0;
}

// lmpar for every problem whose factors are ready, then the step and trial point.
// Dynamic LDS: (5n + 64) doubles, plus 3*NLH_NCH + 8 when EXACT.
template <bool EXACT>
__global__ void __launch_bounds__(1024)
k_lmpar(int m, int n, double *__restrict__ Rall, LmVecs v, double *__restrict__ xall,
        const double *__restrict__ wa4all, LmState *__restrict__ st, int want_stage)
{
    extern __shared__ double smem[];
    const int p = blockIdx.x;
    LmState *s = st + p;
    if (s->stage != want_stage) return;
    const int tid = threadIdx.x, BS = blockDim.x;
    double *xs = smem, *sdiag = smem + n, *wa1 = smem + 2 * n, *wa2n = smem + 3 * n, *z = smem + 4 * n;
    double *red = smem + 5 * n;
    double *scratch = red + 64;
    double *R = Rall + (size_t)p * n * n;
    const int32_t *ipvt = v.ipvt + (size_t)p * n;
    const double *diag = v.diag + (size_t)p * n;
    const double *qtf = v.qtf + (size_t)p * n;
    const double *xc = xall + (size_t)p * n;
    const int ne_mode = (want_stage == ST_NE_READY);

    double par = s->par;
    const double delta = s->delta;
    const int rc = lmpar_dev<EXACT>(m, n, R, n, ipvt, diag, qtf, delta, &par, s->tailsq,
                                    wa4all + (size_t)p * m, xs, sdiag, wa1, wa2n, z, red, scratch, ne_mode);
    __syncthreads();
    if (rc) {                       // Gauss-Newton step rejected on the normal-equations path
        if (tid == 0) s->stage = ST_NEED_QR;
        return;
    }
    // :286-291  p = -x_lmpar ; trial = x + p ; pnorm = ||D p||
    double *pw = v.wa1 + (size_t)p * n, *tw = v.wa2 + (size_t)p * n;
    for (int j = tid; j < n; j += BS) {
        const double pj = -xs[j];
        xs[j] = pj;
        pw[j] = pj;
        tw[j] = xc[j] + pj;
        wa1[j] = diag[j] * pj;
    }
    __syncthreads();
    const double pnorm = nrm2_block<EXACT>([&](int j) { return wa1[j]; }, n, red, scratch);
    __syncthreads();
    // :307-312  wa3 = R (P^T p), row i accumulates columns j ascending
    for (int i = tid; i < n; i += BS) {
        double acc = 0.0;
        for (int j = i; j < n; ++j) acc = acc + R[(size_t)j * n + i] * xs[ipvt[j]];
        z[i] = acc;
    }
    __syncthreads();
    const double t1 = nrm2_block<EXACT>([&](int i) { return z[i]; }, n, red, scratch);
    if (tid == 0) {
        s->par = par;
        s->pnorm = pnorm;
        s->temp1n = t1;
        if (s->iter == 1) s->delta = fmin(delta, pnorm);       // :294
        s->inner_pass += 1;
        s->stage = ST_TRIAL_READY;
    }
}

// :299-365 for every problem with a fresh trial residual.  One workgroup per problem
// (the acceptance copies fvec <- wa4, m entries).  EXACT: fnorm1 = NORM2(wa4) in reference
// order; otherwise from the per-block partial sums written by the residual kernel.
template <bool EXACT>
__global__ void __launch_bounds__(256)
k_lm_update(int m, int n, int nblk, const double *__restrict__ part, LmVecs v,
            double *__restrict__ xall, double *__restrict__ fvec, const double *__restrict__ wa4,
            LmState *__restrict__ st, double ftol, double xtol, int maxeval)
{
    __shared__ double red[16];
    __shared__ double scratch[EXACT ? (3 * NLH_NCH + 8) : 1];
    const int p = blockIdx.x;
    LmState *s = st + p;
    if (s->stage != ST_TRIAL_DONE) return;
    const int tid = threadIdx.x, BS = blockDim.x;
    const double p1 = 0.1, half = 0.5, one = 1.0;

    double fnorm1, tq = 0.0;                                    // fnorm1 = ||wa4|| (:299)
    if (EXACT) {
        const double *w = wa4 + (size_t)p * m;
        fnorm1 = norm2_flang_block([&](int i) { return w[i]; }, m, scratch);
    } else {
        double sq = 0.0;
        for (int k = 0; k < nblk; ++k) {                        // fixed order
            sq = sq + part[((size_t)p * nblk + k) * 2 + 0];
            tq = tq + part[((size_t)p * nblk + k) * 2 + 1];
        }
        fnorm1 = sqrt(sq);
    }
    const double fnorm = s->fnorm, pnorm = s->pnorm, temp1n = s->temp1n, gnorm0 = s->gnorm;
    double par = s->par, delta = s->delta, xnorm = s->xnorm;
    const int iter = s->iter, neval0 = s->neval, fkind = s->factor_kind;
    __syncthreads();            // every thread holds the state before thread 0 rewrites it

    double actred = -one;                                       // :302-303
    if (p1 * fnorm1 < fnorm) { const double q = fnorm1 / fnorm; actred = one - q * q; }
    const double temp1 = temp1n / fnorm;                        // :313-316
    const double temp2 = (sqrt(par) * pnorm) / fnorm;
    const double prered = temp1 * temp1 + (temp2 * temp2) / half;
    const double dirder = -(temp1 * temp1 + temp2 * temp2);
    double ratio = 0.0;                                         // :319-320
    if (prered != 0.0) ratio = actred / prered;

    if (ratio <= 0.25) {                                        // :323-337
        double temp = 0.0;
        if (actred >= 0.0) temp = half;
        if (actred < 0.0) temp = half * dirder / (dirder + half * actred);
        if (p1 * fnorm1 >= fnorm || temp < p1) temp = p1;
        delta = temp * fmin(delta, pnorm / p1);
        par = par / temp;
    } else if (!(par != 0.0 && ratio < 0.75)) {
        delta = pnorm / half;
        par = half * par;
    }

    const int accept = (ratio >= 1.0e-4);                       // :340-349
    if (accept) {
        const double *tw = v.wa2 + (size_t)p * n;
        const double *diag = v.diag + (size_t)p * n;
        for (int j = tid; j < n; j += BS) xall[(size_t)p * n + j] = tw[j];
        xnorm = nrm2_block<EXACT>([&](int j) { return diag[j] * tw[j]; }, n, red, scratch);
        for (int i = tid; i < m; i += BS) fvec[(size_t)p * m + i] = wa4[(size_t)p * m + i];
    }
    if (tid != 0) return;

    int fcnvrg = 0, xcnvrg = 0, flag = 0;
    int niter = iter, neval = neval0 + 1;
    double fn = fnorm;
    if (accept) { fn = fnorm1; niter = iter + 1; }
    if (fabs(actred) <= ftol && prered <= ftol && half * ratio <= one) fcnvrg = 1;   // :352-355
    if (delta <= xtol * xnorm) xcnvrg = 1;
    if (!(fcnvrg || xcnvrg)) {                                  // :358-363
        if (neval >= maxeval) flag = 106;                      // NL_CONVERGENCE_ERROR
        if (fabs(actred) <= NLH_EPS && prered <= NLH_EPS && half * ratio <= one) flag = 208;
        if (delta <= NLH_EPS * xnorm) flag = 208;
        if (gnorm0 <= NLH_EPS) flag = 208;
    }
    s->fnorm1 = fnorm1;
    s->fnorm = fn;
    s->xnorm = xnorm;
    s->par = par;
    s->delta = delta;
    s->iter = niter;
    s->neval = neval;
    s->tailsq = tq;            // wa4 now holds the trial residual (deviation A on the next lmpar)
    s->fcnvrg = fcnvrg;
    s->xcnvrg = xcnvrg;
    s->flag = flag;
    if (fcnvrg || xcnvrg || flag) s->stage = ST_DONE;
    else if (accept) { s->stage = ST_NEED_JAC; s->inner_pass = 0; s->head_done = 0; }
    else s->stage = (fkind == 1) ? ST_QR_READY : ST_NE_READY;   // inner loop again
}
