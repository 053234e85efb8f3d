// nlh_kernels_factor.h -- the two factorisations behind lmfactor's interface and the
// outer-loop head of lss_solve that consumes them.
//
//   k_chol_factor : pivoted Cholesky of G = J^T J (normal-equations path).  With
//                   MINPACK's pivot rule ("largest remaining column norm" = largest
//                   remaining Schur diagonal) P^T G P = R^T R gives lmfactor's R up to
//                   row signs; qtf = R^-T P^T g is carried as an extra column.
//   k_qr_factor   : lmfactor itself (pivoted Householder QR, src/nonlin_least_squares.f90:
//                   569-667) plus the Q^T f sweep (:241-253), one workgroup per problem.
//   lm_head       : :229-238 (first-iteration scaling), :256-267 (scaled gradient norm),
//                   :270-278 (gradient convergence, rescale).
#pragma once
#include "nlh_common.h"

struct LmVecs {          // per-problem n-vectors of the LM driver (device, [nprob][n] each)
    double *diag, *diag_prev, *qtf, *acnorm, *rdiag, *g, *wa1, *wa2, *wa3, *sdiag;
    int32_t *ipvt;
};

// Outer-loop head shared by both factorisations.  R: n-by-n upper (ld = ldr) with the
// true diagonal; x: current iterate.  Whole workgroup; red = reduction scratch.
template <bool EXACT>
__device__ void lm_head(int n, const double *R, int ldr, const int32_t *ipvt,
                        const double *acnorm, const double *qtf, const double *x,
                        double *diag, double *diag_prev, LmState *s, double factor,
                        double gtol, int ready_stage, double *red, double *scratch)
{
    const int tid = threadIdx.x, BS = blockDim.x;
    const int iter = s->iter;
    if (iter == 1) {                                            // :229-238
        for (int j = tid; j < n; j += BS) {
            double d = acnorm[j];
            if (d == 0.0) d = 1.0;
            diag[j] = d;
        }
        __syncthreads();
        const double xnorm = nrm2_block<EXACT>([&](int j) { return diag[j] * x[j]; }, n, red, scratch);
        if (tid == 0) {
            s->xnorm = xnorm;
            double delta = factor * xnorm;
            if (delta == 0.0) delta = factor;
            s->delta = delta;
        }
    } else {
        for (int j = tid; j < n; j += BS) diag_prev[j] = diag[j];
    }
    const double fnorm = s->fnorm;
    double gn = 0.0;                                            // :256-267
    if (fnorm != 0.0) {
        for (int j = tid; j < n; j += BS) {
            const int l = ipvt[j];
            if (acnorm[l] == 0.0) continue;
            double sm = 0.0;
            for (int i = 0; i <= j; ++i) sm = sm + R[(size_t)j * ldr + i] * (qtf[i] / fnorm);
            gn = fmax(gn, fabs(sm / acnorm[l]));
        }
    }
    gn = block_reduce_max(gn, red);
    __syncthreads();
    if (gn <= gtol) {                                           // :270-273
        if (tid == 0) { s->gnorm = gn; s->gcnvrg = 1; s->stage = ST_DONE; }
        return;
    }
    for (int j = tid; j < n; j += BS) diag[j] = fmax(diag[j], acnorm[j]);   // :276-278
    if (tid == 0) { s->gnorm = gn; s->stage = ready_stage; s->inner_pass = 0; s->head_done = 1; }
}

// ---------------------------------------------------------------------------
// Pivoted Cholesky, one workgroup per problem, G (n-by-n, column-major, symmetric,
// both triangles valid on entry) in global memory/L2; the upper triangle is overwritten
// by R.  Row j of R is staged in LDS each step so the rank-1 update reads it as a
// broadcast; columns are updated wave-per-column (contiguous, coalesced).
// Dynamic LDS: (2n + 64) doubles + 64 ints.
// standalone != 0: only factor (stage-level entry point nlh_chol_factor).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_chol_factor(int n, double *__restrict__ Gall, const double *__restrict__ gall,
              LmVecs v, const double *__restrict__ xall, LmState *__restrict__ st,
              int32_t *__restrict__ info, double factor, double gtol, double pivot_tol,
              int standalone)
{
    extern __shared__ double smem[];
    const int p = blockIdx.x;
    LmState *s = st ? st + p : nullptr;
    if (s && s->stage != ST_HAVE_JAC) return;
    const int tid = threadIdx.x, BS = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = BS >> 6;
    double *rowj = smem;            // n   : row j of R
    double *gp = smem + n;          // n   : permuted gradient -> qtf
    double *red = smem + 2 * n;     // 64
    int *redi = reinterpret_cast<int *>(red + 32);
    double *G = Gall + (size_t)p * n * n;
    int32_t *ipvt = v.ipvt + (size_t)p * n;
    double *acnorm = v.acnorm + (size_t)p * n;
    double *qtf = v.qtf + (size_t)p * n;

    for (int k = tid; k < n; k += BS) {
        const double d = G[(size_t)k * n + k];
        acnorm[k] = sqrt(fmax(d, 0.0));        // ||J(:,k)||, lmfactor :611-616
        ipvt[k] = k;
        gp[k] = gall[(size_t)p * n + k];
    }
    __syncthreads();

    int bad = 0;
    for (int j = 0; j < n; ++j) {
        // pivot: first maximum of the remaining Schur diagonal (:622-625)
        double bv = 0.0;
        int bk = 0x7fffffff;
        for (int k = j + tid; k < n; k += BS) {
            const double d = G[(size_t)k * n + k];
            if (bk == 0x7fffffff || d > bv) { bv = d; bk = k; }
        }
        const int q = block_argmax_first(bv, bk, red, redi);
        if (q != j) {                           // symmetric interchange j <-> q (upper storage)
            for (int r = tid; r < j; r += BS) {
                double t = G[(size_t)j * n + r];
                G[(size_t)j * n + r] = G[(size_t)q * n + r];
                G[(size_t)q * n + r] = t;
            }
            for (int r = j + 1 + tid; r < q; r += BS) {
                double t = G[(size_t)r * n + j];
                G[(size_t)r * n + j] = G[(size_t)q * n + r];
                G[(size_t)q * n + r] = t;
            }
            for (int c = q + 1 + tid; c < n; c += BS) {
                double t = G[(size_t)c * n + j];
                G[(size_t)c * n + j] = G[(size_t)c * n + q];
                G[(size_t)c * n + q] = t;
            }
            if (tid == 0) {
                double t = G[(size_t)j * n + j];
                G[(size_t)j * n + j] = G[(size_t)q * n + q];
                G[(size_t)q * n + q] = t;
                int32_t ti = ipvt[j]; ipvt[j] = ipvt[q]; ipvt[q] = ti;
                double tg = gp[j]; gp[j] = gp[q]; gp[q] = tg;
            }
            __syncthreads();
        }
        const double dj = G[(size_t)j * n + j];
        const double an = acnorm[ipvt[j]];
        if (!(dj > pivot_tol * an * an) || !(dj > 0.0)) { bad = j + 1; break; }   // uniform
        const double rjj = sqrt(dj);
        const double yj = gp[j] / rjj;          // qtf(j)
        // row j of R
        for (int c = j + 1 + tid; c < n; c += BS) {
            const double r = G[(size_t)c * n + j] / rjj;
            G[(size_t)c * n + j] = r;
            rowj[c] = r;
            gp[c] = gp[c] - r * yj;
        }
        __syncthreads();
        if (tid == 0) { G[(size_t)j * n + j] = rjj; gp[j] = yj; }
        // trailing update of the upper triangle: G(r,c) -= R(j,r) R(j,c), j < r <= c
        for (int c = j + 1 + wid; c < n; c += nw) {
            const double rc = rowj[c];
            double *col = G + (size_t)c * n;
            for (int r = j + 1 + lane; r <= c; r += 64) col[r] = col[r] - rowj[r] * rc;
        }
        __syncthreads();
    }
    if (bad) {
        if (tid == 0) {
            if (info) info[p] = bad;
            if (s) { s->stage = ST_NEED_QR; }
        }
        return;
    }
    for (int k = tid; k < n; k += BS) qtf[k] = gp[k];
    if (tid == 0 && info) info[p] = 0;
    __syncthreads();
    if (standalone || !s) return;
    if (tid == 0) s->factor_kind = 0;
    lm_head<false>(n, G, n, ipvt, acnorm, qtf, xall + (size_t)p * n, v.diag + (size_t)p * n,
                   v.diag_prev + (size_t)p * n, s, factor, gtol, ST_NE_READY, red, nullptr);
}

// ---------------------------------------------------------------------------
// lmfactor (MINPACK qrfac) + Q^T f, faithful: same pivot rule, same norm down-date
// with the 0.05 (rdiag/wa)^2 <= eps recompute test, same reflector scaling.  One
// workgroup per problem, the m-by-n Jacobian in global memory; a wave owns a trailing
// column per step (dot, axpy, norm down-date), so a step costs a handful of barriers.
// Only reduction order differs from the CPU path.
// Dynamic LDS: (3n + 64) doubles + 64 ints.
// Rout: n-by-n (ld n) receives R (upper, diagonal = rdiag) for lmpar.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_qr_factor(int m, int n, double *__restrict__ Jall, const double *__restrict__ fall,
            double *__restrict__ Rall, LmVecs v, double *__restrict__ wa4all,
            double *__restrict__ scratch_all /* [nprob][m], used when inner_pass > 0 */,
            const double *__restrict__ xall, LmState *__restrict__ st, double factor,
            double gtol, int standalone)
{
    extern __shared__ double smem[];
    const int p = blockIdx.x;
    LmState *s = st ? st + p : nullptr;
    if (s && s->stage != ST_NEED_QR) return;
    const int tid = threadIdx.x, BS = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = BS >> 6;
    double *rdiag = smem;           // n
    double *wa = smem + n;          // n
    double *red = smem + 2 * n;     // 64
    int *redi = reinterpret_cast<int *>(red + 32);
    double *a = Jall + (size_t)p * m * n;
    int32_t *ipvt = v.ipvt + (size_t)p * n;
    double *acnorm = v.acnorm + (size_t)p * n;
    double *qtf = v.qtf + (size_t)p * n;
    const int minmn = m < n ? m : n;
    const double p05 = 5.0e-2;

    // initial column norms (:611-616), wave per column
    for (int j = wid; j < n; j += nw) {
        const double *col = a + (size_t)j * m;
        double sq = 0.0;
        for (int i = lane; i < m; i += 64) sq = sq + col[i] * col[i];
        sq = wave_reduce_sum(sq);
        if (lane == 0) {
            const double nr = sqrt(sq);
            acnorm[j] = nr; rdiag[j] = nr; wa[j] = nr; ipvt[j] = j;
        }
    }
    __syncthreads();

    for (int j = 0; j < minmn; ++j) {
        // pivot (:622-637)
        double bv = 0.0;
        int bk = 0x7fffffff;
        for (int k = j + tid; k < n; k += BS) {
            const double d = rdiag[k];
            if (bk == 0x7fffffff || d > bv) { bv = d; bk = k; }
        }
        const int kmax = block_argmax_first(bv, bk, red, redi);
        if (kmax != j) {
            double *cj = a + (size_t)j * m, *ck = a + (size_t)kmax * m;
            for (int i = tid; i < m; i += BS) { double t = cj[i]; cj[i] = ck[i]; ck[i] = t; }
            if (tid == 0) {
                rdiag[kmax] = rdiag[j];
                wa[kmax] = wa[j];
                int32_t t = ipvt[j]; ipvt[j] = ipvt[kmax]; ipvt[kmax] = t;
            }
            __syncthreads();
        }
        // reflector (:642-646)
        double *cj = a + (size_t)j * m;
        double sq = 0.0;
        for (int i = j + tid; i < m; i += BS) sq = sq + cj[i] * cj[i];
        double ajnorm = sqrt(block_reduce_sum(sq, red));
        if (ajnorm != 0.0) {
            if (cj[j] < 0.0) ajnorm = -ajnorm;
            __syncthreads();                       // everyone has read cj[j]
            for (int i = j + tid; i < m; i += BS) {
                double t = cj[i] / ajnorm;
                if (i == j) t = t + 1.0;
                cj[i] = t;
            }
            __syncthreads();
            const double ajj = cj[j];
            // trailing columns (:652-662), wave per column
            for (int k = j + 1 + wid; k < n; k += nw) {
                double *ck = a + (size_t)k * m;
                double sm = 0.0;
                for (int i = j + lane; i < m; i += 64) sm = sm + cj[i] * ck[i];
                sm = wave_reduce_sum(sm);
                sm = __shfl(sm, 0, 64);
                const double temp = sm / ajj;
                for (int i = j + lane; i < m; i += 64) ck[i] = ck[i] - temp * cj[i];
                double rk = rdiag[k];
                if (rk != 0.0) {                   // wave-uniform
                    const double ajk = __shfl((lane == 0) ? ck[j] : 0.0, 0, 64);
                    const double t2 = ajk / rk;
                    rk = rk * sqrt(fmax(0.0, 1.0 - t2 * t2));
                    const double q = rk / wa[k];
                    if (!(p05 * (q * q) > NLH_EPS)) {
                        double s2 = 0.0;
                        for (int i = j + 1 + lane; i < m; i += 64) s2 = s2 + ck[i] * ck[i];
                        s2 = wave_reduce_sum(s2);
                        rk = sqrt(__shfl(s2, 0, 64));
                        if (lane == 0) wa[k] = rk;
                    }
                    if (lane == 0) rdiag[k] = rk;
                }
            }
        }
        __syncthreads();
        if (tid == 0) rdiag[j] = -ajnorm;          // :665
        __syncthreads();
    }

    // Q^T f (:241-253).  On a fallback after a rejected trial the caller's wa4 still
    // holds the rejected residual (lmpar deviation A reads its tail), so sweep a scratch.
    const bool first = (!s) || (s->inner_pass == 0);
    double *w4 = first ? (wa4all + (size_t)p * m) : (scratch_all + (size_t)p * m);
    const double *f = fall + (size_t)p * m;
    for (int i = tid; i < m; i += BS) w4[i] = f[i];
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        double *cj = a + (size_t)j * m;
        const double ajj = (j < m) ? cj[j] : 0.0;
        if (j < m && ajj != 0.0) {
            double sm = 0.0;
            for (int i = j + tid; i < m; i += BS) sm = sm + cj[i] * w4[i];
            sm = block_reduce_sum(sm, red);
            const double temp = -sm / ajj;
            for (int i = j + tid; i < m; i += BS) w4[i] = w4[i] + cj[i] * temp;
        }
        __syncthreads();
        if (tid == 0 && j < m) { cj[j] = rdiag[j]; qtf[j] = w4[j]; }
    }
    __syncthreads();
    // R for lmpar: strict upper from the factored Jacobian, diagonal = rdiag
    double *R = Rall + (size_t)p * n * n;
    for (int e = tid; e < n * n; e += BS) {
        const int i = e % n, c = e / n;
        if (i <= c && i < m) R[e] = a[(size_t)c * m + i];
    }
    for (int k = tid; k < n; k += BS) v.rdiag[(size_t)p * n + k] = rdiag[k];
    if (first && s) {
        double tq = 0.0;
        for (int i = n + tid; i < m; i += BS) tq = tq + w4[i] * w4[i];
        tq = block_reduce_sum(tq, red);
        if (tid == 0) s->tailsq = tq;
    }
    __syncthreads();
    if (standalone || !s) return;
    if (tid == 0) { s->factor_kind = 1; s->qr_count += 1; }
    if (first) {
        // redo the head from the pre-head scaling (the normal-equations head may have run)
        if (s->iter > 1 && s->head_done) {
            for (int j = tid; j < n; j += BS) v.diag[(size_t)p * n + j] = v.diag_prev[(size_t)p * n + j];
            __syncthreads();
        }
        lm_head<false>(n, R, n, ipvt, acnorm, qtf, xall + (size_t)p * n, v.diag + (size_t)p * n,
                       v.diag_prev + (size_t)p * n, s, factor, gtol, ST_QR_READY, red, nullptr);
    } else {
        if (tid == 0) s->stage = ST_QR_READY;
    }
}
