// nlh_lm_head.h -- the per-problem n-vectors of the LM driver and the outer-loop head of lss_solve
// (src/nonlin_least_squares.f90:229-238 first-iteration scaling, :256-267 scaled gradient norm,
// :270-278 gradient convergence / rescale), shared by every factorisation kernel.
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
                        double gtol, int ready_stage, double *red, double *scratch, double *qs = nullptr)
{   // qs: optional n doubles of LDS for the quotients qtf(i) / fnorm of the gradient test (each thread otherwise divides
    // again for every term of its sum: n^2 / 2 divisions; the same quotients either way, hence the same bits)
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
    if (fnorm != 0.0 && qs) {
        for (int i = tid; i < n; i += BS) qs[i] = qtf[i] / fnorm;
        __syncthreads();
        for (int j = tid; j < n; j += BS) {
            const int l = ipvt[j];
            if (acnorm[l] == 0.0) continue;
            double sm = 0.0;
            for (int i = 0; i <= j; ++i) sm = sm + R[(size_t)j * ldr + i] * qs[i];
            gn = fmax(gn, fabs(sm / acnorm[l]));
        }
    } else if (fnorm != 0.0) {
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

