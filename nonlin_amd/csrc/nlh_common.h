// nlh_common.h -- shared device helpers for the gfx950 kernels (wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NLH_WAVE 64

// Per-problem scalar state of the batched Levenberg-Marquardt driver
// (the locals of lss_solve, src/nonlin_least_squares.f90:152-160).
enum NlhStage : int32_t {
    ST_NEED_JAC = 0,   // outer-loop head: Jacobian + factorisation due
    ST_HAVE_JAC = 1,   // J formed, factorisation due
    ST_NE_READY = 2,   // Gram/Cholesky factors valid, lmpar due
    ST_NEED_QR = 3,    // Householder QR due (policy, ill-conditioning, or GN step rejected)
    ST_QR_READY = 4,   // QR factors valid, lmpar due
    ST_TRIAL_READY = 5,// trial point in wa2, residual evaluation due
    ST_TRIAL_DONE = 6, // residual at trial point in wa4, update due
    ST_DONE = 7
};

struct LmState {
    double fnorm, fnorm1, par, delta, xnorm, gnorm, pnorm;
    double temp1n;   // || R P^T p ||  (src/nonlin_least_squares.f90:307-313)
    double tailsq;   // sum of squares of wa4(n+1:m) (lmpar deviation A, :531)
    int32_t iter, neval, njac;
    int32_t stage;
    int32_t factor_kind;   // 0 = normal equations, 1 = Householder QR (this outer iteration)
    int32_t inner_pass;    // number of lmpar calls already made in this outer iteration
    int32_t fcnvrg, xcnvrg, gcnvrg;
    int32_t flag;          // NL_* failure flag (:358-363)
    int32_t qr_count;      // diagnostics: how many QR fallbacks happened
    int32_t head_done;     // the outer-loop head already ran in this outer iteration
};

__device__ __forceinline__ double wave_reduce_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = v + __shfl_down(v, off, 64);
    return v;   // valid in lane 0
}

__device__ __forceinline__ double wave_reduce_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}

// Deterministic block-wide sum; result broadcast to all threads.
// sh must hold at least blockDim.x/64 + 1 doubles.  Two barriers.
__device__ __forceinline__ double block_reduce_sum(double v, double *sh)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    v = wave_reduce_sum(v);
    __syncthreads();            // protect sh from the previous use
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    double r = 0.0;
    for (int w = 0; w < nw; ++w) r = r + sh[w];   // fixed order, every thread
    return r;
}

__device__ __forceinline__ double block_reduce_max(double v, double *sh)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    v = wave_reduce_max(v);
    __syncthreads();
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    double r = sh[0];
    for (int w = 1; w < nw; ++w) r = fmax(r, sh[w]);
    return r;
}

// First index of the maximum of v over the block (strict '>' => first max wins,
// as in lmfactor's pivot search, src/nonlin_least_squares.f90:622-625).
// Each thread passes its best (value, index) with the smallest index among ties;
// threads with no candidate pass idx = INT_MAX and any value.
__device__ __forceinline__ int block_argmax_first(double v, int idx, double *shv, int *shi)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double ov = __shfl_down(v, off, 64);
        int oi = __shfl_down(idx, off, 64);
        bool take = (oi != 0x7fffffff) && (idx == 0x7fffffff || ov > v || (ov == v && oi < idx));
        if (take) { v = ov; idx = oi; }
    }
    __syncthreads();
    if (lane == 0) { shv[wid] = v; shi[wid] = idx; }
    __syncthreads();
    double bv = shv[0];
    int bi = shi[0];
    for (int w = 1; w < nw; ++w) {
        double ov = shv[w];
        int oi = shi[w];
        bool take = (oi != 0x7fffffff) && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi));
        if (take) { bv = ov; bi = oi; }
    }
    return bi;
}

#define NLH_SQRT_EPS 1.4901161193847656e-08   /* sqrt(epsilon(1d0)), multi_eqn_mult_var.f90:263-264 */
#define NLH_EPS      2.220446049250313e-16
#define NLH_DWARF    2.2250738585072014e-308  /* tiny(1d0), least_squares.f90:442 */

// Forward-difference step of vfh_jac_fcn (src/nonlin_multi_eqn_mult_var.f90:268-270).
__device__ __forceinline__ double fd_step(double xj)
{
    double h = NLH_SQRT_EPS * fabs(xj);
    if (h == 0.0) h = NLH_SQRT_EPS;
    return h;
}
