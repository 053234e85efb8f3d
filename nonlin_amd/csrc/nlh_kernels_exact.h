// nlh_kernels_exact.h -- start of an exact-policy solve (policy NLH_FACTOR_EXACT): fnorm = NORM2(fvec) in the
// reference's operation order.  The factorisation itself (lmfactor + Q^T f, src/nonlin_least_squares.f90:569-667,
// :241-253) lives in nlh_qrx.hip; lmpar / lmsolve and the trust-region update in nlh_kernels_lm.h (EXACT = true).
#pragma once
#include "nlh_common.h"
#include "nlh_kernels_factor.h"

// Column-major m-by-n  ->  row-major m-by-n with row stride ld, 32x32 tiles through LDS (the Householder kernels of
// the quasi-Newton and bounded least-squares paths work on row-major copies).
static __global__ void __launch_bounds__(256)
k_transpose(int m, int n, const double *__restrict__ J, double *__restrict__ Jt, int ld,
            const LmState *__restrict__ st, int want_stage)
{
    __shared__ double tile[32][33];
    const int p = blockIdx.z;
    if (st && st[p].stage != want_stage) return;
    const double *Jp = J + (size_t)p * m * n;
    double *Tp = Jt + (size_t)p * m * ld;
    const int i0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int r = ty; r < 32; r += 8) {                            // r = column offset, tx = row offset
        const int i = i0 + tx, k = k0 + r;
        tile[r][tx] = (i < m && k < n) ? Jp[(size_t)k * m + i] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {                            // r = row offset, tx = column offset
        const int i = i0 + r, k = k0 + tx;
        if (i < m && k < n) Tp[(size_t)i * ld + k] = tile[tx][r];
    }
}

// fnorm = NORM2(fvec) in reference order for every problem (:213), counters reset.
static __global__ void __launch_bounds__(256)
k_lm_init_exact(int m, const double *__restrict__ fvec, LmState *__restrict__ st, int first_stage)
{
    __shared__ double scratch[3 * NLH_NCH + 8];
    const int p = blockIdx.x;
    const double *f = fvec + (size_t)p * m;
    const double fn = norm2_flang_block([&](int i) { return f[i]; }, m, scratch);
    if (threadIdx.x == 0) {
        LmState s;
        memset(&s, 0, sizeof s);
        s.fnorm = fn;
        s.neval = 1;
        s.iter = 1;
        s.par = 0.0;
        s.stage = first_stage;
        st[p] = s;
    }
}

