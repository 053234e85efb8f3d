// nlh_kernels_lu.h -- partial-pivoting LU and its solve: the device stand-in for linalg's
// lu_factor / solve_lu used by newton_solver (call sites src/nonlin_solve.f90:570, 577).
// Right-looking, unblocked, first-maximum pivoting, reciprocal column scaling: every matrix
// element sees exactly the same sequence of operations as the CPU restatement, so the
// factors and the solution are bit-identical to it (no reductions are involved except the
// exact pivot search).  One workgroup per problem; a wave owns a trailing column per step.
#pragma once
#include "nlh_common.h"

__global__ void __launch_bounds__(1024)
k_lu_factor(int n, double *__restrict__ Aall, int32_t *__restrict__ ipvt_all, int32_t *__restrict__ info)
{
    __shared__ double red[64];
    int *redi = reinterpret_cast<int *>(red + 32);
    const int p = blockIdx.x;
    const int tid = threadIdx.x, BS = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = BS >> 6;
    double *a = Aall + (size_t)p * n * n;
    int32_t *ipvt = ipvt_all + (size_t)p * n;
    int inf = 0;
    for (int j = 0; j < n; ++j) {
        double *cj = a + (size_t)j * n;
        double bv = 0.0;
        int bk = 0x7fffffff;
        for (int i = j + tid; i < n; i += BS) {
            const double v = fabs(cj[i]);
            if (bk == 0x7fffffff || v > bv) { bv = v; bk = i; }
        }
        const int piv = block_argmax_first(bv, bk, red, redi);
        const double apj = cj[piv];
        __syncthreads();
        if (tid == 0) ipvt[j] = piv;
        if (apj != 0.0) {
            if (piv != j) {
                for (int k = tid; k < n; k += BS) {
                    double *ck = a + (size_t)k * n;
                    const double t = ck[j]; ck[j] = ck[piv]; ck[piv] = t;
                }
                __syncthreads();
            }
            const double rcp = 1.0 / cj[j];
            __syncthreads();
            for (int i = j + 1 + tid; i < n; i += BS) cj[i] = cj[i] * rcp;
        } else if (inf == 0) {
            inf = j + 1;
        }
        __syncthreads();
        for (int k = j + 1 + wid; k < n; k += nw) {
            double *ck = a + (size_t)k * n;
            const double ujk = ck[j];
            for (int i = j + 1 + lane; i < n; i += 64) ck[i] = ck[i] - cj[i] * ujk;
        }
        __syncthreads();
    }
    if (tid == 0 && info) info[p] = inf;
}

// Solve LU x = b in place (dynamic LDS: n doubles).
__global__ void __launch_bounds__(1024)
k_lu_solve(int n, const double *__restrict__ LUall, const int32_t *__restrict__ ipvt_all,
           double *__restrict__ ball)
{
    extern __shared__ double bs[];
    const int p = blockIdx.x, tid = threadIdx.x, BS = blockDim.x;
    const double *a = LUall + (size_t)p * n * n;
    const int32_t *ipvt = ipvt_all + (size_t)p * n;
    double *b = ball + (size_t)p * n;
    for (int i = tid; i < n; i += BS) bs[i] = b[i];
    __syncthreads();
    if (tid == 0)
        for (int j = 0; j < n; ++j) {
            const int q = ipvt[j];
            if (q != j) { const double t = bs[j]; bs[j] = bs[q]; bs[q] = t; }
        }
    __syncthreads();
    for (int j = 0; j < n; ++j) {                 // L y = P b (unit diagonal)
        const double bj = bs[j];
        if (bj != 0.0) {
            const double *cj = a + (size_t)j * n;
            for (int i = j + 1 + tid; i < n; i += BS) bs[i] = bs[i] - bj * cj[i];
        }
        __syncthreads();
    }
    for (int j = n - 1; j >= 0; --j) {            // U x = y
        const double bjr = bs[j];
        if (bjr != 0.0) {
            const double *cj = a + (size_t)j * n;
            const double bj = bjr / cj[j];
            __syncthreads();
            for (int i = tid; i < j; i += BS) bs[i] = bs[i] - bj * cj[i];
            if (tid == 0) bs[j] = bj;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += BS) b[i] = bs[i];
}
