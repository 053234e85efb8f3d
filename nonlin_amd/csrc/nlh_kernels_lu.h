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

// ---------------------------------------------------------------------------
// Blocked right-looking LU (panel width LU_NB), host-driven: panel factorisation (one
// workgroup per problem), deferred row interchanges, block-row triangular solve and the
// trailing update as a tiled kernel over the whole chip.  Every element still receives
// a(i,k) -= l(i,j)*u(j,k) for j ascending with a separate multiply and subtract, and the
// deferred interchanges permute L and the trailing columns consistently, so the factors are
// bit-identical to the unblocked loop (and to the CPU restatement).
// ---------------------------------------------------------------------------
#define LU_NB 32

__global__ void __launch_bounds__(1024)
k_lu_panel(int n, double *__restrict__ Aall, int32_t *__restrict__ ipvt_all, int32_t *__restrict__ info,
           int jb, int nb)
{
    __shared__ double red[64];
    int *redi = reinterpret_cast<int *>(red + 32);
    const int p = blockIdx.x;
    const int tid = threadIdx.x, BS = blockDim.x;
    double *a = Aall + (size_t)p * n * n;
    int32_t *ipvt = ipvt_all + (size_t)p * n;
    for (int j = jb; j < jb + nb; ++j) {
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
            if (piv != j) {                       // interchange inside the panel; the rest is deferred
                for (int k = jb + tid; k < jb + nb; k += BS) {
                    double *ck = a + (size_t)k * n;
                    const double t = ck[j]; ck[j] = ck[piv]; ck[piv] = t;
                }
                __syncthreads();
            }
            const double rcp = 1.0 / cj[j];
            __syncthreads();
            for (int i = j + 1 + tid; i < n; i += BS) cj[i] = cj[i] * rcp;
        } else if (tid == 0 && info && info[p] == 0) {
            info[p] = j + 1;
        }
        __syncthreads();
        const int nk = jb + nb - (j + 1);          // remaining panel columns
        if (nk > 0) {
            // rows i > j of the remaining panel columns: thread = row, loop over the (few) columns
            for (int i = j + 1 + tid; i < n; i += BS) {
                const double lij = cj[i];
                for (int k = j + 1; k < jb + nb; ++k) {
                    double *ck = a + (size_t)k * n;
                    ck[i] = ck[i] - lij * ck[j];
                }
            }
        }
        __syncthreads();
    }
}

// The same panel factorisation with the panel resident in LDS (rows jb .. n-1 of nb <= LU_PNB columns): the column
// steps are LDS round trips and barriers instead of global-memory round trips.  Identical operation sequence per
// element (pivot search, in-panel interchange, reciprocal scaling, a(i,k) -= l(i) u(j,k) for j ascending), so the
// factors stay bit-identical to the unblocked loop.  One workgroup per problem, thread per row of the panel;
// rows = n - jb <= LU_PROWS.  Dynamic LDS: nb * rows doubles.
#define LU_PNB 16
#define LU_PROWS 1024
__global__ void __launch_bounds__(1024)
k_lu_panel_lds(int n, double *__restrict__ Aall, int32_t *__restrict__ ipvt_all, int32_t *__restrict__ info,
               int jb, int nb)
{
    extern __shared__ double pan[];                // pan[c * rows + r]: column jb + c, row jb + r
    __shared__ double red[64];
    int *redi = reinterpret_cast<int *>(red + 32);
    const int p = blockIdx.x, tid = threadIdx.x, BS = blockDim.x;
    const int rows = n - jb;
    double *a = Aall + (size_t)p * n * n;
    int32_t *ipvt = ipvt_all + (size_t)p * n;
    for (int r = tid; r < rows; r += BS) {         // all nb loads of a thread are in flight together
        double t[LU_PNB];
#pragma unroll
        for (int c = 0; c < LU_PNB; ++c) t[c] = (c < nb) ? a[(size_t)(jb + c) * n + jb + r] : 0.0;
#pragma unroll
        for (int c = 0; c < LU_PNB; ++c) if (c < nb) pan[c * rows + r] = t[c];
    }
    __syncthreads();
    for (int c = 0; c < nb; ++c) {                 // panel column c = matrix column j = jb + c, pivot row >= c
        double *cj = pan + c * rows;
        double bv = 0.0;
        int bk = 0x7fffffff;
        for (int r = c + tid; r < rows; r += BS) {
            const double v = fabs(cj[r]);
            if (bk == 0x7fffffff || v > bv) { bv = v; bk = r; }
        }
        const int piv = block_argmax_first(bv, bk, red, redi);     // row index within the panel
        const double apj = cj[piv];
        __syncthreads();
        if (tid == 0) ipvt[jb + c] = jb + piv;
        if (apj != 0.0) {
            if (piv != c) {                        // interchange inside the panel; the rest of the matrix is deferred
                for (int k = tid; k < nb; k += BS) {
                    double *ck = pan + k * rows;
                    const double t = ck[c]; ck[c] = ck[piv]; ck[piv] = t;
                }
                __syncthreads();
            }
            const double rcp = 1.0 / cj[c];        // row c is not touched below: no barrier needed before the scaling
            for (int r = c + 1 + tid; r < rows; r += BS) {
                const double lij = cj[r] * rcp;     // scale, then the rank-one update of the rest of the panel row
                cj[r] = lij;
                for (int k = c + 1; k < nb; ++k) {
                    double *ck = pan + k * rows;
                    ck[r] = ck[r] - lij * ck[c];
                }
            }
        } else {
            if (tid == 0 && info && info[p] == 0) info[p] = jb + c + 1;
            for (int r = c + 1 + tid; r < rows; r += BS) {          // zero pivot: no scaling, the update still runs (l = column as is)
                const double lij = cj[r];
                for (int k = c + 1; k < nb; ++k) {
                    double *ck = pan + k * rows;
                    ck[r] = ck[r] - lij * ck[c];
                }
            }
        }
        __syncthreads();
    }
    for (int r = tid; r < rows; r += BS) {
#pragma unroll
        for (int c = 0; c < LU_PNB; ++c) if (c < nb) a[(size_t)(jb + c) * n + jb + r] = pan[c * rows + r];
    }
}

// Deferred row interchanges of panel [jb, jb+nb) applied to every column outside the panel.
__global__ void __launch_bounds__(256)
k_lu_swap(int n, double *__restrict__ Aall, const int32_t *__restrict__ ipvt_all, int jb, int nb)
{
    const int p = blockIdx.y;
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n - nb) return;
    if (k >= jb) k += nb;                          // skip the panel's own columns
    double *ck = Aall + (size_t)p * n * n + (size_t)k * n;
    const int32_t *ipvt = ipvt_all + (size_t)p * n;
    for (int j = jb; j < jb + nb; ++j) {
        const int q = ipvt[j];
        if (q != j) { const double t = ck[j]; ck[j] = ck[q]; ck[q] = t; }
    }
}

// Block row: for every column k right of the panel, rows jb..jb+nb: u(j,k) final after the updates
// of the earlier panel columns (unit lower triangular solve, j ascending).
__global__ void __launch_bounds__(256)
k_lu_trsm(int n, double *__restrict__ Aall, int jb, int nb)
{
    __shared__ double L11[LU_NB * LU_NB];          // L11[i + j*LU_NB], i > j used
    const int p = blockIdx.y;
    double *a = Aall + (size_t)p * n * n;
    for (int e = threadIdx.x; e < nb * nb; e += blockDim.x) {
        const int i = e % nb, j = e / nb;
        L11[i + j * LU_NB] = a[(size_t)(jb + j) * n + jb + i];
    }
    __syncthreads();
    const int k = jb + nb + blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double *ck = a + (size_t)k * n + jb;
    double u[LU_NB];
#pragma unroll
    for (int i = 0; i < LU_NB; ++i) u[i] = (i < nb) ? ck[i] : 0.0;
#pragma unroll
    for (int j = 0; j < LU_NB; ++j) {
        if (j < nb) {
            const double uj = u[j];
#pragma unroll
            for (int i = 0; i < LU_NB; ++i)
                if (i > j && i < nb) u[i] = u[i] - L11[i + j * LU_NB] * uj;
        }
    }
#pragma unroll
    for (int i = 0; i < LU_NB; ++i)
        if (i < nb) ck[i] = u[i];
}

// Trailing update A22 -= L21 U12 on 64x64 tiles; each thread owns a 4x4 register tile and
// subtracts the nb products in j order (separate multiply and subtract).
__global__ void __launch_bounds__(256)
k_lu_gemm(int n, double *__restrict__ Aall, int jb, int nb)
{
    __shared__ double Ls[LU_NB * 64];              // Ls[j*64 + r]
    __shared__ double Us[LU_NB * 64];              // Us[j*64 + c]
    const int p = blockIdx.z;
    double *a = Aall + (size_t)p * n * n;
    const int t0 = jb + nb;
    const int r0 = t0 + blockIdx.x * 64, c0 = t0 + blockIdx.y * 64;
    const int tid = threadIdx.x;
    for (int e = tid; e < nb * 64; e += 256) {
        const int j = e >> 6, q = e & 63;
        Ls[e] = (r0 + q < n) ? a[(size_t)(jb + j) * n + r0 + q] : 0.0;     // L21(r, j), contiguous in r
        Us[e] = (c0 + q < n) ? a[(size_t)(c0 + q) * n + jb + j] : 0.0;     // U12(j, c)
    }
    __syncthreads();
    const int tr = (tid & 15) * 4, tc = (tid >> 4) * 4;
    double acc[4][4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = r0 + tr + rr, c = c0 + tc + cc;
            acc[cc][rr] = (r < n && c < n) ? a[(size_t)c * n + r] : 0.0;
        }
    for (int j = 0; j < nb; ++j) {
        double l[4], u[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { l[q] = Ls[j * 64 + tr + q]; u[q] = Us[j * 64 + tc + q]; }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) acc[cc][rr] = acc[cc][rr] - l[rr] * u[cc];
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = r0 + tr + rr, c = c0 + tc + cc;
            if (r < n && c < n) a[(size_t)c * n + r] = acc[cc][rr];
        }
}
