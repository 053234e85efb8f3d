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
    // Each thread owns row tid (+ BS, ...).  Its entry of the next column is fetched before the barrier of the
    // current step, so a step costs a barrier and an LDS update rather than a global load issued after it.
    {
        double pre = (tid > 0 && tid < n) ? a[tid] : 0.0;          // column 0
        for (int j = 0; j < n; ++j) {                              // L y = P b (unit diagonal)
            const double cur = pre;
            if (j + 1 < n) pre = (tid > j + 1 && tid < n) ? a[(size_t)(j + 1) * n + tid] : 0.0;
            const double bj = bs[j];
            if (bj != 0.0) {
                const double *cj = a + (size_t)j * n;
                if (tid > j && tid < n) bs[tid] = bs[tid] - bj * cur;
                for (int i = tid + BS; i < n; i += BS)
                    if (i > j) bs[i] = bs[i] - bj * cj[i];
            }
            __syncthreads();
        }
    }
    {
        const int jl = n - 1;
        double pre = (tid < jl) ? a[(size_t)jl * n + tid] : 0.0, dpre = a[(size_t)jl * n + jl];
        for (int j = n - 1; j >= 0; --j) {                         // U x = y
            const double cur = pre, dcur = dpre;
            if (j > 0) {
                pre = (tid < j - 1) ? a[(size_t)(j - 1) * n + tid] : 0.0;
                dpre = a[(size_t)(j - 1) * n + j - 1];
            }
            const double bjr = bs[j];
            if (bjr != 0.0) {
                const double *cj = a + (size_t)j * n;
                const double bj = bjr / dcur;
                __syncthreads();
                if (tid < j) bs[tid] = bs[tid] - bj * cur;
                for (int i = tid + BS; i < j; i += BS) bs[i] = bs[i] - bj * cj[i];
                if (tid == 0) bs[j] = bj;
            }
            __syncthreads();
        }
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

// The same panel factorisation with the panel held in registers: thread r owns row jb + r of the nb <= LU_PNB panel
// columns (rows = n - jb <= LU_PROWS = blockDim).  A column step is: wave arg-max of |a(r, c)| and a 16-entry scan of
// the per-wave results (first maximum), the pivot row and row c published through LDS, the two owners exchanging
// rows, then every thread scales its multiplier and updates the rest of its row -- two barriers per column, no
// global or LDS traffic for the panel body.  Identical operation sequence per element (pivot search, in-panel
// interchange, reciprocal scaling, a(i,k) -= l(i) u(j,k) for j ascending): bit-identical to the unblocked loop.
#define LU_PNB 16
#define LU_PROWS 1024
__global__ void __launch_bounds__(1024)
k_lu_panel_lds(int n, double *__restrict__ Aall, int32_t *__restrict__ ipvt_all, int32_t *__restrict__ info,
               int jb, int nb)
{
    __shared__ double redv[2][16];
    __shared__ int redi[2][16];
    __shared__ double prow[2][LU_PNB], crow[2][LU_PNB];
    const int p = blockIdx.x, r = threadIdx.x, lane = r & 63, wid = r >> 6, nw = (blockDim.x + 63) >> 6;
    const int rows = n - jb;
    double *a = Aall + (size_t)p * n * n;
    int32_t *ipvt = ipvt_all + (size_t)p * n;
    const bool mine = r < rows;
    double row[LU_PNB];
#pragma unroll
    for (int k = 0; k < LU_PNB; ++k) row[k] = (mine && k < nb) ? a[(size_t)(jb + k) * n + jb + r] : 0.0;
#pragma unroll
    for (int c = 0; c < LU_PNB; ++c) {
        if (c < nb) {                              // uniform
            const int par = c & 1;
            // first maximum of |a(r, c)| over r >= c
            double v = (mine && r >= c) ? fabs(row[c]) : -1.0;
            int idx = (mine && r >= c) ? r : 0x7fffffff;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = __shfl_down(v, off, 64);
                const int oi = __shfl_down(idx, off, 64);
                const bool take = (oi != 0x7fffffff) && (idx == 0x7fffffff || ov > v || (ov == v && oi < idx));
                if (take) { v = ov; idx = oi; }
            }
            if (lane == 0) { redv[par][wid] = v; redi[par][wid] = idx; }
            __syncthreads();
            double bv = redv[par][0];
            int piv = redi[par][0];
            for (int w = 1; w < nw; ++w) {
                const double ov = redv[par][w];
                const int oi = redi[par][w];
                const bool take = (oi != 0x7fffffff) && (piv == 0x7fffffff || ov > bv || (ov == bv && oi < piv));
                if (take) { bv = ov; piv = oi; }
            }
            // the pivot row and row c through LDS
            if (r == piv) {
#pragma unroll
                for (int k = 0; k < LU_PNB; ++k) prow[par][k] = row[k];
            }
            if (r == c) {
#pragma unroll
                for (int k = 0; k < LU_PNB; ++k) crow[par][k] = row[k];
            }
            __syncthreads();
            const double apj = prow[par][c];
            if (r == 0) ipvt[jb + c] = jb + piv;
            if (apj != 0.0) {
                if (piv != c) {                    // interchange inside the panel; the rest of the matrix is deferred
                    if (r == c) {
#pragma unroll
                        for (int k = 0; k < LU_PNB; ++k) row[k] = prow[par][k];
                    } else if (r == piv) {
#pragma unroll
                        for (int k = 0; k < LU_PNB; ++k) row[k] = crow[par][k];
                    }
                }
                if (mine && r > c) {
                    const double rcp = 1.0 / apj;
                    const double lij = row[c] * rcp;
                    row[c] = lij;
#pragma unroll
                    for (int k = c + 1; k < LU_PNB; ++k) row[k] = row[k] - lij * prow[par][k];
                }
            } else {
                if (r == 0 && info && info[p] == 0) info[p] = jb + c + 1;
                if (mine && r > c) {               // zero pivot: no interchange, no scaling; the update still runs
                    const double lij = row[c];
#pragma unroll
                    for (int k = c + 1; k < LU_PNB; ++k) row[k] = row[k] - lij * crow[par][k];
                }
            }
        }
    }
    if (mine) {
#pragma unroll
        for (int k = 0; k < LU_PNB; ++k)
            if (k < nb) a[(size_t)(jb + k) * n + jb + r] = row[k];
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
