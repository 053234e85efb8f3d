// nlh_kernels_lu.h -- partial-pivoting LU and its solve: the device stand-in for linalg's
// lu_factor / solve_lu used by newton_solver (call sites src/nonlin_solve.f90:570, 577).
// Right-looking, unblocked, first-maximum pivoting, reciprocal column scaling: every matrix
// element sees exactly the same sequence of operations as the CPU restatement, so the
// factors and the solution are bit-identical to it (no reductions are involved except the
// exact pivot search).  One workgroup per problem; a wave owns a trailing column per step.
#pragma once
#include "nlh_common.h"

static __global__ void __launch_bounds__(1024)
k_lu_factor(int n, double *__restrict__ Aall, int32_t *__restrict__ ipvt_all, int32_t *__restrict__ info,
            const LmState *__restrict__ st, int want)
{
    __shared__ double red[64];
    int *redi = reinterpret_cast<int *>(red + 32);
    const int p = blockIdx.x;
    if (st && st[p].stage != want) return;                       // lock-step batches: only problems in this stage
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
static __global__ void __launch_bounds__(1024)
k_lu_solve(int n, const double *__restrict__ LUall, const int32_t *__restrict__ ipvt_all,
           double *__restrict__ ball, const LmState *__restrict__ st, int want)
{
    extern __shared__ double bs[];
    const int p = blockIdx.x, tid = threadIdx.x, BS = blockDim.x;
    if (st && st[p].stage != want) return;                       // lock-step batches: only problems in this stage
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
    // Each thread owns row tid (+ BS, ...).  Its entries of the next LUS_PF columns are fetched while the current
    // LUS_PF steps run, so a step costs a barrier and an LDS update, not a trip to L2 (one column ahead was not
    // enough: a step is shorter than the latency).
    constexpr int PF = 8;
    {
        double cur[PF], nxt[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) nxt[u] = (u < n && tid > u && tid < n) ? a[(size_t)u * n + tid] : 0.0;
        for (int j0 = 0; j0 < n; j0 += PF) {                       // L y = P b (unit diagonal)
#pragma unroll
            for (int u = 0; u < PF; ++u) cur[u] = nxt[u];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int jn = j0 + PF + u;
                nxt[u] = (jn < n && tid > jn && tid < n) ? a[(size_t)jn * n + tid] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int j = j0 + u;
                if (j < n) {                                       // uniform
                    const double bj = bs[j];
                    if (bj != 0.0) {
                        const double *cj = a + (size_t)j * n;
                        if (tid > j && tid < n) bs[tid] = bs[tid] - bj * cur[u];
                        for (int i = tid + BS; i < n; i += BS)
                            if (i > j) bs[i] = bs[i] - bj * cj[i];
                    }
                    __syncthreads();
                }
            }
        }
    }
    {
        double cur[PF], nxt[PF], dcur[PF], dnxt[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int j = n - 1 - u;
            nxt[u] = (j >= 0 && tid < j) ? a[(size_t)j * n + tid] : 0.0;
            dnxt[u] = (j >= 0) ? a[(size_t)j * n + j] : 1.0;
        }
        for (int j0 = n - 1; j0 >= 0; j0 -= PF) {                  // U x = y
#pragma unroll
            for (int u = 0; u < PF; ++u) { cur[u] = nxt[u]; dcur[u] = dnxt[u]; }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int jn = j0 - PF - u;
                nxt[u] = (jn >= 0 && tid < jn) ? a[(size_t)jn * n + tid] : 0.0;
                dnxt[u] = (jn >= 0) ? a[(size_t)jn * n + jn] : 1.0;
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int j = j0 - u;
                if (j >= 0) {                                      // uniform
                    const double bjr = bs[j];
                    if (bjr != 0.0) {
                        const double *cj = a + (size_t)j * n;
                        const double bj = bjr / dcur[u];
                        __syncthreads();
                        if (tid < j) bs[tid] = bs[tid] - bj * cur[u];
                        for (int i = tid + BS; i < j; i += BS) bs[i] = bs[i] - bj * cj[i];
                        if (tid == 0) bs[j] = bj;
                    }
                    __syncthreads();
                }
            }
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

static __global__ void __launch_bounds__(1024)
k_lu_panel(int n, double *__restrict__ Aall, int32_t *__restrict__ ipvt_all, int32_t *__restrict__ info,
           int jb, int nb, const LmState *__restrict__ st, int want)
{
    __shared__ double red[64];
    int *redi = reinterpret_cast<int *>(red + 32);
    const int p = blockIdx.x;
    if (st && st[p].stage != want) return;                       // lock-step batches: only problems in this stage
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
// columns (rows = n - jb <= LU_PROWS = blockDim).  A column step is: the first maximum of |a(r, c)| -- DPP reductions
// inside every wave (no LDS round trips), then 16-lane DPP reductions of the per-wave results --, the pivot row
// (with the reciprocal of the pivot) and row c published through LDS, the two owners exchanging rows, then every
// thread scales its multiplier and updates the rest of its row -- two barriers per column, no global or LDS traffic
// for the panel body.  What a step costs is the instruction count of sixteen waves on four SIMDs: with the arg-max
// through six shuffle levels, every thread scanning the sixteen per-wave results and every thread dividing by the
// pivot a step took 3.3 us (n = 1024); in this form ~1 us.  Identical operation sequence per element (pivot search,
// in-panel interchange, reciprocal scaling, a(i,k) -= l(i) u(j,k) for j ascending): bit-identical to the unblocked loop.
#define LU_PNB 16
#define LU_PROWS 1024

// First maximum = the largest value, and among the lanes that hold it the smallest index: a max-reduction of the
// values followed by a min-reduction of the candidate indices (three and two instructions per DPP step instead of
// the dozen a (value, index) pair comparison takes).  Lanes without a candidate pass value -1 and index 0x7fffffff.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double lu_dpp_max_step(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, BANK_MASK, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, BANK_MASK, false);
    return fmax(v, __hiloint2double(ohi, olo));
}
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int lu_dpp_min_step(int i)
{
    return min(i, __builtin_amdgcn_update_dpp(i, i, CTRL, ROW_MASK, BANK_MASK, false));
}
// reductions over a 16-lane row (result in lane 15 of the row) and over the wave (result in lane 63)
__device__ __forceinline__ double lu_row16_max(double v)
{
    v = lu_dpp_max_step<0x111, 0xf, 0xf>(v);       // row_shr:1
    v = lu_dpp_max_step<0x112, 0xf, 0xf>(v);       // row_shr:2
    v = lu_dpp_max_step<0x114, 0xf, 0xf>(v);       // row_shr:4
    return lu_dpp_max_step<0x118, 0xf, 0xf>(v);    // row_shr:8
}
__device__ __forceinline__ int lu_row16_min(int i)
{
    i = lu_dpp_min_step<0x111, 0xf, 0xf>(i);
    i = lu_dpp_min_step<0x112, 0xf, 0xf>(i);
    i = lu_dpp_min_step<0x114, 0xf, 0xf>(i);
    return lu_dpp_min_step<0x118, 0xf, 0xf>(i);
}
__device__ __forceinline__ double lu_uniform_lane(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// wave-uniform (value, index) of the first maximum over the wave's lanes
__device__ __forceinline__ void lu_wave_first_max(double v, int idx, double &vmax, int &imin)
{
    double t = lu_row16_max(v);
    t = lu_dpp_max_step<0x142, 0xa, 0xf>(t);       // row_bcast:15 into rows 1 and 3
    t = lu_dpp_max_step<0x143, 0xc, 0xf>(t);       // row_bcast:31 into rows 2 and 3: lane 63 has the wave's
    vmax = lu_uniform_lane(t, 63);
    int c = (v == vmax) ? idx : 0x7fffffff;
    c = lu_row16_min(c);
    c = lu_dpp_min_step<0x142, 0xa, 0xf>(c);
    c = lu_dpp_min_step<0x143, 0xc, 0xf>(c);
    imin = __builtin_amdgcn_readlane(c, 63);
}

static __global__ void __launch_bounds__(1024)
k_lu_panel_lds(int n, double *__restrict__ Aall, int32_t *__restrict__ ipvt_all, int32_t *__restrict__ info,
               int jb, int nb, const LmState *__restrict__ st, int want)
{
    __shared__ double redv[2][16];
    __shared__ int redi[2][16];
    __shared__ double prow[2][LU_PNB + 1], crow[2][LU_PNB];      // prow[.][LU_PNB]: 1 / pivot
    const int p = blockIdx.x, r = threadIdx.x, lane = r & 63, wid = r >> 6, nw = (blockDim.x + 63) >> 6;
    if (st && st[p].stage != want) return;                       // lock-step batches: only problems in this stage
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
            // NaN entries: the ordered search keeps a NaN diagonal entry (nothing compares greater) and never takes a
            // NaN below it; the max-reduction drops NaNs, so a NaN on the diagonal enters as +Inf at the lowest index
            double v = (mine && r >= c) ? fabs(row[c]) : -1.0;
            if (r == c && v != v) v = __builtin_inf();
            int idx = (mine && r >= c) ? r : 0x7fffffff;
            double wmax;
            int widx;
            lu_wave_first_max(v, idx, wmax, widx);
            if (lane == 0) { redv[par][wid] = wmax; redi[par][wid] = widx; }
            __syncthreads();
            // the per-wave results, one per lane of a 16-lane row (every row of every wave does the same)
            const double cv = (lane & 15) < nw ? redv[par][lane & 15] : -1.0;
            const int ci = (lane & 15) < nw ? redi[par][lane & 15] : 0x7fffffff;
            const double bv = lu_uniform_lane(lu_row16_max(cv), 15);
            int piv = __builtin_amdgcn_readlane(lu_row16_min((cv == bv) ? ci : 0x7fffffff), 15);
            if (piv == 0x7fffffff) piv = c;        // (unreachable: row c is always a candidate)
            // the pivot row and row c through LDS
            if (r == piv) {
#pragma unroll
                for (int k = 0; k < LU_PNB; ++k) prow[par][k] = row[k];
                prow[par][LU_PNB] = 1.0 / row[c];                // (Inf for a zero pivot: not used then)
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
                    const double rcp = prow[par][LU_PNB];
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

// A thread per column outside the panel: the deferred row interchanges of panel [jb, jb+nb), then -- right of the
// panel -- the column's part of the block row: u(j,k) final after the updates of the earlier panel columns (unit lower
// triangular solve, j ascending).  One launch for both: these kernels are launch-latency-bound (n = 1024: 64 panels x
// three launches of 5 - 12 us each were a third of the factorisation).
static __global__ void __launch_bounds__(256)
k_lu_swap_trsm(int n, double *__restrict__ Aall, const int32_t *__restrict__ ipvt_all, int jb, int nb,
               const LmState *__restrict__ st, int want)
{
    __shared__ double L11[LU_NB * LU_NB];          // L11[i + j*LU_NB], i > j used
    __shared__ int32_t piv[LU_NB];
    const int p = blockIdx.y;
    if (st && st[p].stage != want) return;                       // lock-step batches: only problems in this stage
    double *a = Aall + (size_t)p * n * n;
    for (int e = threadIdx.x; e < nb * nb; e += blockDim.x) {
        const int i = e % nb, j = e / nb;
        L11[i + j * LU_NB] = a[(size_t)(jb + j) * n + jb + i];
    }
    if (threadIdx.x < nb) piv[threadIdx.x] = ipvt_all[(size_t)p * n + jb + threadIdx.x];
    __syncthreads();
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n - nb) return;
    if (k >= jb) k += nb;                          // skip the panel's own columns
    double *ck = a + (size_t)k * n;
    // (Measured and dropped: composing the interchanges once per workgroup and loading all touched rows before storing
    // any -- two trips to memory instead of a dependent one per interchange -- is slower: 3.44 vs 3.23 ms at n = 1024.)
    for (int j = 0; j < nb; ++j) {
        const int q = piv[j];
        if (q != jb + j) { const double t = ck[jb + j]; ck[jb + j] = ck[q]; ck[q] = t; }
    }
    if (k < jb + nb) return;                       // left of the panel: interchanges only
    ck += jb;
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
static __global__ void __launch_bounds__(256)
k_lu_gemm(int n, double *__restrict__ Aall, int jb, int nb, const LmState *__restrict__ st, int want)
{
    __shared__ double Ls[LU_NB * 64];              // Ls[j*64 + r]
    __shared__ double Us[LU_NB * 64];              // Us[j*64 + c]
    const int p = blockIdx.z;
    if (st && st[p].stage != want) return;                       // lock-step batches: only problems in this stage
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
