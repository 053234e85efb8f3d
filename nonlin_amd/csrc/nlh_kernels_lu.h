// nlh_kernels_lu.h -- partial-pivoting LU and its solve: the device stand-in for linalg's
// lu_factor / solve_lu used by newton_solver (call sites src/nonlin_solve.f90:570, 577).
// Right-looking, unblocked, first-maximum pivoting, reciprocal column scaling: every matrix
// element sees exactly the same sequence of operations as the CPU restatement, so the
// factors and the solution are bit-identical to it (no reductions are involved except the
// exact pivot search).  One workgroup per problem; a wave owns a trailing column per step.
#pragma once
#include "nlh_common.h"
#define LU_PIN(x) asm volatile("" : "+v"(x))      // keeps a load where it was written (the compiler moves loads back under conditions)

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
            double v = fabs(cj[i]);
            if (v != v) v = (i == j) ? __builtin_inf() : -1.0;    // the ordered search keeps a NaN diagonal entry and never takes a NaN below it
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

// Solve LU x = b in place (dynamic LDS: lu_solve_lds(n)).
//
// Round 4: BLOCKED substitution.  The straightforward form is 2 n barrier-separated steps of a sixteen-wave workgroup
// (n = 1024: 0.85 ms, 0.4 us a step) after one thread has walked the n interchanges.  Here
//  * every element of b traces its own way through the interchanges (position i is touched by step j only if j == i or
//    ipvt(j) == i, and an element that has been moved into position j < current step is final): all rows in parallel;
//  * columns are taken LUS_W at a time: ONE wave solves the LUS_W x LUS_W triangle of the block out of LDS -- the solved
//    entry of step j goes to the other lanes with a v_readlane, no barrier --, then every thread applies the block's
//    LUS_W solved entries to its own row outside the block, its column entries prefetched before the triangle was solved.
// Every element of b still receives b(i) = b(i) - x(j) * a(i,j) for j ascending (L) / descending (U), skipped where the
// sequential loop skips (an exactly zero b(j)), and x(j) = b(j) / u(j,j) is the same IEEE division: the same bits.
#define LUS_W 16
static inline size_t lu_solve_lds(int n) { return sizeof(double) * ((size_t)n + 2 * LUS_W * LUS_W) + sizeof(int32_t) * ((size_t)n + 8 + 4); }

static __global__ void __launch_bounds__(1024)
k_lu_solve(int n, const double *__restrict__ LUall, const int32_t *__restrict__ ipvt_all,
           double *__restrict__ ball, const LmState *__restrict__ st, int want)
{
    constexpr int W = LUS_W;
    extern __shared__ double bs[];
    double *tri = bs + n;                                         // two buffers of tri[j * W + l] = a(jb + l, jb + j)
    int32_t *ips = reinterpret_cast<int32_t *>(tri + 2 * W * W);
    int32_t *flag = ips + n + 8;                                  // which of the block's back-substitution steps ran
    const int p = blockIdx.x, tid = threadIdx.x, BS = blockDim.x, lane = tid & 63, wid = tid >> 6;
    if (st && st[p].stage != want) return;                       // lock-step batches: only problems in this stage
    const double *a = LUall + (size_t)p * n * n;
    const int32_t *ipvt = ipvt_all + (size_t)p * n;
    double *b = ball + (size_t)p * n;
    // The loads of block k + 1 (this thread's row entries of its columns, its element of the triangle) are issued
    // before block k is worked on, and the barriers in between wait for LDS only: the memory latency is paid once.
    // All loads are unconditional with clamped indices (a load under a condition is waited for where it stands).
    // Blocks are numbered along the whole sweep: 0 .. nblk - 1 the forward substitution (columns k * W ...), nblk ..
    // 2 nblk - 1 the back substitution (columns (2 nblk - 1 - k) * W ...).  The loads of block k + 1 (this thread's row
    // entries of its columns, its element of the triangle) are issued before block k is worked on, and the barriers in
    // between wait for LDS only.  (Two blocks ahead: no faster -- a block costs its instruction chains, 2 - 3 us, not
    // the memory latency.)  All loads are unconditional with clamped indices, uniform column base + 32-bit row offset.
    const int nblk = (n + W - 1) / W;
    double pf[W], pf1[W], t1 = 0.0;
    auto issue = [&](int k) {
        const int kk = k < 2 * nblk ? k : 2 * nblk - 1;
        const bool fwd = kk < nblk;
        const int jb = (fwd ? kk : 2 * nblk - 1 - kk) * W;
        const int row = fwd ? jb + W + tid : tid;                 // this thread's first row outside the block
        const int w = n - jb < W ? n - jb : W;
        const unsigned rc = row < n ? row : n - 1;
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const double *col = a + (size_t)(jb + (j < w ? j : w - 1)) * n;      // (uniform)
            pf1[j] = col[rc];
        }
        const unsigned e = tid < W * W ? tid : 0, j = e / W, l = e % W;
        t1 = a[(size_t)(jb + (j < (unsigned)w ? j : w - 1)) * n + jb + (l < (unsigned)w ? l : w - 1)];
    };
    auto land = [&](int buf) {                                    // the issued loads arrive: triangle to LDS, row entries to pf
        LU_PIN(t1);
        if (tid < W * W) tri[buf * W * W + tid] = t1;
#pragma unroll
        for (int j = 0; j < W; ++j) { LU_PIN(pf1[j]); pf[j] = pf1[j]; }
    };
#ifdef LU_DBG_CLK
    long long sc[6]; sc[0] = wall_clock64();
#endif
    issue(0);
    for (int i = tid; i < n + 8; i += BS) ips[i] = i < n ? ipvt[i] : -1;
    __syncthreads();
    for (int i = tid; i < n; i += BS) {                           // P b: where element i ends
        const double bi = b[i];
        int cur = i;
        for (int j = 0; j < n && j <= cur; j += 8) {              // (eight interchanges per trip: one LDS latency, not eight)
            int q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = ips[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) cur = (cur == j + u) ? q[u] : ((cur == q[u]) ? j + u : cur);
        }
        bs[cur] = bi;
    }
#ifdef LU_DBG_CLK
    sc[1] = wall_clock64();
#endif
    land(0);
    int buf = 0;
    // L y = P b (unit diagonal)
    for (int k = 0; k < nblk; ++k, buf ^= 1) {
        const int jb = k * W;
        const int w = n - jb < W ? n - jb : W;
        const int i0 = jb + W + tid;
        issue(k + 1);
        nlh_lds_barrier();
        const double *tr = tri + buf * W * W;
        if (wid == 0) {
            double bl = (lane < w) ? bs[jb + lane] : 0.0;
            double trv[W];                                       // this lane's row of the triangle: all LDS reads in flight together
#pragma unroll                                                   // (read inside the steps, each one is an LDS latency on the chain)
            for (int j = 0; j < W; ++j) trv[j] = tr[j * W + (lane & (W - 1))];
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const double yj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(bl), j), __builtin_amdgcn_readlane(__double2loint(bl), j));
                const double t = bl - yj * trv[j];
                bl = (yj != 0.0 && lane > j) ? t : bl;
            }
            if (lane < w) bs[jb + lane] = bl;
        }
        nlh_lds_barrier();
        for (int i = i0; i < n; i += BS) {
            double bi = bs[i];
            if (i == i0) {
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    const double yj = bs[jb + (j < w ? j : 0)];
                    const double t = bi - yj * pf[j];
                    bi = (j < w && yj != 0.0) ? t : bi;
                }
            } else {
                for (int j = 0; j < w; ++j) {
                    const double yj = bs[jb + j];
                    if (yj != 0.0) bi = bi - yj * a[(size_t)(jb + j) * n + i];
                }
            }
            bs[i] = bi;
        }
        land(buf ^ 1);
    }
#ifdef LU_DBG_CLK
    sc[2] = wall_clock64();
#endif
    // U x = y
    for (int k = nblk; k < 2 * nblk; ++k, buf ^= 1) {
        const int jb = (2 * nblk - 1 - k) * W;
        const int w = n - jb < W ? n - jb : W;
        issue(k + 1);
        nlh_lds_barrier();
        const double *tr = tri + buf * W * W;
        if (wid == 0) {
            const int l = lane & (W - 1);
            double bl = (lane < w) ? bs[jb + lane] : 0.0;
            const double dl = tr[l * W + l];
            double trv[W];
#pragma unroll
            for (int j = 0; j < W; ++j) trv[j] = tr[j * W + l];
            unsigned ran = 0u;
#pragma unroll
            for (int j = W - 1; j >= 0; --j) {
                const double raw = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(bl), j), __builtin_amdgcn_readlane(__double2loint(bl), j));
                const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(dl), j), __builtin_amdgcn_readlane(__double2loint(dl), j));
                const bool go = raw != 0.0;                       // (uniform) the sequential loop skips an exactly zero b(j)
                const double xj = raw / d;
                const double t = bl - xj * trv[j];
                bl = (go && lane < j) ? t : ((go && lane == j) ? xj : bl);
                ran |= go ? (1u << j) : 0u;
            }
            if (lane < w) bs[jb + lane] = bl;
            if (lane == 0) flag[0] = (int32_t)ran;
        }
        nlh_lds_barrier();
        const unsigned ran = (unsigned)flag[0];
        for (int i = tid; i < jb; i += BS) {
            double bi = bs[i];
            if (i == tid) {
#pragma unroll
                for (int j = W - 1; j >= 0; --j) {
                    const double xj = bs[jb + (j < w ? j : 0)];
                    const double t = bi - xj * pf[j];
                    bi = ((ran >> j) & 1u) ? t : bi;
                }
            } else {
                for (int j = w - 1; j >= 0; --j)
                    if ((ran >> j) & 1u) bi = bi - bs[jb + j] * a[(size_t)(jb + j) * n + i];
            }
            bs[i] = bi;
        }
        land(buf ^ 1);
    }
#ifdef LU_DBG_CLK
    sc[3] = wall_clock64();
    if (tid == 0 && p == 0) printf("k_lu_solve n %d: interchanges %lld, forward %lld, backward %lld (x10 ns)\n", n, sc[1] - sc[0], sc[2] - sc[1], sc[3] - sc[2]);
#endif
    __syncthreads();
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
            double v = fabs(cj[i]);
            if (v != v) v = (i == j) ? __builtin_inf() : -1.0;    // the ordered search keeps a NaN diagonal entry and never takes a NaN below it
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
template <bool FAST>
static __global__ void __launch_bounds__(256)
k_lu_gemm(int n, double *__restrict__ Aall, int jb, int nb, const LmState *__restrict__ st, int want, int cskip = 0)
{   // cskip > 0 (the look-ahead of lu_blocked): the first cskip trailing columns are the next panel's, updated by its own kernel
    __shared__ double Ls[LU_NB * 64];              // Ls[j*64 + r]
    __shared__ double Us[LU_NB * 64];              // Us[j*64 + c]
    const int p = blockIdx.z;
    if (st && st[p].stage != want) return;                       // lock-step batches: only problems in this stage
    double *a = Aall + (size_t)p * n * n;
    const int t0 = jb + nb;
    const int r0 = t0 + blockIdx.x * 64, c0 = t0 + blockIdx.y * 64;
    const int tid = threadIdx.x;
    // Every load of the tile -- its L and U strips and its own sixteen entries -- is issued before any is used, with
    // clamped indices (round 4: the staging loop used to wait for each pair, four to eight trips to L2 per tile).
    constexpr int NST = LU_NB * 64 / 256;
    double lv[NST], uv[NST];
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        const int e = tid + s * 256, j = e >> 6, q = e & 63, jc = j < nb ? j : nb - 1;
        lv[s] = a[(size_t)(jb + jc) * n + (r0 + q < n ? r0 + q : n - 1)];      // L21(r, j), contiguous in r
        uv[s] = a[(size_t)(c0 + q < n ? c0 + q : n - 1) * n + jb + jc];        // U12(j, c)
    }
    const int tr = (tid & 15) * 4, tc = (tid >> 4) * 4;
    double acc[4][4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = r0 + tr + rr, c = c0 + tc + cc;
            acc[cc][rr] = a[(size_t)(c < n ? c : n - 1) * n + (r < n ? r : n - 1)];
        }
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        const int e = tid + s * 256, q = e & 63;
        asm volatile("" : "+v"(lv[s]), "+v"(uv[s]));
        Ls[e] = (r0 + q < n) ? lv[s] : 0.0;
        Us[e] = (c0 + q < n) ? uv[s] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) asm volatile("" : "+v"(acc[cc][rr]));
    for (int j = 0; j < nb; ++j) {
        double l[4], u[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { l[q] = Ls[j * 64 + tr + q]; u[q] = Us[j * 64 + tc + q]; }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) acc[cc][rr] = FAST ? __builtin_fma(-l[rr], u[cc], acc[cc][rr]) : acc[cc][rr] - l[rr] * u[cc];
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = r0 + tr + rr, c = c0 + tc + cc;
            if (r < n && c < n && c - t0 >= cskip) a[(size_t)c * n + r] = acc[cc][rr];
        }
}

// ---------------------------------------------------------------------------
// Round 4: the panel with SEVERAL ROWS PER THREAD, implicit interchanges and ONE barrier per column.
//
// k_lu_panel_lds runs sixteen waves for 1024 rows, and what a column step costs there is the instruction stream of the
// reductions, executed by four waves on every SIMD, and two barriers.  Here thread t holds RPT rows (row q*T + t of the
// panel in slot q) of an NB-column panel, so a 1024-row panel is four waves -- one per SIMD.
//
// * Interchanges are not performed: every row carries its current POSITION (pos[q]: what LAPACK's sequence of
//   interchanges would have made its row index); rows at positions < c are finished, the pivot search runs over the
//   rows at positions >= c and takes the first maximum in POSITION order (so ipvt is the sequence the unblocked loop
//   finds, ties included), the winner takes position c and the row that was there takes the winner's.  Rows go back
//   to memory at their final positions, and the panel leaves a MOVE LIST (which row ends where) so that the columns
//   outside the panel are permuted with all loads in flight at once instead of nb dependent exchanges.
// * The maximum of |a| is an unsigned maximum of its bit pattern, high word then low word: 32-bit DPP steps.
// * Every wave's winner publishes its whole row and the reciprocal of its entry BEFORE the barrier (the reciprocal is
//   formed by every lane for its own candidate in the shadow of the reduction: the same IEEE division whoever performs
//   it); after the barrier every thread picks the winning wave from the per-wave results and reads that row.
// * Column c + 1 is updated first and its candidates go out while the other columns are updated.
//
// Per element the operations are those of the unblocked loop in the same order: multiplier = a * (1 / pivot), then
// a(i,k) = a(i,k) - l(i) * u(k) for the pivots in ascending order -- with a separate multiply and subtract the factors
// are bit-identical to k_lu_panel_lds / the CPU restatement (FAST = false); FAST = true contracts them into one FMA.
// NaN handling as in k_lu_panel_lds: a NaN on the diagonal is taken, a NaN below it never.
// ---------------------------------------------------------------------------
#define LU_MV_STRIDE 128   // ints per problem: [i < 32] source row of the row that ends at panel position i; [32] number of
                           // displaced rows; [33 + 2e], [34 + 2e] destination and source of displaced row e (panel-relative)

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t lu_dpp_umax_step(uint32_t v)
{
    // bound_ctrl: lanes without a source (and rows outside the mask) read 0 -- neutral for an unsigned maximum
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
    return o > v ? o : v;
}
__device__ __forceinline__ uint32_t lu_wave_umax(uint32_t v)
{
    v = lu_dpp_umax_step<0x111, 0xf>(v);
    v = lu_dpp_umax_step<0x112, 0xf>(v);
    v = lu_dpp_umax_step<0x114, 0xf>(v);
    v = lu_dpp_umax_step<0x118, 0xf>(v);
    v = lu_dpp_umax_step<0x142, 0xa>(v);
    v = lu_dpp_umax_step<0x143, 0xc>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ int lu_wave_min(int c)
{
    c = lu_row16_min(c);
    c = lu_dpp_min_step<0x142, 0xa, 0xf>(c);
    c = lu_dpp_min_step<0x143, 0xc, 0xf>(c);
    return __builtin_amdgcn_readlane(c, 63);
}

// dynamic LDS of k_lu_panel_reg: the finished multipliers, column by column, indexed by the row's slot
static inline size_t lu_panel_reg_lds(int rpt, int nbw, int threads) { return sizeof(double) * (size_t)nbw * rpt * threads; }

template <int RPT, int NB, int TMAX, bool FAST>
static __global__ void __launch_bounds__(TMAX)
k_lu_panel_reg(int n, double *__restrict__ Aall, int32_t *__restrict__ ipvt_all, int32_t *__restrict__ info,
               int32_t *__restrict__ mv_all, int jb, int nb, const LmState *__restrict__ st, int want,
               int pnb = 0, const int32_t *__restrict__ pmv_all = nullptr)
{   // pnb > 0 (the look-ahead of lu_blocked): the PREVIOUS panel [jb - pnb, jb) has not been applied to this panel's columns
    // yet -- this kernel does it first, for exactly its own columns (row moves from pmv_all, the block row by forward
    // substitution, the trailing product), with the operations k_lu_move_trsm / k_lu_gemm would have performed, in their order
    constexpr int NWMAX = TMAX / 64;
    constexpr int WS = NB + 2;                                   // doubles per published row: the window, then 1 / its entry
    __shared__ __attribute__((aligned(16))) uint32_t red_hi[2][NWMAX];
    __shared__ __attribute__((aligned(16))) uint32_t red_lo[2][NWMAX];
    __shared__ __attribute__((aligned(16))) int red_p[2][NWMAX];
    __shared__ __attribute__((aligned(16))) double wrow[2][NWMAX][WS];
    __shared__ __attribute__((aligned(16))) double ubuf[NB][NB]; // ubuf[c][j]: u(c, c + j)
    __shared__ int piv_s[NB], src_s[NB];                         // pivot position and slot row of step c
    __shared__ int zero_s;                                       // first step with a zero pivot + 1
    __shared__ int mvcnt;
    extern __shared__ __attribute__((aligned(16))) double lbuf[];            // lbuf[c * RT + slot row]: multiplier of column c
    const int p = blockIdx.x, t = threadIdx.x, T = blockDim.x, lane = t & 63, wid = t >> 6, RT = RPT * T;
    if (st && st[p].stage != want) return;                       // lock-step batches: only problems in this stage
    const int rows = n - jb;
    double *a = Aall + (size_t)p * n * n + (size_t)jb * n + jb;  // a[k * n + r]: panel column k, panel row r
    int32_t *ipvt = ipvt_all + (size_t)p * n + jb;
    int32_t *mv = mv_all ? mv_all + (size_t)p * LU_MV_STRIDE : nullptr;
    // The window: w[q][j] is column c + j of the row in slot q at step c.  The step loop is NOT unrolled and has (almost)
    // no branches: a branch into code that has never run costs 200 cycles, 40 when warm (profiles/ubench/icache_branch.hip),
    // and the unrolled, branchy form of this kernel took 2 us a column.  Finished rows keep being "updated": their
    // windows are never read again, so no predicate is needed.
    double w[RPT][NB];
    int pos[RPT];
#ifdef LU_DBG_CLK       // -DLU_DBG_CLK: in-kernel phase clocks (100 MHz) of the first panel, printed by thread 0
    long long clk[8]; clk[0] = wall_clock64();
    long long sck[6] = {0, 0, 0, 0, 0, 0};      // core cycles per phase of the step loop, summed over the panel's steps
#endif
    // loads are unconditional (clamped indices) and all issued before any is used: the compiler puts a load under a
    // condition back under it, and a load in a conditional block is waited for before the next one is issued
    if (pnb > 0) {
        // ---- the previous panel's update of THESE columns (uniform branch) ----------------------------------------------
        __shared__ __attribute__((aligned(16))) double u12[32][NB];          // the block row: U(pjb + j, jb + k)
        __shared__ __attribute__((aligned(16))) double l11[32 * 32];         // l11[i + j * pnb], i > j used
        const int pjb = jb - pnb;
        const int32_t *pmv = pmv_all + (size_t)p * LU_MV_STRIDE;
        const int cnt2 = pmv[32];
        double *ab = Aall + (size_t)p * n * n;
        // this thread's rows as the moves leave them: position pnb + r of the previous panel's frame holds what was at `src`
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = q * T + t;
            pos[q] = r < rows ? r : -1;
            const int i = pnb + (r < rows ? r : rows - 1);
            int src = i;
            for (int e = 0; e < cnt2; ++e) src = (pmv[33 + 2 * e] == i) ? pmv[34 + 2 * e] : src;     // (uniform list: scalar loads)
#pragma unroll
            for (int k = 0; k < NB; ++k) w[q][k] = ab[(size_t)(jb + (k < nb ? k : nb - 1)) * n + pjb + src];
        }
        for (int e = t; e < pnb * pnb; e += T) { const int i = e % pnb, j = e / pnb; l11[e] = ab[(size_t)(pjb + j) * n + pjb + i]; }
        double ut[32];
        const int uk = t < nb ? t : nb - 1;
        if (t < NB) {
#pragma unroll
            for (int i = 0; i < 32; ++i) ut[i] = ab[(size_t)(jb + uk) * n + pjb + pmv[i < pnb ? i : 0]];   // the rows that end in the block row
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q)
#pragma unroll
            for (int k = 0; k < NB; ++k) LU_PIN(w[q][k]);
        if (t < NB) {
#pragma unroll
            for (int i = 0; i < 32; ++i) LU_PIN(ut[i]);
        }
        __syncthreads();                                         // every old value is in registers: the block row may be overwritten
        if (t < NB) {                                            // U12 = L11^-1 (moved rows), column t: j ascending, separate multiply / subtract
            for (int j = 0; j < pnb; ++j) {
                const double uj = ut[0];
                if (t < nb) ab[(size_t)(jb + t) * n + pjb + j] = uj;
                u12[j][t] = t < nb ? uj : 0.0;
#pragma unroll
                for (int i = 1; i < 32; ++i) {
                    const double li = (j + i < pnb) ? l11[(j + i) + j * pnb] : 0.0;
                    ut[i - 1] = FAST ? __builtin_fma(-li, uj, ut[i]) : ut[i] - li * uj;
                }
                ut[31] = 0.0;
            }
        }
        __syncthreads();
        // the trailing product: a(i, k) -= l(i, j) u(j, k), j ascending (l: the previous panel's multipliers at the rows' final places)
        for (int j0 = 0; j0 < pnb; j0 += 8) {
            double lq[RPT][8];
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int r = q * T + t, rc = r < rows ? r : rows - 1;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) lq[q][jj] = ab[(size_t)(pjb + (j0 + jj < pnb ? j0 + jj : pnb - 1)) * n + jb + rc];
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                if (j0 + jj < pnb) {                              // (uniform)
#pragma unroll
                    for (int q = 0; q < RPT; ++q)
#pragma unroll
                        for (int k = 0; k < NB; ++k)
                            w[q][k] = FAST ? __builtin_fma(-lq[q][jj], u12[j0 + jj][k], w[q][k]) : w[q][k] - lq[q][jj] * u12[j0 + jj][k];
                }
            }
        }
    } else {
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = q * T + t;
        pos[q] = r < rows ? r : -1;
        const int rc = r < rows ? r : rows - 1;
#pragma unroll
        for (int k = 0; k < NB; ++k) w[q][k] = a[(size_t)(k < nb ? k : nb - 1) * n + rc];
    }
    }
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = q * T + t;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            LU_PIN(w[q][k]);
            w[q][k] = (r < rows && k < nb) ? w[q][k] : 0.0;
        }
    }
    if (t < 2 * NWMAX) {                                         // waves that do not exist never win
        (&red_hi[0][0])[t] = 0u; (&red_lo[0][0])[t] = 0u; (&red_p[0][0])[t] = 0x7fffffff;
    }
    if (t == 0) { mvcnt = 0; zero_s = 0; }
    __syncthreads();
    // this lane's candidate of the column being searched: key = bits of |a|, position, reciprocal of the signed entry
    int lp = 0x7fffffff;
    bool winner = false;
    double wlrc = 1.0;
    auto search = [&](int cc) {
        const int par = cc & 1;
        uint32_t lhi = 0, llo = 0;
        lp = 0x7fffffff;
        double lx = 1.0;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const double x = w[q][0];
            const bool nan = x != x, alive = pos[q] >= cc, dia = pos[q] == cc;
            uint32_t hi = (uint32_t)__double2hiint(x) & 0x7fffffffu, lo = (uint32_t)__double2loint(x);
            hi = nan ? 0x7ff00000u : hi; lo = nan ? 0u : lo;     // a NaN on the diagonal is taken, as the ordered search does
            const bool valid = alive & (!nan | dia);
            const bool take = valid & ((hi > lhi) | ((hi == lhi) & ((lo > llo) | ((lo == llo) & (pos[q] < lp)))));
            lhi = take ? hi : lhi; llo = take ? lo : llo; lp = take ? pos[q] : lp; lx = take ? x : lx;
        }
        wlrc = 1.0 / lx;
        const bool have = lp != 0x7fffffff;
        const uint32_t mhi = lu_wave_umax(have ? lhi : 0u);
        const uint32_t mlo = lu_wave_umax((have & (lhi == mhi)) ? llo : 0u);
        const int cp = (have & (lhi == mhi) & (llo == mlo)) ? lp : 0x7fffffff;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(cp != 0x7fffffff);
        int wp;
        if (__builtin_expect(__builtin_popcountll(m) == 1, 1)) wp = __builtin_amdgcn_readlane(cp, __builtin_ctzll(m));
        else wp = lu_wave_min(cp);
        winner = (cp == wp) & (cp != 0x7fffffff);
        if (lane == 0) { red_hi[par][wid] = mhi; red_lo[par][wid] = mlo; red_p[par][wid] = wp; }
    };
    // ... and, once the step's updates are complete, this wave's winner publishes its window and reciprocal
    auto publish = [&](int cc) {
        double *wr = &wrow[cc & 1][wid][0];
        if (winner) {
#pragma unroll
            for (int q = 0; q < RPT; ++q)
                if (pos[q] == lp) {
#pragma unroll
                    for (int k = 0; k < NB; ++k) wr[k] = w[q][k];
                }
            wr[NB] = wlrc;
        }
    };
#ifdef LU_DBG_CLK
    clk[1] = wall_clock64();
#endif
    search(0);
    publish(0);
#ifdef LU_DBG_CLK
    clk[2] = wall_clock64();
#endif
#pragma unroll 1
    for (int c = 0; c < nb; ++c) {
        const int par = c & 1;
#ifdef LU_DBG_CLK
        if (c == 4) clk[3] = wall_clock64();
        if (c == 12) clk[4] = wall_clock64();
        long long sc0 = clock64(), sc1;
#define LUSC(i) { sc1 = clock64(); sck[i] += sc1 - sc0; sc0 = sc1; }
#else
#define LUSC(i)
#endif
        nlh_lds_barrier();                         // LDS only: nothing in the loop goes to global memory
        LUSC(0)
        // the winning wave: largest key, smallest position among equals
        uint32_t bhi = red_hi[par][0], blo = red_lo[par][0];
        int ppos = red_p[par][0], ws = 0;
#pragma unroll
        for (int v = 1; v < NWMAX; ++v) {
            const uint32_t oh = red_hi[par][v], ol = red_lo[par][v];
            const int op = red_p[par][v];
            const bool take = (oh > bhi) | ((oh == bhi) & ((ol > blo) | ((ol == blo) & (op < ppos))));
            bhi = take ? oh : bhi; blo = take ? ol : blo; ppos = take ? op : ppos; ws = take ? v : ws;
        }
        const double *pr = &wrow[par][ws][0];
        double u[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) u[j] = pr[j];
        const double rcp = pr[NB];
        LUSC(1)
        const bool nz = u[0] != 0.0;
        int myslot = 0;
#pragma unroll
        for (int q = 0; q < RPT; ++q) myslot = pos[q] == ppos ? q * T + t : myslot;
        if (lp == ppos) { piv_s[c] = ppos; src_s[c] = myslot; }  // the owner
        if (t < NB) ubuf[c][t] = pr[t];
        if (!nz && t == 0 && zero_s == 0) zero_s = c + 1;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const bool own = pos[q] == ppos, dia = pos[q] == c;
            pos[q] = own ? c : (dia ? ppos : pos[q]);
        }
        // multipliers, then column c + 1 first: its candidates go out while the other columns are updated
        double l[RPT];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            l[q] = nz ? w[q][0] * rcp : w[q][0];   // zero pivot: no scaling; the update still runs
            lbuf[(size_t)c * RT + q * T + t] = l[q];
            w[q][0] = FAST ? __builtin_fma(-l[q], u[1], w[q][1]) : w[q][1] - l[q] * u[1];
        }
        LUSC(2)
        if (c + 1 < nb) search(c + 1);
        LUSC(3)
#pragma unroll
        for (int j = 2; j < NB; ++j) {
#pragma unroll
            for (int q = 0; q < RPT; ++q) w[q][j - 1] = FAST ? __builtin_fma(-l[q], u[j], w[q][j]) : w[q][j] - l[q] * u[j];
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) w[q][NB - 1] = 0.0;
        LUSC(4)
        if (c + 1 < nb) publish(c + 1);
        LUSC(5)
    }
#ifdef LU_DBG_CLK
    clk[5] = wall_clock64();
#endif
    __syncthreads();
    // rows to their final places: multipliers left of the row's pivot step, its U part from there on
#pragma unroll
    for (int q = 0; q < RPT; ++q)
        if (pos[q] >= 0) {
            const int fin = pos[q], cp = fin < nb ? fin : nb, slot = q * T + t;
            const int cpc = cp < NB ? cp : NB - 1;
            double v[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {                       // (both LDS reads unconditional: all in flight together)
                const double lv = lbuf[(size_t)k * RT + slot], uv = ubuf[cpc][k >= cp ? k - cp : 0];
                v[k] = k < cp ? lv : uv;
            }
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (k < nb) a[(size_t)k * n + fin] = v[k];
            if (mv && fin >= nb && fin != slot) {                // displaced below the panel's block row
                const int e = atomicAdd(&mvcnt, 1);
                mv[33 + 2 * e] = fin;
                mv[34 + 2 * e] = slot;
            }
        }
    if (t < nb) {
        ipvt[t] = jb + piv_s[t];
        if (mv) mv[t] = src_s[t];
    }
    if (t == 0 && zero_s != 0 && info && info[p] == 0) info[p] = jb + zero_s;
    if (mv) {
        __syncthreads();
        if (t == 0) mv[32] = mvcnt;
    }
#ifdef LU_DBG_CLK
    clk[6] = wall_clock64();
    if (t == 0 && p == 0 && (jb == 0 || jb == 512 || jb == 768))
        printf("k_lu_panel_reg<%d,%d> jb %d rows %d: load %lld, first search %lld, steps 0-3 %lld (x10 ns), steps 4-11 %lld, rest %lld, output %lld\n", RPT, NB, jb,
               rows, (clk[1] - clk[0]), (clk[2] - clk[1]), (clk[3] - clk[2]), (clk[4] - clk[3]), (clk[5] - clk[4]), (clk[6] - clk[5]));
    if (t == 0 && p == 0 && (jb == 0 || jb == 512 || jb == 768))
        printf("   per step, core cycles (thread 0, %d steps): barrier %lld, winner + u row %lld, multipliers + column c+1 %lld, search %lld, other columns %lld, publish %lld\n",
               nb, sck[0] / nb, sck[1] / nb, sck[2] / nb, sck[3] / nb, sck[4] / nb, sck[5] / nb);
#endif
}

// The columns outside panel [jb, jb + nb) after k_lu_panel_reg: the panel's row moves (all sources loaded before any
// destination is stored: one trip to memory each way instead of nb dependent exchanges), then -- right of the panel --
// the column's part of the block row (unit lower triangular solve, j ascending; separate multiply and subtract unless
// FAST).  A thread per column.
template <int TR, bool FAST>
static __global__ void __launch_bounds__(256)
k_lu_move_trsm(int n, double *__restrict__ Aall, const int32_t *__restrict__ mv_all, int jb, int nb,
               const LmState *__restrict__ st, int want, int skip_lo = 0, int skip_hi = 0)
{   // skip_lo .. skip_hi (outside-column indices; the look-ahead of lu_blocked): columns the next panel's kernel updates itself
    __shared__ double L11[TR * TR + TR];           // L11[i + j*TR], i > j used (+ TR: the shifting window reads past the end)
    const int p = blockIdx.y;
    if (st && st[p].stage != want) return;                       // lock-step batches: only problems in this stage
    double *a = Aall + (size_t)p * n * n;
    const int32_t *mv = mv_all + (size_t)p * LU_MV_STRIDE;       // (uniform addresses: scalar loads, no LDS copy, no barrier before use)
    // this thread's elements of L11, its column's sources: all in flight together
    constexpr int NL = (TR * TR + 255) / 256;
    double lv[NL];
#pragma unroll
    for (int s = 0; s < NL; ++s) {
        const int e = threadIdx.x + s * 256, i = e % TR, j = (e / TR) < TR ? e / TR : TR - 1;
        lv[s] = a[(size_t)(jb + (j < nb ? j : nb - 1)) * n + jb + (i < nb ? i : nb - 1)];
    }
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = k < n - nb && !(k >= skip_lo && k < skip_hi);
    if (!live) k = 0;
    if (k >= jb) k += nb;                          // skip the panel's own columns
    if (k >= n) k = n - 1;
    double *ck = a + (size_t)k * n + jb;
    const int cnt2 = mv[32];
    // every source before any destination; the loads are unconditional (index 0 for unused entries): a load under a
    // condition is waited for before the next one is issued
    double u[TR], d[TR];
#pragma unroll
    for (int i = 0; i < TR; ++i) u[i] = ck[i < nb ? mv[i] : 0];
#pragma unroll
    for (int e = 0; e < TR; ++e) d[e] = ck[e < cnt2 ? mv[34 + 2 * e] : 0];
#pragma unroll
    for (int s = 0; s < NL; ++s) {
        const int e = threadIdx.x + s * 256, i = e % TR, j = e / TR;
        LU_PIN(lv[s]);
        if (e < TR * TR) L11[e] = (i < nb && j < nb) ? lv[s] : 0.0;
    }
    if (threadIdx.x < TR) L11[TR * TR + threadIdx.x] = 0.0;
    __syncthreads();
    if (!live) return;
#pragma unroll
    for (int i = 0; i < TR; ++i) { LU_PIN(u[i]); LU_PIN(d[i]); }
#pragma unroll
    for (int e = 0; e < TR; ++e)
        if (e < cnt2) ck[mv[33 + 2 * e]] = d[e];
    if (k < jb + nb) {                             // left of the panel: the moves only
#pragma unroll
        for (int i = 0; i < TR; ++i)
            if (i < nb && mv[i] != i) ck[i] = u[i];
        return;
    }
    // right of the panel: the block row.  u(j) is final when step j begins; the window u[0 ..] holds rows j, j + 1, ...
    // (a rolled loop: TR - 1 multiply / subtract pairs a step, the code stays in the instruction cache)
#pragma unroll 1
    for (int j = 0; j < nb; ++j) {
        const double uj = u[0];
        ck[j] = uj;
        const double *lj = L11 + j * TR + j;       // lj[i]: l(j + i, j); beyond the panel it reads zeros or the next column (rows >= nb: unused)
#pragma unroll
        for (int i = 1; i < TR; ++i) {
            const double li = lj[i];
            u[i - 1] = FAST ? __builtin_fma(-li, uj, u[i]) : u[i] - li * uj;
        }
        u[TR - 1] = 0.0;
    }
}
