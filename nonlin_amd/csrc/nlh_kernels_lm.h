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

// -DNLH_DEBUG_LMPAR_CLK: in-kernel clocks (100 MHz) of lmpar's phases, printed by workgroup 0 when its iteration ran
#ifdef NLH_DEBUG_LMPAR_CLK
__device__ unsigned long long g_lmclk[16];
#define LMCLK(i) { __syncthreads(); const unsigned long long t_ = wall_clock64(); if (threadIdx.x == 0) g_lmclk[i] += t_ - lmclk_t; lmclk_t = t_; }
#define LMCLK_START() unsigned long long lmclk_t = wall_clock64();
#else
#define LMCLK(i)
#define LMCLK_START()
#endif

// ---------------------------------------------------------------------------------------------------------------
// The Givens sweeps of lmsolve (:717-765) ON CHIP, n <= LMS_MAX_N (round 6).  Same wavefront as the global-memory form
// below -- rotation (j,k) at time step t = j + k -- and every datum sees the reference's sequence of operations
// (bit-identical), but nothing of a time step goes through L2:
//   * the working row W_j of elimination j lives in the REGISTERS of the wave that owns it (wave j mod NW, slot
//     (j / NW) mod 8; element i in lane i & 63, register (i >> 6) - v0 with v0 = (k+1) >> 6 the first 64-element chunk
//     that still holds a live entry -- the registers move down by one whenever k+1 crosses a multiple of 64, so the
//     entry the next rotation eliminates is always in register 0);
//   * the columns of S that are live (column k from its first rotation at t = k to its last at t = 2k) sit in an LDS
//     ring: columns enter and leave in the order of k, so a FIFO; column k occupies the elements i >= 64 v0(k), whole
//     chunks, so that a chunk of 64 lanes needs NO lane mask: the lanes below the diagonal (i <= k) and beyond n read and
//     write the column's own padding and carry garbage that nothing reads.  The live band peaks at ~n^2/6 doubles plus
//     padding (115 KB at n = 256: lmsolve_ring_bytes()); column t + LMS_AHEAD is fetched from global memory during step
//     t (a load per lane of waves 0..NV-1, stored into the ring a step later), a column's final values go to the lower
//     triangle of r in global memory straight from the registers of its last rotation (t = 2k);
//   * the rotations of a time step (:733-748) are formed once, by the first one or two waves (a lane per rotation), and
//     published through LDS; sdiag / wa / qtbp / rot as in the global form; two LDS-only barriers per time step.
// What bounds it: a wave issues at most one instruction per four cycles, so a time step costs (instructions a wave executes)
// x 4 cycles -- the code of a step is kept to the rotation's arithmetic plus a few scalar instructions per slot
// (measured: docs/lab_notebook.md).  Before: 6.6 us per time step (two dependent trips to L2 and a full barrier),
// 3.35 - 3.9 ms per sweep at n = 256; now 1.16 ms (2.3 us per step).
// ---------------------------------------------------------------------------------------------------------------
#define LMS_MAX_N 256
#define LMS_AHEAD 8         // columns fetched ahead of their first rotation ...
#define LMS_LDGRP 4         // ... in groups of four (< LMS_AHEAD): a group's loads are issued together and stored four time steps later --
                            // a load stored one step after its issue made every time step wait a memory latency (1.5 us)
#define LMS_SLOTS 8
#define LMS_NV (LMS_MAX_N / 64)
// doubles column k occupies in the ring: the elements i = 64 v0(k) .. 64 ceil(n / 64) - 1
__host__ __device__ inline int lms_colsize(int k, int n) { return ((n + 63) & ~63) - (((k + 1) >> 6) << 6); }
// Column k's place in the ring: behind column k-1, or at the ring's start when it would not fit before the end (a column
// never straddles the end, so an element's address needs no wrap-around).  The same rule on the host and on the device.
#define LMS_PLACE(pos, len, cap) (((pos) + (len) > (cap)) ? 0 : (pos))
// Does a ring of `cap` doubles hold every column from the step it is stored (the step after its load group's, 0 for the
// first AHEAD columns) to its last rotation (step 2c) without touching a live one?
static inline bool lms_ring_fits(int n, int cap)
{
    std::vector<int> base(n, 0);
    int pos = 0;
    // (column n-1 has no element below the diagonal, but its rotations run over its chunk like any other: it gets its padding)
    for (int k = 0; k < n; ++k) { const int len = lms_colsize(k, n); pos = LMS_PLACE(pos, len, cap); base[k] = pos; pos += len; }
    if (pos > cap || lms_colsize(0, n) > cap) return false;
    for (int c = 0; c < n; ++c) {
        const int tw = c >= LMS_AHEAD ? ((c - LMS_AHEAD) / LMS_LDGRP + 1) * LMS_LDGRP : 0, kmin = c >= LMS_AHEAD ? tw / 2 : 0;
        for (int k = kmin; k < c; ++k)
            if (base[k] < base[c] + lms_colsize(c, n) && base[c] < base[k] + lms_colsize(k, n)) return false;
    }
    return true;
}
// LDS the ring needs for an n-by-n sweep by `threads` threads: *cap = doubles of the ring proper (0: the on-chip form does
// not apply -- n too large, or too few waves for the n/2 rows that are live at a time); returns bytes incl. the offset table.
static inline size_t lmsolve_ring_bytes(int n, int threads, int *cap)
{
    *cap = 0;
    if (n > LMS_MAX_N || n < 2 || (n + 1) / 2 > (threads / 64) * LMS_SLOTS) return 0;
    static int cached[LMS_MAX_N + 1];                           // (benign race: every writer stores the same value)
    int c = cached[n];
    if (!c) {
        c = 64;
        while (!lms_ring_fits(n, c)) c += 64;
        cached[n] = c;
    }
    *cap = c;
    // + the table of column offsets (n int32) + the time step's rotations (cs, sn: 2 n doubles; ring offset and flag: n int32)
    return sizeof(double) * ((size_t)c + (size_t)(n + 1) / 2 + 8 + 2 * (size_t)n + (size_t)(n + 1) / 2 + 8);
}
// doubles of LDS behind lmpar's n-vectors for the exact reductions (norm2_flang_block: 3 NLH_NCH + 8; the lanes form of the
// m-entry norm of deviation A: its chunk + flags, 64 EL + 128 + 40 + threads / 2 with EL = 32 for sixteen waves, else 16)
__host__ __device__ inline int lmpar_scratch_doubles(int threads) { return threads >= 1024 ? 64 * 32 + 128 + 40 + 512 + 8 : 3 * NLH_NCH + 8; }
// threads of a k_lmpar workgroup: the on-chip sweep needs a wave per eight live rows (n / 2 rows are live at a time) -- four
// waves to n = 64, eight to n = 128 (two workgroups then share a CU: families whose trust region binds on most iterations
// run lmpar's iteration for hundreds of problems at once), sixteen beyond
static inline int lmpar_threads(int n) { return n > 128 ? 1024 : n > 64 ? 512 : 256; }

template <int NV>
__device__ __forceinline__ void lmsolve_sweep_lds(int n, double *r, int ldr, double *sdiag, double *wa, double *qtbp,
                                                  double *rot, double *ring, int cap)
{
    const int tid = threadIdx.x, BS = blockDim.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), nw = BS >> 6, P = nw * LMS_SLOTS;
    const int nvt = (n + 63) >> 6;                              // chunks of 64 elements in all (<= NV)
    int32_t *cbt = reinterpret_cast<int32_t *>(ring + cap);    // cbt[k]: where column k (its element 64 v0(k)) starts in the ring
    double2 *parcs = reinterpret_cast<double2 *>(ring + cap + (n + 1) / 2 + 8 - ((n + 1) / 2 & 1));   // (cs, sn) of row j's rotation of this time step
    int32_t *parcb = reinterpret_cast<int32_t *>(reinterpret_cast<double *>(parcs) + 2 * (size_t)n);     // its column's ring offset | applied << 30
    if (tid == 0) {
        int pos = 0;
        for (int k = 0; k < n; ++k) { const int len = lms_colsize(k, n); pos = LMS_PLACE(pos, len, cap); cbt[k] = pos; pos += len; }
    }
    __syncthreads();
    // columns 0 .. AHEAD-1 straight into the ring
    {
        const int kend = LMS_AHEAD < n ? LMS_AHEAD : n;
        for (int k = 0; k < kend; ++k) {
            const int base = cbt[k] - (((k + 1) >> 6) << 6);
            for (int i = k + 1 + tid; i < n; i += BS) ring[base + i] = r[(size_t)k * ldr + i];
        }
    }
    double w[LMS_SLOTS][NV];
#pragma unroll
    for (int u = 0; u < LMS_SLOTS; ++u)
#pragma unroll
        for (int v = 0; v < NV; ++v) w[u][v] = 0.0;
    int myj = wid + nw * (lane & (LMS_SLOTS - 1));             // lane u < 8: the row in slot u of this wave
    double ldv[LMS_LDGRP];                                      // chunk wid of the columns ldc .. ldc + LDGRP - 1, in flight
#pragma unroll
    for (int d = 0; d < LMS_LDGRP; ++d) ldv[d] = 0.0;
    int ldc = -1;
    __syncthreads();
#ifdef NLH_DEBUG_SWEEP_CLK      // (per time step: four s_memtime reads and their waits -- slows the sweep it measures by a third)
    long long ck_ld = 0, ck_form = 0, ck_app = 0, ck_bar = 0, ck0 = clock64(), ck1;
#define SWCLK(acc) { ck1 = clock64(); acc += ck1 - ck0; ck0 = ck1; }
#else
#define SWCLK(acc)
#endif
    for (int t = 0; t <= 2 * (n - 1); ++t) {
        // ring traffic of the look-ahead, every LDGRP-th step: store the group fetched LDGRP steps ago, fetch the next one
        if (wid < NV && (t % LMS_LDGRP) == 0) {
            if (ldc >= 0) {
#pragma unroll
                for (int d = 0; d < LMS_LDGRP; ++d) {
                    const int c = ldc + d, i = (((c + 1) >> 6) + wid) * 64 + lane;
                    if (c < n - 1 && i > c && i < n) ring[cbt[c] + 64 * wid + lane] = ldv[d];
                }
            }
            ldc = t + LMS_AHEAD < n - 1 ? t + LMS_AHEAD : -1;
            if (ldc >= 0) {
#pragma unroll
                for (int d = 0; d < LMS_LDGRP; ++d) {
                    const int c = ldc + d, cc = c < n - 1 ? c : n - 2, i = (((cc + 1) >> 6) + wid) * 64 + lane;
                    ldv[d] = (i > cc && i < n) ? r[(size_t)cc * ldr + i] : 0.0;
                }
            }
        }
        SWCLK(ck_ld)
        // The rotations of the time step (:733-748) are formed ONCE, by the first one or two waves -- lane q of them takes row
        // jlo + q --, and published through LDS; a barrier later every wave applies the rotations of its own slots.  (Until
        // this form every wave formed its own slots' rotations with 8 of its 64 lanes: 130 instructions per wave and step for
        // 8 lanes' worth of work, a third of what a step executed.)
        if (myj + n - 1 < t) myj += P;                          // the slot's row is finished: its next row
        const bool mine = lane < LMS_SLOTS && myj < n && 2 * myj <= t && t <= myj + n - 1;
        {
            const int jlo = t - (n - 1) > 0 ? t - (n - 1) : 0, jhi = t >> 1;
            const int j = jlo + 64 * wid + lane;
            if (wid < (NV > 2 ? 2 : 1) && j <= jhi) {
                const int k = t - j;
                // everything the rotation needs is read before anything is tested (one LDS latency, not three)
                const double sk = rot[j], rkk = sdiag[k], wk = wa[k], qj = qtbp[j];
                int cbf = cbt[k];                                // ring offset of the column; bit 30: the rotation is applied (:732)
                double fcs = 1.0, fsn = 0.0;
                if (sk != 0.0) {                                   // :732 (diag(l) == 0, :721, is the sk == 0 case: see the global form)
                    // :733-741, both branches through one divide / square root / divide (the lanes of a wave take different
                    // branches, and a wave pays for every instruction of both): the same operations on the same operands
                    const bool tan_form = !(fabs(rkk) < fabs(sk));  // cs = 0.5 / sqrt(...), sn = cs * (sk / rkk)
                    const double q = (tan_form ? sk : rkk) / (tan_form ? rkk : sk);
                    const double pr = 0.5 / sqrt(0.25 + 0.25 * (q * q));
                    const double ot = pr * q;
                    fcs = tan_form ? pr : ot;
                    fsn = tan_form ? ot : pr;
                    sdiag[k] = fcs * rkk + fsn * sk;               // :745
                    wa[k] = fcs * wk + fsn * qj;                   // :746-748
                    qtbp[j] = -fsn * wk + fcs * qj;
                    cbf |= 1 << 30;
                }
                parcs[j] = make_double2(fcs, fsn);
                parcb[j] = cbf;
            }
        }
        const int actmask = (int)__builtin_amdgcn_readfirstlane((unsigned)(__ballot(mine) & 0xff));
        nlh_lds_barrier();
        SWCLK(ck_form)
#pragma unroll
        for (int u = 0; u < LMS_SLOTS; ++u) {
            if (!((actmask >> u) & 1)) continue;               // uniform
            const int j = __builtin_amdgcn_readlane(myj, u), k = t - j;
            const double2 csn = parcs[j];                       // (one address for the whole wave: a broadcast read)
            const int cbf = __builtin_amdgcn_readfirstlane(parcb[j]), ok = cbf >> 30;
            const double cs = csn.x, sn = csn.y;
            const int hl = (k + 1) & 63, v0 = (k + 1) >> 6, nlive = nvt - v0;   // live chunks: the column's elements 64 (v0 + c) + lane
            double *col = ring + (cbf & 0x3fffffff) + lane;    // chunk c at col[64 c]
            if (j == k) {
                // The row's first rotation (sdiag(j+1:n) = zero, :722) is column k's last: the column's final values go to
                // the lower triangle of r in memory instead of the ring.  Once per time step, one wave.
                double *gcol = r + (size_t)k * ldr + 64 * v0 + lane;
#pragma unroll
                for (int c = 0; c < NV; ++c) {
                    const int i = 64 * (v0 + c) + lane;
                    w[u][c] = 0.0;
                    if (c < nlive && i > k && i < n) {
                        const double rcur = col[64 * c], scur = 0.0;
                        w[u][c] = ok ? -sn * rcur + cs * scur : 0.0;
                        gcol[64 * c] = ok ? cs * rcur + sn * scur : rcur;
                    }
                }
            } else {
                if (hl == 0) {                                 // k+1 crossed a multiple of 64: chunk 0 died, the registers move down
                    asm volatile("" ::: "memory");             // (keeps this a branch taken once in 64 steps: as selects it was six instructions per slot and step)
#pragma unroll
                    for (int c = 0; c + 1 < NV; ++c) w[u][c] = w[u][c + 1];
                }
                if (ok) {                                      // (not ok: rotation skipped, :732 cycle -- row and column untouched)
#pragma unroll
                    for (int c = 0; c < NV; ++c) {
                        if (c < nlive) {                       // uniform; no lane mask (see above)
                            const double rcur = col[64 * c], scur = w[u][c];
                            w[u][c] = -sn * rcur + cs * scur;  // :753-757
                            col[64 * c] = cs * rcur + sn * scur;
                        }
                    }
                }
            }
            if (lane == hl && k + 1 < n) rot[j] = w[u][0];     // the entry the row's next rotation eliminates
        }
        SWCLK(ck_app)
        nlh_lds_barrier();
        SWCLK(ck_bar)
    }
#ifdef NLH_DEBUG_SWEEP_CLK
    if (blockIdx.x == 0 && lane == 0 && (wid == 0 || wid == 5 || wid == 15))
        printf("[sweep clk64 wave %d, kcycles] loader %lld form %lld apply %lld barrier %lld\n", wid, ck_ld / 1000, ck_form / 1000, ck_app / 1000, ck_bar / 1000);
#endif
}

// Faithful lmsolve on the n-by-n R (global, ld = ldr; strict lower triangle is scratch).
// diagv[l] is the diagonal of sqrt(par) D; x (indexed by original column), sdiag, wa in LDS.
//
// The Givens sweeps (:717-765) are run as a wavefront: rotation (j,k) -- elimination of row j
// of D against column k -- depends only on (j,k-1) and (j-1,k), so all rotations with j + k = t
// are independent and touch disjoint data (column k of r, working row j).  Time step t runs
// them concurrently, one wave per rotation, one barrier per step: 2n-1 steps instead of
// n(n+1)/2 serial rotations.  Every datum still sees exactly the reference's sequence of
// operations, so the result is bit-identical to the serial sweep.
// Wrows: n*n doubles of global scratch (working row of elimination j at Wrows + j*n);
// qtbp: n doubles (LDS), the running qtbpj of each elimination.
template <bool EXACT>
__device__ __forceinline__ void lmsolve_dev(int n, double *r, int ldr, const int32_t *ipvt, const double *diagv,
                            const double *qtb, double *x, double *sdiag, double *wa, double *red,
                            double *Wrows, double *qtbp, double *rot /* LDS, n + 8 doubles */,
                            double *sbuf /* LDS, n doubles, EXACT only */,
                            double *ring = nullptr /* LDS, lmsolve_ring_doubles(): the on-chip sweep */, int ringcap = 0)
{
    const int tid = threadIdx.x, BS = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = BS >> 6;
    LMCLK_START()
    // :710-714, the upper triangle copied into the lower one: eight elements per thread are loaded before any is stored
    // (sources and destinations never overlap, but the compiler cannot know; element by element, every store would
    // wait for its own load: n memory latencies in the column-by-column form)
    for (int e0 = tid; e0 < n * n; e0 += 8 * BS) {
        double t8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * BS, j = e / n, i = e - j * n;
            t8[u] = (e < n * n && i > j) ? r[(size_t)i * ldr + j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * BS, j = e / n, i = e - j * n;
            if (e < n * n && i > j) r[(size_t)j * ldr + i] = t8[u];
        }
    }
    for (int j = tid; j < n; j += BS) { x[j] = r[(size_t)j * ldr + j]; wa[j] = qtb[j]; qtbp[j] = 0.0; }
    if (!ring) {
        for (int e = tid; e < n * n; e += BS) Wrows[e] = 0.0;       // sdiag(j:n) = zero, per elimination (:722)
        __syncthreads();
        for (int j = tid; j < n; j += BS) Wrows[(size_t)j * n + j] = diagv[ipvt[j]];   // sdiag(j) = diag(l) (:723)
    }
    __syncthreads();

    // State the sweep hands from one time step to the next lives in LDS, so that forming a rotation needs no
    // global access: sdiag[k] carries the current diagonal of column k (its value after (k,k) is the sdiag of
    // :763, and the diagonal of r in memory is never touched, which is the restore of :764), rot[j] the entry
    // W_j[k] of working row j that the next rotation of elimination j eliminates.
    for (int j = tid; j < n; j += BS) { sdiag[j] = x[j]; rot[j] = diagv[ipvt[j]]; }
    __syncthreads();
    LMCLK(0)
    if (ring) {                                                 // :717-765 on chip (n <= LMS_MAX_N)
        if (n <= 64) lmsolve_sweep_lds<1>(n, r, ldr, sdiag, wa, qtbp, rot, ring, ringcap);
        else if (n <= 128) lmsolve_sweep_lds<2>(n, r, ldr, sdiag, wa, qtbp, rot, ring, ringcap);
        else lmsolve_sweep_lds<LMS_NV>(n, r, ldr, sdiag, wa, qtbp, rot, ring, ringcap);
    }
    else
    for (int t = 0; t <= 2 * (n - 1); ++t) {                   // :717-765 as a wavefront
        const int jlo = t - (n - 1) > 0 ? t - (n - 1) : 0, jhi = t >> 1;
        const int nrot = jhi - jlo + 1;
        // A wave takes RG rotations at a time: their first loads are issued, lane u forms rotation u
        // (:733-748) while they are in flight, and all lanes then apply the rotations to the rest of
        // column k and working row j (:753-757).  One barrier per time step.
        constexpr int RG = 4;
        for (int q0 = wid * RG; q0 < nrot; q0 += nw * RG) {
            int kk[RG];
            double *cp[RG], *wp[RG];
            double rv[RG], sv[RG];
#pragma unroll
            for (int u = 0; u < RG; ++u) {
                const int q = q0 + u < nrot ? q0 + u : q0;
                const int j = jlo + q, k = t - j;
                kk[u] = k;
                cp[u] = r + (size_t)k * ldr;
                wp[u] = Wrows + (size_t)j * n;
                const int i = k + 1 + lane;
                const bool in = (q0 + u < nrot) && i < n;
                rv[u] = in ? cp[u][i] : 0.0;
                sv[u] = in ? wp[u][i] : 0.0;
            }
            double mycs = 1.0, mysn = 0.0;
            int myok = 0;
            if (lane < RG && q0 + lane < nrot) {
                const int j = jlo + q0 + lane, k = t - j;
                // :721 (diag(l) == 0: the elimination is skipped) needs no test of its own: rot[j] starts as diag(l) and the
                // working row of a skipped elimination stays zero, so sk == 0 covers it -- and the two dependent global loads
                // diagv[ipvt[j]] cost every time step two trips to L2 on its critical path (round 4)
                const double sk = rot[j];
                if (sk != 0.0) {                                   // :732
                    const double rkk = sdiag[k];
                    if (fabs(rkk) < fabs(sk)) {                    // :733-741
                        const double ctan = rkk / sk;
                        mysn = 0.5 / sqrt(0.25 + 0.25 * (ctan * ctan));
                        mycs = mysn * ctan;
                    } else {
                        const double tn = sk / rkk;
                        mycs = 0.5 / sqrt(0.25 + 0.25 * (tn * tn));
                        mysn = mycs * tn;
                    }
                    const double wk = wa[k], qj = qtbp[j];
                    sdiag[k] = mycs * rkk + mysn * sk;             // :745
                    wa[k] = mycs * wk + mysn * qj;                 // :746-748
                    qtbp[j] = -mysn * wk + mycs * qj;
                    myok = 1;
                }
            }
#pragma unroll
            for (int u = 0; u < RG; ++u) {
                const double cs = __shfl(mycs, u, 64), sn = __shfl(mysn, u, 64);
                const int ok = __shfl(myok, u, 64);
                if (q0 + u >= nrot) continue;                      // uniform
                const int k = kk[u], j = t - k;
                if (!ok) {                                         // rotation skipped (:732 cycle): row and column untouched
                    if (lane == 0 && k + 1 < n) rot[j] = sv[u];
                    continue;
                }
                double rcur = rv[u], scur = sv[u];
                for (int i = k + 1 + lane; i < n; i += 64) {
                    const bool more = i + 64 < n;
                    const double rnx = more ? cp[u][i + 64] : 0.0, snx = more ? wp[u][i + 64] : 0.0;
                    const double wnew = -sn * rcur + cs * scur;
                    cp[u][i] = cs * rcur + sn * scur;
                    wp[u][i] = wnew;
                    if (i == k + 1) rot[j] = wnew;
                    rcur = rnx; scur = snx;
                }
            }
        }
        __syncthreads();
    }

    __syncthreads();      // (the on-chip sweep's columns went to memory by plain stores)
    LMCLK(1)
    // singular tail (:769-773) and back-substitution on S^T stored in the lower triangle (:774-784)
    int ns = n;
    for (int j = tid; j < n; j += BS)
        if (sdiag[j] == 0.0) ns = min(ns, j);
    ns = -(int)block_reduce_max((double)(-ns), red);
    __syncthreads();
    for (int j = ns + tid; j < n; j += BS) wa[j] = 0.0;
    __syncthreads();
    if (EXACT && ring) {
        // n <= LMS_MAX_N: the whole back-substitution by ONE wave, no barriers -- lane l holds the products E l .. E l + E - 1 of a
        // column and the ordered sum runs down the lanes (nlh_common.h: 2.2 ns per term against 6 - 10 for a thread that reads
        // its terms from LDS); the next column's elements are fetched from memory while the chain of this one runs
        if (wid == 0) {
            constexpr int E = LMS_MAX_N / 64;
            double nx[E];
#pragma unroll
            for (int u = 0; u < E; ++u) nx[u] = 0.0;             // column ns-1 has no term
            for (int k = 1; k <= ns; ++k) {
                const int j = ns - k, len = k - 1;                // terms i = j+1 .. ns-1
                double d[E];
#pragma unroll
                for (int u = 0; u < E; ++u) {
                    const int q = E * lane + u;
                    d[u] = q < len ? nx[u] * wa[j + 1 + q] : -0.0;
                }
                if (j > 0) {
                    const double *cn = r + (size_t)(j - 1) * ldr + j;
#pragma unroll
                    for (int u = 0; u < E; ++u) { const int q = E * lane + u; nx[u] = q < len + 1 ? cn[q] : 0.0; }
                }
                const double sm = ordered_sum_wave_regs<E>(d, len, 0.0);
                if (lane == 0) wa[j] = (wa[j] - sm) / sdiag[j];
            }
        }
        __syncthreads();
    } else
    for (int k = 1; k <= ns; ++k) {
        const int j = ns - k;
        const double *colj = r + (size_t)j * ldr;
        if (EXACT && n <= 3 * NLH_NCH) {
            // products in parallel, then their ascending sum by one thread (the reference's dot product, :779)
            const int len = ns - j - 1;
            for (int i = tid; i < len; i += BS) sbuf[i] = colj[j + 1 + i] * wa[j + 1 + i];
            __syncthreads();
            if (tid == 0) wa[j] = (wa[j] - ordered_sum_lds(sbuf, len)) / sdiag[j];
            __syncthreads();
        } else {
            const double sm = sum_block<EXACT>([&](int i) { return colj[j + 1 + i] * wa[j + 1 + i]; }, ns - j - 1, red);
            __syncthreads();
            if (tid == 0) wa[j] = (wa[j] - sm) / sdiag[j];
            __syncthreads();
        }
    }
    for (int j = tid; j < n; j += BS) x[ipvt[j]] = wa[j];       // :787-790
    __syncthreads();
    LMCLK(2)
}

// lmpar.  Vectors x, sdiag, wa1, wa2n (the first n entries of the caller's wa4), z are LDS.
// Deviation A (:531) takes the norm over all m entries of the caller's wa4: EXACT reads the
// tail wa4[n..m) itself, otherwise tailsq = its sum of squares.  ne_mode: the factors come
// from the Gram matrix (row signs unknown, no Q^T f tail), so only the Gauss-Newton
// acceptance test is run; returns 1 when the iteration would be needed (=> QR fallback).
template <bool EXACT>
__device__ __forceinline__ int lmpar_dev(int m, int n, double *r, int ldr, const int32_t *ipvt, const double *diag,
                         const double *qtb, double delta, double *par_io, double tailsq,
                         const double *wa4, double *x, double *sdiag, double *wa1, double *wa2n,
                         double *z, double *red, double *scratch, double *Wrows, double *rot, int ne_mode,
                         double *ring = nullptr, int ringcap = 0)
{
    const int tid = threadIdx.x, BS = blockDim.x;
    const double p1 = 0.1, p001 = 1.0e-3;
    double par = *par_io;
    LMCLK_START()

    // Gauss-Newton direction (:447-469)
    int nsing = n;
    for (int j = tid; j < n; j += BS)
        if (r[(size_t)j * ldr + j] == 0.0) nsing = min(nsing, j);
    nsing = -(int)block_reduce_max((double)(-nsing), red);
    for (int j = tid; j < n; j += BS) wa1[j] = (j < nsing) ? qtb[j] : 0.0;
    __syncthreads();
    {
        // column j-1 (first element per thread and the diagonal) is fetched while column j is applied, so that a
        // step costs a barrier instead of a dependent global load
        double pre = 0.0, dpre = 1.0;
        if (nsing > 0) {
            const int j0 = nsing - 1;
            pre = (tid < j0) ? r[(size_t)j0 * ldr + tid] : 0.0;
            dpre = r[(size_t)j0 * ldr + j0];
        }
        for (int k = 1; k <= nsing; ++k) {
            const int j = nsing - k;
            const double cur = pre, dcur = dpre;
            if (j > 0) {
                pre = (tid < j - 1) ? r[(size_t)(j - 1) * ldr + tid] : 0.0;
                dpre = r[(size_t)(j - 1) * ldr + j - 1];
            }
            const double temp = wa1[j] / dcur;
            const double *colj = r + (size_t)j * ldr;
            if (tid < j) wa1[tid] = wa1[tid] - cur * temp;
            for (int i = tid + BS; i < j; i += BS) wa1[i] = wa1[i] - colj[i] * temp;
            if (tid == 0) z[j] = temp;
            __syncthreads();
        }
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
    LMCLK(3)
    if (fp <= p1 * delta) { *par_io = 0.0; return 0; }
    if (ne_mode) return 1;

    // lower bound parl (:486-503)
    double parl = 0.0, temp;
    if (nsing == n) {
        __syncthreads();
        for (int j = tid; j < n; j += BS) { const int l = ipvt[j]; wa1[j] = diag[l] * (wa2n[l] / dxnorm); }
        __syncthreads();
        if (EXACT && n <= 3 * NLH_NCH) {
            // forward substitution with one accumulator per row: row j adds r(i,j) * wa1(i) as soon as wa1(i) is
            // final, i.e. in ascending i -- the order of the reference's inner loop (:495-499) -- for all rows at once
            for (int j = tid; j < n; j += BS) scratch[j] = 0.0;
            __syncthreads();
            if (tid == 0) wa1[0] = (wa1[0] - 0.0) / r[0];
            __syncthreads();
            for (int i = 0; i + 1 < n; ++i) {
                const double wi = wa1[i];
                for (int j = i + 1 + tid; j < n; j += BS) {
                    const double *colj = r + (size_t)j * ldr;
                    const double acc = scratch[j] + colj[i] * wi;
                    scratch[j] = acc;
                    if (j == i + 1) wa1[j] = (wa1[j] - acc) / colj[j];
                }
                __syncthreads();
            }
        } else {
            for (int j = 0; j < n; ++j) {
                const double *colj = r + (size_t)j * ldr;
                const double sm = sum_block<EXACT>([&](int i) { return colj[i] * wa1[i]; }, j, red);
                __syncthreads();
                if (tid == 0) wa1[j] = (wa1[j] - sm) / colj[j];
                __syncthreads();
            }
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

    bool at_fixed_point = false;
    LMCLK(4)
    for (int iter = 1;; ++iter) {                              // :522-563
        // Once an iteration has STARTED with par = +Inf and run to its end, the loop is at a fixed point: the next
        // iteration starts from the same par = parl = +Inf, the same R and qtb (lmsolve restarts from them), computes
        // the same x (= 0), sdiag, dxnorm and fp, cannot exit before iter == 10 (fp repeats) and leaves par = +Inf
        // again.  Everything lmpar returns is therefore already what the tenth iteration would leave, bit for bit;
        // the remaining (identical) sweeps are skipped.  This is where deviation A sends every problem whose trust
        // region has shrunk below the residual tail it adds to ||D x||: par explodes super-exponentially (4.2 ->
        // 1.9e19 -> 8.3e74 -> 7.3e241 -> Inf on the benchmark family) and the reference spends five more sweeps there.
        if (at_fixed_point && par == __builtin_inf()) break;
        const bool started_inf = (par == __builtin_inf());
        if (par == 0.0) par = fmax(NLH_DWARF, p001 * paru);
        temp = sqrt(par);
        __syncthreads();
        for (int i = tid; i < n; i += BS) wa1[i] = temp * diag[i];
        __syncthreads();
        LMCLK(5)
        lmsolve_dev<EXACT>(n, r, ldr, ipvt, wa1, qtb, x, sdiag, wa2n, red, Wrows, z, rot, scratch, ring, ringcap);
        LMCLK(6)
#ifdef NLH_DEBUG_LMPAR_CLK
        if (threadIdx.x == 0) g_lmclk[10] += 1;                 // lmsolve calls
#endif
        for (int i = tid; i < n; i += BS) wa2n[i] = diag[i] * x[i];
        __syncthreads();
        if (EXACT) {                                           // :531 deviation A: norm over all m entries
            // NORM2 with its recurrence down the lanes of a wave (nlh_common.h; bit-identical to norm2_flang_block, whose
            // one-thread recurrence cost 130 us per lmpar iteration at m = 4096 -- a third of an iteration at n = 96)
            auto g = [&](int i) { return i < n ? wa2n[i] : wa4[i]; };
            if (BS == 1024) dxnorm = norm2_flang_block_lanes<32, 1024>(g, m, scratch, scratch + 64 * 32 + 128);
            else if (BS == 512) dxnorm = norm2_flang_block_lanes<16, 512>(g, m, scratch, scratch + 64 * 16 + 128);
            else dxnorm = norm2_flang_block_lanes<16, 256>(g, m, scratch, scratch + 64 * 16 + 128);
        } else {
            double sq = 0.0;
            for (int i = tid; i < n; i += BS) sq = sq + wa2n[i] * wa2n[i];
            dxnorm = sqrt(block_reduce_sum(sq, red) + tailsq);
        }
        temp = fp;
        fp = dxnorm - delta;
        LMCLK(7)
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
        LMCLK(8)
        temp = nrm2_block<EXACT>([&](int j) { return wa1[j]; }, n, red, scratch);
        LMCLK(9)
        const double parc = ((fp / delta) / temp) / temp;
        if (fp > 0.0) parl = fmax(parl, par);                  // :558-559
        if (fp < 0.0) paru = fmin(paru, par);
        par = fmax(parl, par + parc);                          // :562
        at_fixed_point = started_inf;
    }
    *par_io = par;
#ifdef NLH_DEBUG_LMPAR_CLK
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        printf("[lmpar clk, us] n %d, %llu lmsolve calls: copy %.1f sweep %.1f backsub %.1f | GN %.1f bounds %.1f pre %.1f lmsolve(all) %.1f dxnorm %.1f newton-solve %.1f nrm %.1f\n",
               n, g_lmclk[10], g_lmclk[0] * 1e-2, g_lmclk[1] * 1e-2, g_lmclk[2] * 1e-2, g_lmclk[3] * 1e-2, g_lmclk[4] * 1e-2, g_lmclk[5] * 1e-2,
               g_lmclk[6] * 1e-2, g_lmclk[7] * 1e-2, g_lmclk[8] * 1e-2, g_lmclk[9] * 1e-2);
        for (int i = 0; i < 16; ++i) g_lmclk[i] = 0;
    }
#endif
    return 0;
}

// Normal-equations path, trust region binding.  lmpar's iteration is not invariant under the
// row signs of R (its deviation B mixes rows of R with rows of S, :552), so the Cholesky factor
// R~ (positive diagonal) has to be given lmfactor's signs: R = S R~, qtf = S qtf~ with
// s_j = -sign of the j-th pivot entry Householder sees.  Those signs follow from the top n-by-n
// block of the positive thin Q, W = (J P)(1:n,:) R~^-1, by a sign-tracking LU of W - S
// (Householder reconstruction): s_j = -sgn(W_jj) after j-1 elimination steps, pivot W_jj - s_j.
// O(n^3) on n-by-n data instead of an O(m n^2) QR.  W: n*n global scratch; sg, rowj, lcol: LDS n.
__device__ void ne_recover_signs(int m, int n, const double *J, const int32_t *ipvt, double *R,
                                 double *qtf, double *W, double *sg, double *rowj, double *lcol)
{
    const int tid = threadIdx.x, BS = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = BS >> 6;
    for (int i = tid; i < n; i += BS) {                         // row i of W = (row i of J P) R~^-1
        for (int c = 0; c < n; ++c) {
            double acc = J[(size_t)ipvt[c] * m + i];
            const double *Rc = R + (size_t)c * n;
            int k = 0;
            for (; k + 8 <= c; k += 8) {
                double wv[8], rv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { wv[u] = W[(size_t)(k + u) * n + i]; rv[u] = Rc[k + u]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = acc - wv[u] * rv[u];
            }
            for (; k < c; ++k) acc = acc - W[(size_t)k * n + i] * Rc[k];
            W[(size_t)c * n + i] = acc / Rc[c];
        }
    }
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        const double wjj = W[(size_t)j * n + j];
        const double sj = (wjj < 0.0) ? 1.0 : -1.0;             // lmfactor :644 / :665
        const double piv = wjj - sj;
        for (int c = j + 1 + tid; c < n; c += BS) {
            rowj[c] = W[(size_t)c * n + j];
            lcol[c] = W[(size_t)j * n + c] / piv;
        }
        if (tid == 0) sg[j] = sj;
        __syncthreads();
        constexpr int CG = 4;
        for (int c0 = j + 1 + wid * CG; c0 < n; c0 += nw * CG) {
            for (int i = j + 1 + lane; i < n; i += 64) {
                const double li = lcol[i];
                double wv[CG];
#pragma unroll
                for (int u = 0; u < CG; ++u) wv[u] = (c0 + u < n) ? W[(size_t)(c0 + u) * n + i] : 0.0;
#pragma unroll
                for (int u = 0; u < CG; ++u)
                    if (c0 + u < n) W[(size_t)(c0 + u) * n + i] = wv[u] - li * rowj[c0 + u];
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < n * n; e += BS) {                     // R = S R~ (upper triangle), qtf = S qtf~
        const int i = e % n, c = e / n;
        if (i <= c && sg[i] < 0.0) R[e] = -R[e];
    }
    for (int j = tid; j < n; j += BS)
        if (sg[j] < 0.0) qtf[j] = -qtf[j];
    __syncthreads();
}

// lmpar for every problem whose factors are ready, then the step and trial point.
// Dynamic LDS: (6n + 72) doubles, plus 3*NLH_NCH + 8 when EXACT.
template <bool EXACT, bool GV = false>
__global__ void __launch_bounds__(1024)
k_lmpar(int m, int n, double *__restrict__ Rall, LmVecs v, double *__restrict__ xall,
        const double *__restrict__ wa4all, double *__restrict__ Wall /* [nprob][m*n] scratch */,
        const double *__restrict__ Jall, double *__restrict__ W2all /* [nprob][n*n] scratch */,
        LmState *__restrict__ st, int want_stage, double *__restrict__ gv = nullptr, int ringcap = 0)
{   // ringcap > 0: lmsolve's sweep on chip, its ring behind the other LDS vectors (lmsolve_ring_doubles(n, threads) doubles).
    // GV: lmpar's n-vectors in [nprob][6 n + 8] doubles of global memory (gv) when they do not fit LDS (n > 3000; the
    // reference allocates for any n, src/nonlin_least_squares.f90:199-208): the same code through other pointers --
    // a workgroup's barriers order its own global accesses -- at L2 instead of LDS latency.  A compile-time switch: with
    // a run-time choice of the base the compiler loses the vectors' address space and every access to them becomes a
    // flat_load / flat_store instead of ds_read / ds_write (round 5: the all-accepted path 1.41 -> 2.00 ms).
    extern __shared__ double smem[];
    const int p = blockIdx.x;
    LmState *s = st + p;
    if (s->stage != want_stage) return;
    const int tid = threadIdx.x, BS = blockDim.x;
    double *nv = GV ? gv + (size_t)p * (6 * (size_t)n + 8) : smem;
    double *xs = nv, *sdiag = nv + n, *wa1 = nv + 2 * n, *wa2n = nv + 3 * n, *z = nv + 4 * n;
    double *rot = nv + 5 * n;           // n + 8
    double *red = GV ? smem : smem + 6 * n + 8;
    double *scratch = red + 64;
    double *ring = ringcap > 0 ? scratch + (EXACT ? lmpar_scratch_doubles(BS) : 0) : nullptr;
    double *R = Rall + (size_t)p * n * n;
    const int32_t *ipvt = v.ipvt + (size_t)p * n;
    const double *diag = v.diag + (size_t)p * n;
    const double *qtf = v.qtf + (size_t)p * n;
    const double *xc = xall + (size_t)p * n;
    const int ne_mode = !EXACT && (want_stage == ST_NE_READY);
    double *Wp = Wall + (size_t)p * m * n;

    double par = s->par;
    const double delta = s->delta;
    // ONE call site of lmpar_dev (it is inlined -- as a called function its LDS pointers would be generic ones): the
    // normal-equations policy's second pass, after the signs are recovered, is a second trip round this loop
    int rc, mode = ne_mode;
    double tailsq_in = s->tailsq;
    for (;;) {
        rc = lmpar_dev<EXACT>(m, n, R, n, ipvt, diag, qtf, delta, &par, tailsq_in,
                              wa4all + (size_t)p * m, xs, sdiag, wa1, wa2n, z, red, scratch,
                              Wp, rot, mode, ring, ringcap);
        __syncthreads();
        if (EXACT || !rc || !mode) break;
        // Deviation A (:531) adds ||wa4(n+1:m)|| to ||D x|| inside the loop.  If that tail alone exceeds
        // 1.1*delta the exit test |fp| <= 0.1*delta can never pass for any par: the reference runs its ten
        // iterations with par growing super-exponentially to +Inf and returns x = 0 (the solve then ends
        // with converge_on_chng because delta becomes 0).  Produce that outcome directly instead of ten
        // lmsolve sweeps; the exact policy does not take this shortcut.
        double tq = s->tailsq;
        if (s->inner_pass == 0) {
            const double *qt = v.qtf + (size_t)p * n;
            double q2 = 0.0;
            for (int j = tid; j < n; j += BS) q2 = q2 + qt[j] * qt[j];
            q2 = block_reduce_sum(q2, red);
            const double f2 = s->fnorm * s->fnorm;
            tq = f2 > q2 ? f2 - q2 : 0.0;
        }
        if (sqrt(tq) > 1.1 * delta * (1.0 + 1.0e-6)) {
            __syncthreads();
            for (int j = tid; j < n; j += BS) xs[j] = 0.0;
            par = __builtin_inf();
            rc = 0;
            __syncthreads();
            break;
        }
        if (!s->pivoted) {
            // the lmpar iteration needs lmfactor's pivot order: ask for the pivoted factorisation
            if (tid == 0) s->stage = ST_NEED_PCHOL;
            return;
        }
        // Gauss-Newton step rejected on the normal-equations path: give the factors lmfactor's
        // signs, reconstruct ||(Q^T f)(n+1:m)||^2 = ||f||^2 - ||qtf||^2 for deviation A on the
        // first inner pass (later passes use the rejected trial residual's tail), run full lmpar.
        double tailsq = s->tailsq;
        double *qtfw = v.qtf + (size_t)p * n;
        if (!s->signs_done) {
            ne_recover_signs(m, n, Jall + (size_t)p * m * n, ipvt, R, qtfw, W2all + (size_t)p * n * n, xs, sdiag, wa1);
        }
        if (s->inner_pass == 0) {
            double q2 = 0.0;
            for (int j = tid; j < n; j += BS) q2 = q2 + qtfw[j] * qtfw[j];
            q2 = block_reduce_sum(q2, red);
            const double f2 = s->fnorm * s->fnorm;
            tailsq = f2 > q2 ? f2 - q2 : 0.0;
        }
        __syncthreads();
        if (tid == 0) { s->signs_done = 1; s->tailsq = tailsq; }
        par = s->par;
        tailsq_in = tailsq;
        mode = 0;
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
    for (int j = tid; j < n; j += BS) wa1[j] = xs[ipvt[j]];    // P^T p (wa1 is free after pnorm)
    __syncthreads();
    for (int i = tid; i < n; i += BS) {
        double acc = 0.0;
        int j = i;
        for (; j + 8 <= n; j += 8) {                           // eight loads in flight, adds in column order
            double rv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) rv[u] = R[(size_t)(j + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = acc + rv[u] * wa1[j + u];
        }
        for (; j < n; ++j) acc = acc + R[(size_t)j * n + i] * wa1[j];
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
        if (par != 0.0) { s->slow_lmpar += 1; if (!s->first_slow) s->first_slow = s->iter; }   // (par == 0: the early exit of :481)
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
        // NORM2 down the lanes of a wave (nlh_common.h; bit-identical to norm2_flang_block, whose one-thread recurrence
        // took 110 of this kernel's 128 us at m = 4096)
        __shared__ __attribute__((aligned(16))) double ncd[EXACT ? 64 * 64 + 128 : 2];
        __shared__ __attribute__((aligned(16))) double naux[EXACT ? 40 + 128 : 2];
        const double *w = wa4 + (size_t)p * m;
        fnorm1 = norm2_flang_block_lanes<64, 256>([&](int i) { return w[i]; }, m, ncd, naux);
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
    else if (accept) { s->stage = ST_NEED_JAC; s->inner_pass = 0; s->head_done = 0; s->signs_done = 0; s->pivoted = 0; }
    else {                                                      // inner loop again
        s->stage = (fkind == 1) ? ST_QR_READY : ST_NE_READY;
        s->rejects += 1;
        if (!s->first_slow) s->first_slow = iter;
    }
}
