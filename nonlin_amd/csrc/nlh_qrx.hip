// nlh_qrx.hip -- lmfactor + Q^T f in the reference's operation order (policy NLH_FACTOR_EXACT),
// streaming form: the whole batch advances through the Householder steps in lock step, two launches
// per step, and the trailing matrices stream from HBM once per step.
//
// Reference: src/nonlin_least_squares.f90:569-667 (lmfactor), :241-253 (Q^T f).  Bit-identical to the
// CPU path: a trailing column's dot product with the reflector (:652-653) is summed by ONE thread in
// ascending row order, every elementwise operation is the reference's, NORM2 is the flang runtime's
// algorithm (nlh_common.h).
//
// Why this shape.  A Householder step needs every trailing column's complete dot product before the next
// pivot is known (the pivot rule looks at the down-dated norms, :657-661), so one full pass over the
// trailing matrix per step is inherent; a 4096 x 256 problem is 8 MiB and a batch of 512 is 4 GiB, far
// beyond LDS + registers (168 MB on the chip) or the Infinity Cache (256 MiB), and a resident subset of
// problems would leave most SIMDs idle behind the serial row recurrences.  The floor is therefore HBM:
// 8 * sum_j (m - j)(n - j + 1) bytes per problem (1.06 GB at 4096 x 256).  The kernels are built for that floor:
//   * the working matrix T is ROW-BLOCKED (qrx_at: eight rows of a column per 64-byte sector, the residual rides along
//     as the last column); one lane owns one trailing column and walks down the rows: two rows per 16-byte load through
//     a buffer descriptor (scalar block offset + per-lane column offset, no address arithmetic), a wave reads 4 KB
//     contiguous per 8-row block, 32 rows of loads in flight ahead of the arithmetic, one wave per workgroup, no
//     barrier between waves;
//   * block rows END on a 64-column boundary (qrx_coff), so the trailing columns of any step fill whole 64-column
//     windows counted from the end and a step launches exactly ceil(live columns / 64) waves per problem;
//   * reflector entries are wave-uniform: each wave stages a 64-row tile of them in LDS one tile ahead
//     (coalesced reads of the slot vectors) and reads a row's entries back as broadcast ds_reads, issued one
//     row pair ahead of the arithmetic; two rows are processed together so that the dependent mul / sub chains
//     of the pending updates interleave;
//   * column updates are DEFERRED: after step j a trailing column is not rewritten; its multiplier
//     t_k = s_k / a_jj is kept and later passes apply the pending updates on the fly, oldest first --
//     e = ((a - t_0 v_0) - t_1 v_1) ... -- the very roundings of the eager update (:655).  Every 7th step
//     the pass stores e back (flush, whole 64-byte sectors): 8 B read + 8/7 B written per element and step instead of 24.
//     (Measured: 12 or 16 slots per bank are slower -- the extra multiply / subtract pairs cost more than the
//     flushes they save; a flushing pass runs at 4.8 TB/s, a plain one at 5.7-6.0 TB/s of the 6.3 achievable.)
//   * the column interchange (:626-637) never moves data: slot k of the permuted matrix carries a source
//     column index src[k]; the pivot column is consumed into the reflector and slot kmax simply inherits
//     slot j's source and pending multipliers.  The flush writes every slot to its own position.
//   * RULE: no workgroup of a pass writes anything another workgroup of the same launch reads.  Workgroups of one
//     launch need not run at the same time (more of them than the chip holds, or rows so few that one ends before a
//     later one starts).  The slot maps in particular are written only by the pivot kernel (one workgroup per
//     problem): after a flush the NEXT pivot kernel resets them to the identity (its `fresh` flag).
// Launches per step: k_qrx_pivot (one workgroup per problem: pivot search, bookkeeping, gather of the pivot
// column with its pending updates, NORM2 -- a serial chain of m - j adds --, scaling -> reflector) and
// k_qrx_pass (lane per trailing column; k_qrx_pass_rp, the row-parallel form, when the launch cannot fill the chip:
// a few problems still iterating, one problem alone, the last narrow steps).  The driver keeps several sub-batches in flight on private streams
// (nlh_lm.hip, lm_sub_batches) so that one sub-batch's pivot kernels run under another's passes.
#include "nlh_qrx.h"
#include "nlh_common.h"
#include <type_traits>
#include <cstdlib>
#include <algorithm>

#ifndef QRX_C
#define QRX_C 10           // reflector slots per bank = pending updates before a flush + 1 (measured, 2048 x 4096x256,
                           // ms per factorisation of the batch, windows as separate workgroups: 8: 560, 9: 539, 10: 545,
                           // 12: 539, 14: 540, 16: 573, 20: 609; windows as the waves of one workgroup, which makes
                           // the flushing passes 10-20 % cheaper: 8: 520, 10: 510, 12: 525, 14: 522)
#endif
#define QRX_TR 64           // rows per reflector tile of the pass
#define QRX_PAD_ROWS 160   // read-ahead padding behind the last problem's matrix (a tile + a load group)
#ifndef QRX_AUX_LOAD
#define QRX_AUX_LOAD 0    // cache policy of the matrix stream: default.  Non-temporal loads (2) were measured: 1137 vs 836 ms
#endif                    // per 512 x 4096x256 solve
#ifndef QRX_AUX_STORE
#define QRX_AUX_STORE 1   // the flush: sc0 (measured, 512 x 4096x256 on one box: default policy 727 ms, sc0 714, nt 740,
#endif                    // sc0 sc1 847, sc0 sc1 nt 1217)
#ifndef QRX_UF
#define QRX_UF(flush) 16
#endif
// The flushing pass, measured at 2048 x 4096x256, full width (profiles/ubench/rw_stream.hip, w_stream.hip reproduce the
// memory-system side): 8.7 ms against 2.9 ms for a plain pass; without its stores 3.3 ms.  A write-only stream in the
// flush's shape (16 bytes per lane at a 64-byte stride) runs at 3.7 TB/s, line-contiguous stores at 5.5 TB/s -- but a
// kernel that reads AND writes in place reaches 4.8-5.0 TB/s of combined traffic whichever shape its stores have
// (read-only: 6.1 TB/s), and the flush with its sectors transposed through LDS into 1 KB-contiguous stores took the
// same 8.7 ms.  What a flush costs is set by mixed read/write traffic; the lever is fewer flushes (QRX_C).
// NORM2 of the pivot kernel: elements per lane of the serial phase = chunk / 64.  64 (chunks of 4096) when columns are
// longer than 2048, 32 otherwise: half the LDS and 64 registers less, four workgroups per CU instead of two.

typedef unsigned int qrx_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int qrx_u32x4 __attribute__((ext_vector_type(4)));
struct QrxStep { double ajnorm, ajj; int32_t kmax, srck; double t0; };   // srck, t0: pivot column and its pending multiplier (split long-column step)

// Physical column of slot k (k = 0 .. n, n = the residual) is k + qrx_coff(n): the row ENDS on a 64-column boundary, so
// that the trailing slots j+1 .. n of any step fill whole 64-column windows counted from the end: every wave-level load
// and store of the pass is one aligned 512-byte span, and the number of waves is exactly ceil((n - j) / 64).
static int qrx_coff(int n) { return (64 - ((n + 1) & 63)) & 63; }
int qrx_ld(int n) { return n + 1 + qrx_coff(n); }

static size_t qrx_tstride(int m, int n) { return (size_t)((m + 7) & ~7) * qrx_ld(n); }   // doubles between two problems' matrices

// Element (row i, physical column c) of a working matrix.  Row-blocked: eight consecutive rows of a column share one
// 64-byte sector ((i / 8) * ld + c is the sector index), so a lane reads its column two rows per 16-byte load, a wave
// reads 4 KB contiguous per row block, and a walk down one column (the pivot gather) costs a sector per 8 elements.
__host__ __device__ __forceinline__ size_t qrx_at(int i, int c, int ld) { return ((size_t)(i >> 3) * ld + c) * 8 + (i & 7); }
static size_t qrx_vstride(int m)      // doubles between two slots of a reflector bank
{
    return ((size_t)m + 7) & ~(size_t)7;
}

size_t qrx_matrix_stride(int m, int n) { return qrx_tstride(m, n); }

size_t qrx_matrix_doubles(int nprob, int m, int n)
{
    return (size_t)nprob * qrx_tstride(m, n) + (size_t)QRX_PAD_ROWS * qrx_ld(n);
}

struct QrxWs {
    double *V;         // [nprob][2][QRX_C][vst]: reflector banks, one contiguous vector per slot
    double *tp;        // [nprob][2][QRX_C][n + 1]
    double *rdiag;     // [nprob][n]
    double *wa;        // [nprob][n]
    QrxStep *step;     // [nprob]
    int32_t *src;      // [nprob][n + 1]: physical column holding slot k
    int32_t *slotof;   // [nprob][ld]: slot held by a physical column, -1 = consumed / never used
    int32_t *plist;    // [nprob]: the problems a factorisation works on, compacted (k_qrx_list), -1 = none; nullptr = all
    int ny;            // ... and how many (host side)
};

static size_t qrx_carve(void *base, int nprob, int m, int n, QrxWs *w)
{
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t oV = take(sizeof(double) * ((size_t)nprob * 2 * QRX_C * qrx_vstride(m) + 2 * QRX_TR));   // + read-ahead of the last slot
    const size_t otp = take(sizeof(double) * (size_t)nprob * 2 * QRX_C * (n + 1));
    const size_t ord = take(sizeof(double) * (size_t)nprob * n);
    const size_t owa = take(sizeof(double) * (size_t)nprob * n);
    const size_t ost = take(sizeof(QrxStep) * (size_t)nprob);
    const size_t osr = take(sizeof(int32_t) * (size_t)nprob * (n + 1));
    const size_t oso = take(sizeof(int32_t) * (size_t)nprob * qrx_ld(n));
    const size_t opl = take(sizeof(int32_t) * (size_t)nprob);
    if (w) {
        char *b = (char *)base;
        w->V = (double *)(b + oV); w->tp = (double *)(b + otp); w->rdiag = (double *)(b + ord);
        w->wa = (double *)(b + owa); w->step = (QrxStep *)(b + ost); w->src = (int32_t *)(b + osr);
        w->slotof = (int32_t *)(b + oso);
        w->plist = (int32_t *)(b + opl);
    }
    return off;
}

size_t qrx_workspace_bytes(int nprob, int m, int n) { return qrx_carve(nullptr, nprob, m, n, nullptr); }

// Column-major m x n  ->  the row-blocked working matrix: eight rows of a column are contiguous in both layouts, so a
// thread moves one 64-byte sector (consecutive threads: consecutive columns, i.e. contiguous writes).
__global__ void __launch_bounds__(256)
k_qrx_transpose(int m, int n, int ld, int coff, size_t tst, const double *__restrict__ J, double *__restrict__ T,
                const LmState *__restrict__ st)
{
    const int p = blockIdx.y;
    if (st && st[p].stage != ST_NEED_QR) return;
    const double *Jp = J + (size_t)p * m * n;
    double *Tp = T + (size_t)p * tst;
    const int nblk = (m + 7) >> 3;
    const size_t total = (size_t)nblk * n;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(e % n), b = (int)(e / n);
        const double *src = Jp + (size_t)k * m + (size_t)b * 8;
        double *dst = Tp + ((size_t)b * ld + coff + k) * 8;
        double vv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) vv[r] = (b * 8 + r < m) ? src[r] : 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) dst[r] = vv[r];
    }
}

// wa4 = fvec as column n (:241), initial column norms (:611-616), identity maps.
// Initial column norms (:611-616) for a HANDFUL of problems: a workgroup per column (the serial recurrence of flang's
// NORM2 down the lanes of a wave, as in the pivot kernel) instead of a thread per column -- one 65536 x 512 problem:
// 25 ms -> well under a millisecond.  Same algorithm on the same elements in the same order: the same bits.
// The problems due for a factorisation (stage ST_NEED_QR), compacted in ascending order: the launches whose grids are
// (columns x problems) -- the column sweep, the initial norms -- then cover the ny problems that work instead of every
// problem of the batch.  In the straggler rounds of a 2048-problem batch a column-sweep launch was 175,000 workgroups of
// which 256 had work: 124 us per launch against 18 for a problem alone (1.4 % of the headline step).  ny is the host's
// count (exact: the problems that were at ST_NEED_JAC when the round before ended); entries past the true count are -1,
// and a problem past ny -- which cannot happen -- would be retired with its flag set rather than silently skipped.
__global__ void __launch_bounds__(256)
k_qrx_list(int nprob, int ny, LmState *__restrict__ st, int32_t *__restrict__ plist)
{
    __shared__ int wsum[4], base;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int p0 = 0; p0 < nprob; p0 += 256) {
        const int p = p0 + tid;
        const bool on = p < nprob && st[p].stage == ST_NEED_QR;
        const unsigned long long b = __ballot(on);
        const int before = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wid] = __popcll(b);
        __syncthreads();
        int off = base;
        for (int w2 = 0; w2 < wid; ++w2) off += wsum[w2];
        if (on) {
            const int pos = off + before;
            if (pos < ny) plist[pos] = p;
            else { st[p].flag = 1; st[p].stage = ST_DONE; }
        }
        __syncthreads();
        if (tid == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    for (int i = base + tid; i < ny; i += 256) plist[i] = -1;
}

__global__ void __launch_bounds__(256)
k_qrx_init_norms(int m, int n, int ld, int coff, size_t tst, const double *__restrict__ T, QrxWs w, LmVecs v,
                 const LmState *__restrict__ st)
{
    __shared__ __attribute__((aligned(16))) double cd[64 * 64 + 128];
    __shared__ __attribute__((aligned(16))) double aux[40 + 128];
    const int p = w.plist ? w.plist[blockIdx.y] : (int)blockIdx.y, k = blockIdx.x;
    if (p < 0 || (st && st[p].stage != ST_NEED_QR)) return;
    const double *a = T + (size_t)p * tst;
    const double nr = norm2_flang_block_lanes<64, 256>([&](int i) { return a[qrx_at(i, coff + k, ld)]; }, m, cd, aux);
    if (threadIdx.x == 0) {
        v.acnorm[(size_t)p * n + k] = nr;
        w.rdiag[(size_t)p * n + k] = nr;
        w.wa[(size_t)p * n + k] = nr;
        v.ipvt[(size_t)p * n + k] = k;
    }
}

template <bool NORMS>
__global__ void __launch_bounds__(256)
k_qrx_init(int m, int n, int ld, int coff, size_t tst, double *__restrict__ T, const double *__restrict__ fall, QrxWs w, LmVecs v,
           const LmState *__restrict__ st)
{
    const int p = blockIdx.x;
    if (st && st[p].stage != ST_NEED_QR) return;
    const int tid = threadIdx.x, BS = blockDim.x;
    double *a = T + (size_t)p * tst;
    const double *f = fall + (size_t)p * m;
    for (int i = tid; i < m; i += BS) a[qrx_at(i, coff + n, ld)] = f[i];
    for (int k = tid; NORMS && k < n; k += BS) {
        // flang NORM2 of column k, rows ascending: a sector (8 rows) per load group, the next one in flight
        double mx = 0.0, sq = 0.0;
        auto step = [&](double vv) {
            const double av = fabs(vv);
            if (mx == 0.0) {
                mx = av;
            } else if (av > mx) {
                const double t = mx / av, tsq = t * t;
                sq = sq * tsq;
                sq = sq + tsq;
                mx = av;
            } else if (av != 0.0) {
                const double t = av / mx;
                sq = sq + t * t;
            }
        };
        const int nblk = (m + 7) >> 3;
        const double *colp = a + (size_t)(coff + k) * 8;
        double xa[8], xb[8];
        auto ld8 = [&](double (&x)[8], int b) {
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] = colp[(size_t)b * ld * 8 + r];
        };
        ld8(xa, 0);
        for (int b = 0; b < nblk; b += 2) {
            if (b + 1 < nblk) ld8(xb, b + 1);
#pragma unroll
            for (int r = 0; r < 8; ++r) if (b * 8 + r < m) step(xa[r]);
            if (b + 2 < nblk) ld8(xa, b + 2);
            if (b + 1 < nblk) {
#pragma unroll
                for (int r = 0; r < 8; ++r) if ((b + 1) * 8 + r < m) step(xb[r]);
            }
        }
        const double nr = mx * sqrt(1.0 + sq);
        v.acnorm[(size_t)p * n + k] = nr;
        w.rdiag[(size_t)p * n + k] = nr;
        w.wa[(size_t)p * n + k] = nr;
        v.ipvt[(size_t)p * n + k] = k;
    }
    for (int k = tid; k <= n; k += BS) w.src[(size_t)p * (n + 1) + k] = coff + k;    // physical column of slot k
    for (int c = tid; c < ld; c += BS) w.slotof[(size_t)p * ld + c] = c >= coff ? c - coff : -1;
}

// Step j, part 1: pivot (:622-637), the pivot column with its pending updates -> reflector (:642-646).
// The new reflector goes to slot np of the current bank, or to slot 0 of the other bank when this step's pass
// flushes (np == QRX_C - 1).
#define QRX_LONG_EL 48                                           // the pipelined NORM2: chunks of 64 * 48 rows, three preparing waves
                                                                 // (chunks of 4096 rows with four preparing waves, 320 threads, also as the
                                                                 // stand-alone k_qrx_norm_long: 199 against 195 ms per 65536 x 512 factorisation)
#define QRX_LONG_THREADS 256                                     // (512 -- four more waves for the gather and the scaling -- measured slower: 250 against 241 us per 65536-row step)
#ifndef QRX_LONG_GU
#define QRX_LONG_GU 8                                            // row pairs per thread in flight in the long-column gather
#endif
#ifndef QRX_LONG_SU
#define QRX_LONG_SU 4                                            // sectors per thread in flight in the long-column scaling
#endif
#ifndef QRX_PIV64_WG
#define QRX_PIV64_WG 1                                           // workgroups per CU the batch instance of the 4096-row pivot kernel is compiled for
#endif
#define QRX_LONG_MAXCH 96                                        // chunks the pipelined NORM2 keeps maxima for
template <int QRX_NL, bool LONG = false, bool FEW = false>       // FEW: a handful of problems (the workgroup has its CU to itself)
// (QRX_NL = 32, m <= 2048: 133 registers would leave three workgroups per CU; held to 128 -- four dwords spilled -- a launch of
// 1024 problems is one round instead of two: 1024 x 2048x128 solves 1.8 % faster)
__global__ void __launch_bounds__(LONG ? QRX_LONG_THREADS : 256, (QRX_NL == 32 && !FEW) ? 4 : (QRX_NL == 64 && !FEW && !LONG) ? QRX_PIV64_WG : 1)
k_qrx_pivot(int p0, int m, int n, int ld, int coff, size_t tst, size_t vst, int j, int cur, int np, int flush, double *__restrict__ T, QrxWs w,
            double *__restrict__ Rall, LmVecs v, const LmState *__restrict__ st)
{
    // LONG (the column is more than one NORM2 chunk): two coefficient buffers and the per-chunk maxima for the
    // pipelined NORM2 -- three preparing waves and the chain wave, one wave per SIMD (with four preparing waves, 320
    // threads, the chain wave shared its SIMD with one of them: 253 us per 65536-row step against 241), 105 KB of LDS,
    // one workgroup per CU, which is all a handful of long-column problems need
    __shared__ __attribute__((aligned(16))) double cd[LONG ? 2 * (64 * QRX_LONG_EL + 128) : (64 * QRX_NL + (FEW ? 256 : 128))];
    __shared__ __attribute__((aligned(16))) double aux[LONG ? 8 + 256 : 40 + 128];
    __shared__ double wmx[LONG ? 3 * QRX_LONG_MAXCH : 1];
    __shared__ double red[64];
    const int p = p0 + blockIdx.x;
    if (st && st[p].stage != ST_NEED_QR) return;
    const int tid = threadIdx.x, BS = blockDim.x, ldp = n + 1;
#ifdef QRX_DBG_CLK      // -DQRX_DBG_CLK: in-kernel phase clocks (100 MHz), printed for step 100 -- where DESIGN.md's phase figures come from
    long long clk[6]; clk[0] = wall_clock64();
#endif
    int *redi = reinterpret_cast<int *>(red + 32);
    double *rdiag = w.rdiag + (size_t)p * n, *wa = w.wa + (size_t)p * n;
    int32_t *src = w.src + (size_t)p * ldp;
    int32_t *ipvt = v.ipvt + (size_t)p * n;
    double *tpc = w.tp + ((size_t)p * 2 + cur) * QRX_C * ldp;
    double *R = Rall + (size_t)p * n * n;

    // Everything the step needs that does not depend on the pivot is fetched up front, alongside the norms the pivot
    // search looks at: with at most BS candidate columns each thread also brings its candidate's source column, output
    // index and pending multipliers, and the winner publishes them through LDS -- the gather below then starts one
    // memory latency after the kernel does, instead of four dependent ones (norms -> src[kmax] -> ... ).
    const bool onecand = (n - j <= BS);
    int32_t *slotof = w.slotof + (size_t)p * ld;
    // The pass before this step flushed: every live slot k >= j now sits at its own position coff + k, everything left
    // of that is consumed.  (The flushing pass itself must not write the maps, see qrx_pass_tail.)
    const bool fresh = (flush & 2) != 0;
    if (fresh) {
        for (int c = tid; c < ld; c += BS) slotof[c] = (c - coff >= j && c - coff <= n) ? c - coff : -1;
        for (int k = j + tid; k <= n; k += BS) src[k] = coff + k;
    }
    double bv = 0.0, my_tk[QRX_C - 1];
    int bk = 0x7fffffff, my_src = 0, my_ipvt = 0;
    if (onecand) {
        const int k = j + tid;
        if (k < n) {
            bv = rdiag[k]; bk = k;
            my_src = fresh ? coff + k : src[k]; my_ipvt = ipvt[k];
#pragma unroll
            for (int q = 0; q < QRX_C - 1; ++q) my_tk[q] = (q < np) ? tpc[(size_t)q * ldp + k] : 0.0;
        }
    } else {
        for (int k = j + tid; k < n; k += BS) {
            const double d = rdiag[k];
            if (bk == 0x7fffffff || d > bv) { bv = d; bk = k; }
        }
    }
    // slot j's own entries (tid 0 does the interchange bookkeeping)
    double rd_j = 0.0, wa_j = 0.0;
    int src_j = 0, ipvt_j = 0, src_0 = 0;
    if (tid == 0) { rd_j = rdiag[j]; wa_j = wa[j]; src_j = fresh ? coff + j : src[j]; ipvt_j = ipvt[j]; src_0 = src[0]; }
    __shared__ uint32_t amx[3 * 16];
    // (the DPP search in the m <= 2048 instances only: 2048x128 alone 13.2 -> 12.85 ms per solve; in the 4096-row instances
    // it measured 0.8 % SLOWER per solve -- the compiler's schedule of what follows, not the search itself)
    const int kmax = (QRX_NL == 32 && onecand) ? j + block_argmax_first_norms(bv, bk != 0x7fffffff, amx) : block_argmax_first(bv, bk, red, redi);
    double *pub = red + 40;                                      // [0 .. QRX_C-2] multipliers, then src, ipvt (as ints)
    int *pubi = reinterpret_cast<int *>(pub + QRX_C);
    if (onecand) {
        if (j + tid == kmax) {
#pragma unroll
            for (int q = 0; q < QRX_C - 1; ++q) pub[q] = my_tk[q];
            pubi[0] = my_src; pubi[1] = my_ipvt;
        }
    } else if (tid == 0) {
        pubi[0] = fresh ? coff + kmax : src[kmax]; pubi[1] = ipvt[kmax];
    } else if (tid <= np) {
        pub[tid - 1] = tpc[(size_t)(tid - 1) * ldp + kmax];
    }
    __syncthreads();
    const int srck = pubi[0];
#ifdef QRX_DBG_CLK
    clk[1] = wall_clock64();
#endif
    double tk[QRX_C];
#pragma unroll
    for (int q = 0; q < QRX_C; ++q) tk[q] = (q < QRX_C - 1 && q < np) ? pub[q] : 0.0;
    if (kmax != j) {
        if (tid == 0) {
            rdiag[kmax] = rd_j;
            wa[kmax] = wa_j;
            ipvt[j] = pubi[1]; ipvt[kmax] = ipvt_j;
            if (j == 0) {
                // first step only: a physical interchange (slot 0's column is copied over the consumed pivot column in the
                // gather below), so that the live columns are coff + 1 .. from the start -- with n + 1 = 1 (mod 64) a
                // live column at coff + 0 would cost every pass of the first cycle a whole extra 64-column window
                slotof[src_0] = -1;
                slotof[srck] = kmax;
            } else {
                src[kmax] = src_j;                               // slot kmax now lives where slot j's data is
                slotof[src_j] = kmax;
                slotof[srck] = -1;                               // the pivot column is consumed
            }
        }
        if (tid < np) tpc[(size_t)tid * ldp + kmax] = tpc[(size_t)tid * ldp + j];
        for (int i = tid; i < j; i += BS) {                      // rows of R already final
            const double t = R[(size_t)j * n + i];
            R[(size_t)j * n + i] = R[(size_t)kmax * n + i];
            R[(size_t)kmax * n + i] = t;
        }
    } else if (tid == 0) {
        slotof[srck] = -1;                                       // the pivot column is consumed
    }
    const bool move0 = (j == 0 && kmax != 0);                   // see above: slot 0's column takes the pivot column's place
    if (LONG && (flush & 8)) {
        // split long-column step, part 1 of 3: the search and the bookkeeping only; the gather (k_qrx_gather_long, the
        // whole chip) and NORM2 (k_qrx_norm_long) follow as launches of their own
        if (tid == 0) {
            QrxStep s;
            s.ajnorm = 0.0; s.ajj = 0.0; s.kmax = kmax; s.srck = srck; s.t0 = np ? tk[0] : 0.0;
            w.step[p] = s;
        }
        return;
    }
    const int src0 = coff;                                       // physical column of slot 0 at step 0
    // The pivot column with its pending updates applied, oldest first.  Consecutive threads take consecutive rows (eight
    // of them share a sector of the row-blocked matrix); four rows per thread are loaded together before any is stored.
    const double *__restrict__ Vc = w.V + ((size_t)p * 2 + cur) * QRX_C * vst;          // slot q at Vc + q * vst
    double *__restrict__ Vn = (flush & 1) ? w.V + ((size_t)p * 2 + (cur ^ 1)) * QRX_C * vst
                                    : w.V + (((size_t)p * 2 + cur) * QRX_C + np) * vst;
    double *Tp = T + (size_t)p * tst;
    // The gathered column also goes to LDS (the region NORM2 later fills with its coefficients: it reads every element
    // before it writes any) when it fits one NORM2 chunk, so that the norm does not start with another trip to memory.
    const bool staged = !LONG && (m - j <= 64 * QRX_NL);
    double *stage = cd;
    // (handful instances: a word of padding per NORM2 thread's KEEP consecutive elements -- read back at a stride of KEEP
    // doubles, every lane of a wave hit the same two banks: 1.2 us of a 4096-row step)
    auto spos = [](int i) __attribute__((always_inline)) { return FEW ? i + i / (64 * QRX_NL / 256) : i; };
    // ... and in that case every thread KEEPS its rows (at most 64 * QRX_NL / 256 of them) in registers for the scaling
    // below: the raw column then never goes to memory -- one write of the reflector instead of write, read, write.
    constexpr int KEEP = 64 * QRX_NL / 256;                      // rows per thread when staged (16 or 8), four per iteration
    double keep[KEEP];
#pragma unroll
    for (int u = 0; u < KEEP; ++u) keep[u] = 0.0;
    auto gather4 = [&](int i0, double (&e)[4]) __attribute__((always_inline)) {
        double vq[4][QRX_C - 1], mv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = min(i0 + u * BS, m - 1);
            e[u] = Tp[qrx_at(row, srck, ld)];
            mv[u] = move0 ? Tp[qrx_at(row, src0, ld)] : 0.0;
#pragma unroll
            for (int q = 0; q < QRX_C - 1; ++q) vq[u][q] = (q < np) ? Vc[(size_t)q * vst + row] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int q = 0; q < QRX_C - 1; ++q)
                if (q < np) e[u] = e[u] - tk[q] * vq[u][q];
            if (move0 && i0 + u * BS < m) Tp[qrx_at(i0 + u * BS, srck, ld)] = mv[u];
        }
    };
    if (FEW && staged && np == 1 && !move0) {
        // a handful of problems (one update pending: k_qrx_pass_col's regime): all of the thread's rows and their reflector
        // entries in flight together -- four rows at a time were four dependent trips to memory (6 us of a 26 us step)
        double e[KEEP], pv[KEEP];
#pragma unroll
        for (int u = 0; u < KEEP; ++u) {
            const int row = min(j + tid + u * BS, m - 1);
            e[u] = Tp[qrx_at(row, srck, ld)];
            pv[u] = Vc[row];
        }
#pragma unroll
        for (int u = 0; u < KEEP; ++u) { asm volatile("" : "+v"(e[u])); asm volatile("" : "+v"(pv[u])); }
#pragma unroll
        for (int u = 0; u < KEEP; ++u) {
            e[u] = e[u] - tk[0] * pv[u];
            keep[u] = e[u];
            if (j + tid + u * BS < m) stage[spos(tid + u * BS)] = e[u];
        }
    } else if (staged) {
#pragma unroll
        for (int it = 0; it < KEEP / 4; ++it) {
            const int i0 = j + tid + it * 4 * BS;
            if (i0 < m) {                                        // (not uniform: the last iteration is ragged)
                double e[4];
                gather4(i0, e);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    keep[it * 4 + u] = e[u];
                    if (i0 + u * BS < m) stage[spos(i0 + u * BS - j)] = e[u];
                }
            }
        }
    } else if (LONG && np <= 1 && !move0) {
        // a long column with at most one update pending (k_qrx_pass_col's regime): a lane takes a PAIR of rows (16
        // bytes), four adjacent lanes a sector of the row-blocked matrix -- a load instruction of the wave covers 16 whole
        // sectors and 1 KB of the pending reflector, eight pairs per thread in flight.  (A sector per lane had every load
        // instruction touch 64 lines for a quarter of each: 62 us for 65536 rows; the row-at-a-time loop below: 67 us.)
        const int jb = j & ~7, npair = (m - jb + 1) >> 1;
        const double *colp = Tp + qrx_at(jb, srck, ld);
        const size_t blk = (size_t)ld * 8;
        const double t0 = np ? tk[0] : 0.0;
        constexpr int GU = QRX_LONG_GU;
        for (int pb = tid; pb < npair; pb += GU * BS) {
            double2 ax[GU], px[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int pr = min(pb + u * BS, npair - 1);             // pair pr: rel rows 2 pr, 2 pr + 1 of sector pr >> 2
                ax[u] = *reinterpret_cast<const double2 *>(colp + (size_t)(pr >> 2) * blk + 2 * (pr & 3));
                px[u] = np ? *reinterpret_cast<const double2 *>(Vc + jb + 2 * pr) : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int pr = pb + u * BS, row = jb + 2 * pr;
                if (pr >= npair) continue;
                if (np) { ax[u].x = ax[u].x - t0 * px[u].x; ax[u].y = ax[u].y - t0 * px[u].y; }
                if (row >= j && row + 1 < m) *reinterpret_cast<double2 *>(Vn + row) = ax[u];
                else {
                    if (row >= j && row < m) Vn[row] = ax[u].x;
                    if (row + 1 >= j && row + 1 < m) Vn[row + 1] = ax[u].y;
                }
            }
        }
    } else {
        for (int i0 = j + tid; i0 < m; i0 += 4 * BS) {
            double e[4];
            gather4(i0, e);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * BS < m) Vn[i0 + u * BS] = e[u];
        }
    }
    __syncthreads();
    const double ejj = staged ? stage[0] : Vn[j];                 // (spos(0) = 0)
#ifdef QRX_DBG_CLK
    clk[2] = wall_clock64();
#endif
                 // the diagonal entry before scaling (read before NORM2 reuses the region)
    double ajnorm;                                                // :642
    if constexpr (LONG) ajnorm = norm2_flang_block_lanes_pipe<QRX_LONG_EL, 192>([&](int i) { return Vn[j + i]; }, m - j, cd, aux, wmx);
    else ajnorm = staged ? norm2_flang_block_lanes<QRX_NL, 256, FEW>([&](int i) { return stage[spos(i)]; }, m - j, cd, aux)
                         : norm2_flang_block_lanes<QRX_NL, 256, FEW>([&](int i) { return Vn[j + i]; }, m - j, cd, aux);
#ifdef QRX_DBG_CLK
    clk[3] = wall_clock64();
#endif
    double ajj = 0.0;
    if (staged) {
        // :645-646 from the registers (ajnorm == 0: the column is zero; the reflector slot still gets the column)
        if (ajnorm != 0.0 && ejj < 0.0) ajnorm = -ajnorm;         // :644
        // the divisions of all the thread's rows first, unconditionally (x / 1.0 == x stands for "no reflector"): under
        // the row and zero-norm conditions they were one division sequence after the other, 1.6 us of a 4096-row step
        const double dn = (ajnorm != 0.0) ? ajnorm : 1.0;
#pragma unroll
        for (int u = 0; u < KEEP; ++u) keep[u] = keep[u] / dn;
#pragma unroll
        for (int u = 0; u < KEEP; ++u) {
            const int i = j + tid + u * BS;
            double t = keep[u];
            if (ajnorm != 0.0 && i == j) { t = t + 1.0; ajj = t; }
            if (i < m) Vn[i] = t;
        }
    } else if (LONG && (flush & 4) && ajnorm != 0.0) {
        // the scaling of the rows below j is left to k_qrx_scale_long, which spreads it over the chip (on this one
        // workgroup the 65536 divisions of a long column take 24 us); row j and the step record are settled here
        if (ejj < 0.0) ajnorm = -ajnorm;                         // :644
        if (tid == 0) { ajj = ejj / ajnorm + 1.0; Vn[j] = ajj; } // :645-646 at i = j
    } else if (LONG && ajnorm != 0.0) {
        if (ejj < 0.0) ajnorm = -ajnorm;                         // :644
        // :645-646 a sector at a time, four per thread in flight (thread 0 holds row j)
        const int jb = j & ~7, nsec = (m - jb + 7) >> 3;
        for (int sb = tid; sb < nsec; sb += QRX_LONG_SU * BS) {
            double2 tx[QRX_LONG_SU][4];
#pragma unroll
            for (int q = 0; q < QRX_LONG_SU; ++q) {
                const double2 *sp = reinterpret_cast<const double2 *>(Vn + jb + (size_t)min(sb + q * BS, nsec - 1) * 8);
#pragma unroll
                for (int h = 0; h < 4; ++h) tx[q][h] = sp[h];
            }
#pragma unroll
            for (int q = 0; q < QRX_LONG_SU; ++q) {
                const int sc = sb + q * BS, row = jb + sc * 8;
                if (sc >= nsec) continue;
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    tx[q][h].x = tx[q][h].x / ajnorm;
                    tx[q][h].y = tx[q][h].y / ajnorm;
                    if (row + 2 * h == j) { tx[q][h].x = tx[q][h].x + 1.0; ajj = tx[q][h].x; }
                    if (row + 2 * h + 1 == j) { tx[q][h].y = tx[q][h].y + 1.0; ajj = tx[q][h].y; }
                }
                if (row >= j && row + 8 <= m) {
                    double2 *dp = reinterpret_cast<double2 *>(Vn + row);
#pragma unroll
                    for (int h = 0; h < 4; ++h) dp[h] = tx[q][h];
                } else {
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        if (row + 2 * h >= j && row + 2 * h < m) Vn[row + 2 * h] = tx[q][h].x;
                        if (row + 2 * h + 1 >= j && row + 2 * h + 1 < m) Vn[row + 2 * h + 1] = tx[q][h].y;
                    }
                }
            }
        }
    } else if (ajnorm != 0.0) {
        if (ejj < 0.0) ajnorm = -ajnorm;                         // :644
        // :645-646; eight rows per thread are loaded together before any is stored: the compiler must assume that the
        // store of one row aliases the load of the next, and a row-at-a-time loop pays a memory latency per row
        for (int i0 = j + tid; i0 < m; i0 += 8 * BS) {
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = Vn[min(i0 + u * BS, m - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                t[u] = t[u] / ajnorm;
                if (i0 + u * BS == j) { t[u] = t[u] + 1.0; ajj = t[u]; }
                if (i0 + u * BS < m) Vn[i0 + u * BS] = t[u];
            }
        }
    }
#ifdef QRX_DBG_CLK
    clk[4] = wall_clock64();
    if (tid == 0 && j == 100) printf("pivot j=100: head %lld gather %lld norm %lld scale %lld (x10 ns)\n", clk[1]-clk[0], clk[2]-clk[1], clk[3]-clk[2], clk[4]-clk[3]);
#endif
    if (tid == 0) {                                              // thread 0 scaled row j
        QrxStep s;
        s.ajnorm = ajnorm; s.ajj = ajj; s.kmax = kmax; s.srck = srck; s.t0 = 0.0;
        w.step[p] = s;
        rdiag[j] = -ajnorm;                                      // :665
    }
}

// Split long-column step, part 2 of 3: the pivot column with its (at most one) pending update applied, for the whole
// chip -- on the pivot kernel's one workgroup the 1.5 MB of a 65536-row gather took 29 us.  A lane takes a pair of rows.
__global__ void __launch_bounds__(256)
k_qrx_gather_long(int m, int ld, size_t tst, size_t vst, int j, int cur, int np, int flush, const double *__restrict__ T, QrxWs w,
                  const LmState *__restrict__ st)
{
    const int p = blockIdx.y;
    if (st && st[p].stage != ST_NEED_QR) return;
    const QrxStep step = w.step[p];
    const double *Tp = T + (size_t)p * tst;
    const double *__restrict__ Vc = w.V + ((size_t)p * 2 + cur) * QRX_C * vst;
    double *__restrict__ Vn = (flush & 1) ? w.V + ((size_t)p * 2 + (cur ^ 1)) * QRX_C * vst
                                    : w.V + (((size_t)p * 2 + cur) * QRX_C + np) * vst;
    const int jb = j & ~7, npair = (m - jb + 1) >> 1;
    const double *colp = Tp + qrx_at(jb, step.srck, ld);
    const size_t blk = (size_t)ld * 8;
    const double t0 = step.t0;
    const int pb = (blockIdx.x * 256 + threadIdx.x) * 4;
    double2 ax[4], px[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int pr = min(pb + u, npair - 1);                   // pair pr: rel rows 2 pr, 2 pr + 1 of sector pr >> 2
        ax[u] = *reinterpret_cast<const double2 *>(colp + (size_t)(pr >> 2) * blk + 2 * (pr & 3));
        px[u] = np ? *reinterpret_cast<const double2 *>(Vc + jb + 2 * pr) : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int pr = pb + u, row = jb + 2 * pr;
        if (pr >= npair) continue;
        if (np) { ax[u].x = ax[u].x - t0 * px[u].x; ax[u].y = ax[u].y - t0 * px[u].y; }
        if (row >= j && row + 1 < m) *reinterpret_cast<double2 *>(Vn + row) = ax[u];
        else {
            if (row >= j && row < m) Vn[row] = ax[u].x;
            if (row + 1 >= j && row + 1 < m) Vn[row + 1] = ax[u].y;
        }
    }
}

// Split long-column step, part 3 of 3: NORM2 of the gathered column (pipelined: three preparing waves, a chain wave),
// the sign, row j of the reflector and the step record; the rows below j are scaled by k_qrx_scale_long.
// (Round 4: since the running sum of most chunks is formed without the serial chain -- ordered_possum_wave_int -- the
// kernel is bound by its PREPARING waves, divisions and prefix maxima: QRX_NORM_PREP / 64 of them instead of three.)
#ifndef QRX_NORM_PREP
#define QRX_NORM_PREP 384
#endif
#define QRX_NORM_THREADS (QRX_NORM_PREP + 64)
__global__ void __launch_bounds__(QRX_NORM_THREADS)
k_qrx_norm_long(int m, int n, size_t vst, int j, int cur, int np, int flush, QrxWs w, const LmState *__restrict__ st)
{
    __shared__ __attribute__((aligned(16))) double cd[2 * (64 * QRX_LONG_EL + 128)];
    __shared__ __attribute__((aligned(16))) double aux[8 + QRX_NORM_PREP];
    __shared__ double wmx[(QRX_NORM_PREP / 64) * QRX_LONG_MAXCH];
    const int p = blockIdx.x;
    if (st && st[p].stage != ST_NEED_QR) return;
    double *__restrict__ Vn = (flush & 1) ? w.V + ((size_t)p * 2 + (cur ^ 1)) * QRX_C * vst
                                    : w.V + (((size_t)p * 2 + cur) * QRX_C + np) * vst;
    const double ejj = Vn[j];                                     // the diagonal entry before scaling
    double ajnorm = norm2_flang_block_lanes_pipe<QRX_LONG_EL, QRX_NORM_PREP>([&](int i) { return Vn[j + i]; }, m - j, cd, aux, wmx);   // :642
    if (threadIdx.x == 0) {
        QrxStep s = w.step[p];
        double ajj = 0.0;
        if (ajnorm != 0.0) {
            if (ejj < 0.0) ajnorm = -ajnorm;                      // :644
            ajj = ejj / ajnorm + 1.0;                             // :645-646 at i = j
            Vn[j] = ajj;
        }
        s.ajnorm = ajnorm; s.ajj = ajj;
        w.step[p] = s;
        w.rdiag[(size_t)p * n + j] = -ajnorm;                     // :665
    }
}

// :645-646 for the rows below j of a long pivot column (k_qrx_pivot with flush bit 2 left them unscaled): a launch of its
// own so that every CU takes a share of the divisions.  Vn as in k_qrx_pivot; each thread a pair of rows per load, four in flight.
__global__ void __launch_bounds__(256)
k_qrx_scale_long(int m, size_t vst, int j, int cur, int np, int flush, QrxWs w, const LmState *__restrict__ st)
{
    const int p = blockIdx.y;
    if (st && st[p].stage != ST_NEED_QR) return;
    const double ajnorm = w.step[p].ajnorm;
    if (ajnorm == 0.0) return;                                   // (the column is zero: nothing to scale)
    double *__restrict__ Vn = (flush & 1) ? w.V + ((size_t)p * 2 + (cur ^ 1)) * QRX_C * vst
                                    : w.V + (((size_t)p * 2 + cur) * QRX_C + np) * vst;
    const int first = (j + 2) & ~1;                              // first even row above j (row j + 1 alone when j is even)
    if (blockIdx.x == 0 && threadIdx.x == 0 && first == j + 2 && j + 1 < m) Vn[j + 1] = Vn[j + 1] / ajnorm;
    const int npair = (m - first) >> 1;
    const int pb = (blockIdx.x * 256 + threadIdx.x) * 4;
    double2 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = (pb + u < npair) ? *reinterpret_cast<const double2 *>(Vn + first + 2 * (pb + u)) : make_double2(0.0, 0.0);
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (pb + u < npair) {
            t[u].x = t[u].x / ajnorm; t[u].y = t[u].y / ajnorm;
            *reinterpret_cast<double2 *>(Vn + first + 2 * (pb + u)) = t[u];
        }
    if (blockIdx.x == 0 && threadIdx.x == 0 && ((m - first) & 1)) Vn[m - 1] = Vn[m - 1] / ajnorm;     // odd last row
}

// The end of a trailing column's pass, given its dot product s with the reflector: multiplier (:654), row j becomes
// final (R(j,k) / qtf(j), :655 at i = j), norm down-date (:656-661) with the rare recomputation.
template <int NP, bool FLUSH>
__device__ __forceinline__ void
qrx_pass_tail(int p, int j, int k, int col, int wcol, int m, int n, int ld, int cur, size_t vst, double s, double rowj, double rk, double wak,
              bool refl, double ajj, const double (&tq)[NP > 0 ? NP : 1], const double *__restrict__ Tp, const double *__restrict__ vc,
              const double *__restrict__ vo, double *__restrict__ tpall,
              double *__restrict__ rdall, double *__restrict__ waall, double *__restrict__ Rall, double *__restrict__ qtfall)
{
    constexpr int NPI = NP < QRX_C ? NP : 0;
    const int ldp = n + 1, jb = j & ~7;
    double *tpc = tpall + ((size_t)p * 2 + cur) * QRX_C * ldp;
    auto pending = [&](double a, const auto &vr) {
        double e = a;
#pragma unroll
        for (int q = 0; q < NP; ++q) e = e - tq[q] * vr[q];
        return e;
    };
    double *tpn = FLUSH ? tpall + ((size_t)p * 2 + (cur ^ 1)) * QRX_C * ldp : tpc + (size_t)NPI * ldp;
    double *rdiag = rdall + (size_t)p * n, *wa = waall + (size_t)p * n;
    const double temp = refl ? s / ajj : 0.0;                           // :654 (residual: w + v*(-s/a) == w - (s/a)*v bit for bit)
    tpn[k] = temp;
    // (A flush has put the column at its slot's own position coff + k.  The slot maps are NOT touched here: another
    // workgroup of this very launch may not have read its lanes' entries yet -- with more workgroups than the chip holds
    // at once, or rows so few that a workgroup ends before a later one starts, it would take the moved column for a live
    // one still waiting for its pending updates.  The next step's pivot kernel resets the maps instead.)
    const double rjk = refl ? rowj - temp * ajj : rowj;                 // :655 at i = j: row j is final
    if (k == n) { qtfall[(size_t)p * n + j] = rjk; return; }
    Rall[(size_t)p * n * n + (size_t)k * n + j] = rjk;
    if (!refl) return;
    // rk = rdiag(k), wak = wa(k): fetched by the caller at the start of the pass (only this column's owner ever writes
    // them), so that the tail is not two dependent trips to memory
    if (rk != 0.0) {                                                    // :656-661
        const double t2 = rjk / rk;
        rk = rk * sqrt(fmax(0.0, 1.0 - t2 * t2));
        const double q = rk / wak;
        if (!(5.0e-2 * (q * q) > NLH_EPS)) {
            rk = norm2_flang_serial([&](int i2) {
                const int row = j + 1 + i2, rel = row - jb;
                double vr[NP + 1];
#pragma unroll
                for (int q2 = 0; q2 < NP; ++q2) vr[q2] = vc[(size_t)q2 * vst + rel];
                // a flush has just rewritten the column (pending updates applied) at physical column wcol
                const double e = FLUSH ? Tp[qrx_at(row, wcol, ld)] : pending(Tp[qrx_at(row, col, ld)], vr);
                const double vn = FLUSH ? vo[rel] : vc[(size_t)NPI * vst + rel];
                return e - temp * vn;
            }, m - j - 1);
            wa[k] = rk;
        }
        rdiag[k] = rk;
    }
}

// Step j, part 2: every trailing column k = j+1 .. n (n = the residual): pending updates, dot product with
// the reflector in ascending row order (:652-653), multiplier (:654), row j becomes final (R(j,k) / qtf(j)),
// norm down-date (:656-661).
// One wave per workgroup; a lane owns one physical column and walks down the rows.  The matrix is row-blocked (qrx_at):
// a lane reads two rows of its column per 16-byte load, a wave 4 KB contiguous per 8-row block, 32 loads (64 rows) in
// flight ahead of the arithmetic.  The reflector entries of a row (wave-uniform) come from an LDS tile of 64 rows that
// the wave stages for itself one tile ahead (coalesced slot vectors in, broadcast ds_reads out).
// SHARE: the windows of a problem are the WAVES OF ONE WORKGROUP (64 nwin threads), kept together by one LDS-only barrier
// per 64-row tile, so that they ask for a reflector tile -- (NP + 1) / 64 of a window's matrix bytes -- at the same
// time and the fabric delivers it once per problem, not once per window (as separate workgroups the sibling waves
// drift apart by more than the few microseconds a line survives in an L2 that streams 750 GB/s: the counters showed
// 1.18x the algorithmic bytes read).
// Built on this form, measured and dropped (all bit-identical; 2048 x 4096x256, ms per factorisation of the batch against
// 510 for pivot and pass launches alternating): with a problem's pass in ONE workgroup, its pivot steps need no other
// workgroup either, so they can run inside the pass kernels -- (1) one persistent workgroup per problem for all n steps:
// 581 (every pass variant and the pivot step in one kernel: 256 registers with spills, two workgroups per CU; 627 at one
// per CU without spills); (2) step j + 1's pivot step as the TAIL of step j's pass kernel: 523, the launches longer by the
// whole pivot time -- workgroups that share HBM evenly finish their passes together, so their pivot steps still leave
// HBM idle together; (3) the problems in two classes half a step apart (pivot step at the head of the launch for the odd
// ones, at the tail for the even ones): 532; (4) the same with the 2048-element NORM2 chunk (35 KB of LDS): the merged
// kernels need 184-256 registers, two waves per SIMD.  The reason is the same each time: at two waves per SIMD the
// pass streams at the HBM rate only with EVERY resident wave streaming (forcing today's pass kernels to that occupancy
// costs nothing: 508), so a wave slot that spends 10-15 % of its time in a pivot step is 10-15 % of the bandwidth gone;
// hiding the pivot step inside the launch needs spare wave slots, i.e. pass and pivot step within ~168 registers.
#define QRX_SHARE_MAXWIN 4
template <int NP, bool FLUSH, bool SHARE>
__global__ void __launch_bounds__(SHARE ? 64 * QRX_SHARE_MAXWIN : 64) __attribute__((amdgpu_waves_per_eu(1, (FLUSH && NP >= 4) ? 2 : 4)))
k_qrx_pass(int p0, int nprob, int nwin, int lo, int m, int n, int ld, int coff, size_t tst, size_t vst, int j, int cur,
           double *__restrict__ T, const double *__restrict__ Vall, double *__restrict__ tpall,
           int32_t *__restrict__ srcall, int32_t *__restrict__ slotall, double *__restrict__ rdall,
           double *__restrict__ waall, const QrxStep *__restrict__ stepall, double *__restrict__ Rall,
           double *__restrict__ qtfall, const LmState *__restrict__ st)
{
    constexpr int U = QRX_UF(FLUSH);       // rows per load group (8 or 16 loads per lane; the flush is write-bound and
                                           // register-hungry: half the read-ahead)
    constexpr int TR = QRX_TR;             // rows per reflector tile
    constexpr int LP = QRX_C;              // LDS row: the pending entries and the new one (NP + 1 <= QRX_C doubles)
    constexpr int NPI = NP < QRX_C ? NP : 0;
    __shared__ double vt[2][TR * LP];
    // Workgroup -> (problem, window): consecutive workgroup ids go to consecutive XCDs, so the windows of one problem
    // are given ids that agree modulo 8: they share an L2 (reflector tiles, multipliers are fetched from the fabric once).
    const int b_ = blockIdx.x, grp = b_ / (8 * nwin), r_ = b_ % (8 * nwin);
    const int pl = SHARE ? b_ : grp * 8 + (r_ & 7);
    const int win = SHARE ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : r_ >> 3;
    if (pl >= nprob) return;
    const int p = p0 + pl;
    if (st && st[p].stage != ST_NEED_QR) return;
    const int lane = threadIdx.x & 63, ldp = n + 1;
    const QrxStep step = stepall[p];
    const bool refl = step.ajnorm != 0.0;
    const double ajj = step.ajj;
    int32_t *slotp = slotall + (size_t)p * ld;
    double *tpc = tpall + ((size_t)p * 2 + cur) * QRX_C * ldp;
    double *Tp = T + (size_t)p * tst;
    // tile barrier: with one wave per workgroup the compiler's __syncthreads (no s_barrier, counted waits) is what the
    // loops below were tuned with; shared tiles need a real barrier, ordering LDS traffic only (never the matrix loads)
    auto tile_barrier = [&]() __attribute__((always_inline)) {
        if (SHARE) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else __syncthreads();
    };
    // rows are counted from the start of row j's 8-row block: rel row r = absolute row jb + r, first live one r0
    const int jb = j & ~7, r0 = j & 7, mrel = m - jb;
    const double *vc = Vall + ((size_t)p * 2 + cur) * QRX_C * vst + jb;          // slot q, rel row r at vc[q * vst + r]
    const double *vo = Vall + ((size_t)p * 2 + (cur ^ 1)) * QRX_C * vst + jb;   // slot 0 of the other bank

    // A lane owns a PHYSICAL column; which slot of the permuted matrix that column currently holds comes from the
    // inverse map (the interchange never moves data).  Since the last flush the live columns are coff + lo .. ld - 1
    // (lo = the step after that flush, 1 before the first one) minus the consumed ones, at most QRX_C - 1 of them,
    // whose lanes idle: every load is the lane's own column.
    const int col = ld - 64 * (win + 1) + lane;
    const int kslot = (col >= coff + lo) ? slotp[col] : -1;
    const bool act = kslot > j;
    const int k = act ? kslot : n;                                      // idle lanes: any valid slot for the table reads
    double tq[NP > 0 ? NP : 1];
#pragma unroll
    for (int q = 0; q < NP; ++q) tq[q] = tpc[(size_t)q * ldp + k];
    auto pending = [&](double a, const auto &vr) __attribute__((always_inline)) {
        double e = a;
#pragma unroll
        for (int q = 0; q < NP; ++q) e = e - tq[q] * vr[q];
        return e;
    };
    const double rk0 = (act && k < n) ? rdall[(size_t)p * n + k] : 0.0, wa0 = (act && k < n) ? waall[(size_t)p * n + k] : 1.0;
    double rowj;
    {   // row j with its pending updates (becomes final below)
        double v0[NP + 1];
#pragma unroll
        for (int q = 0; q < NP; ++q) v0[q] = vc[(size_t)q * vst + r0];
        rowj = pending(Tp[qrx_at(j, col, ld)], v0);
    }

    // reflector tile t: lane l fetches the entries of rel row t*TR + l (one coalesced 512-byte read per slot); the staged
    // LDS row is [pending v_0 .. v_NP-1, new v]
    // (SHARE: every wave still fetches and stores the whole tile -- the same values to the same LDS words, a benign
    // duplication: what the shared workgroup buys is that its waves ask for the same lines within a barrier interval of
    // each other, so the L2 serves three of the four requests.  Dealing the staging to the waves -- by slot or by tile --
    // was tried: any wave-dependent condition around these loads makes the compiler spill the row loops, 10x slower.)
    double sv[NP + 1];
    auto vfetch = [&](int t) __attribute__((always_inline)) {
        const int row = t * TR + lane;
#pragma unroll
        for (int q = 0; q < NP; ++q) sv[q] = vc[(size_t)q * vst + row];
        sv[NP] = FLUSH ? vo[row] : vc[(size_t)NPI * vst + row];
    };
    auto vstore = [&](int t) __attribute__((always_inline)) {           // tile t -> vt[t & 1]
        double *d = &vt[t & 1][lane * LP];
#pragma unroll
        for (int q = 0; q <= NP; ++q) d[q] = sv[q];
    };
    // Matrix accesses go through a buffer descriptor of this problem's matrix from row block jb on: address = descriptor
    // base + wave-uniform block offset (scalar register) + per-lane column offset (one VGPR for the whole pass) + the
    // row pair inside the block (immediate); reads past the last block return zero and are never used, idle lanes are
    // parked past the range and fetch nothing.
    const int nblk = (mrel + 7) >> 3;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(Tp + (size_t)(jb >> 3) * ld * 8, 0, (int)((size_t)nblk * ld * 64), 0x00020000);
    const unsigned so = act ? (unsigned)col * 64u : 0x80000000u;
    const unsigned ko = act ? (unsigned)(coff + k) * 64u : 0x80000000u;  // where a flush puts the column (idle lanes: nowhere)
    const unsigned ldb = (unsigned)ld * 64u;                            // bytes per row block
    double a0[U], a1[U];
    auto load = [&](double (&buf)[U], int rbase) __attribute__((always_inline)) {                      // rbase: rel row, a multiple of 8
#pragma unroll
        for (int b4 = 0; b4 < U / 8; ++b4) {
            const unsigned boff = (unsigned)((rbase >> 3) + b4) * ldb;
#pragma unroll
            for (int q2 = 0; q2 < 4; ++q2) {
                const qrx_u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, so + 16u * q2, boff, QRX_AUX_LOAD);
                buf[b4 * 8 + 2 * q2] = __hiloint2double((int)w.y, (int)w.x);
                buf[b4 * 8 + 2 * q2 + 1] = __hiloint2double((int)w.w, (int)w.z);
            }
        }
    };
    double s = 0.0;
    auto compute = [&](auto guarded, const double (&buf)[U], int rbase, const double *tile, int g) __attribute__((always_inline)) {
        constexpr bool GD = decltype(guarded)::value;
        double va[NP + 1], vb[NP + 1], est[8];
        auto ldsrow = [&](double (&dst)[NP + 1], int u) __attribute__((always_inline)) {
            const double *vr = tile + (g * U + u) * LP;
#pragma unroll
            for (int q = 0; q <= NP; ++q) dst[q] = vr[q];
        };
        // Two rows at a time with their update chains interleaved (each chain is a dependent mul / sub sequence, and a
        // wave alone on its SIMD has nobody else to hide that latency); the LDS reads of a row pair are issued one pair
        // ahead.  A pair (even row, odd row) is also the 16-byte unit of the flush.
        ldsrow(va, 0);
        ldsrow(vb, 1);
#pragma unroll
        for (int u = 0; u < U; u += 2) {
            const int row = rbase + u;
            if (GD && row >= mrel) break;                               // uniform
            double v0[NP + 1], v1[NP + 1];
#pragma unroll
            for (int q = 0; q <= NP; ++q) { v0[q] = va[q]; v1[q] = vb[q]; }
            if (u + 2 < U) { ldsrow(va, u + 2); ldsrow(vb, u + 3); }
            double e0 = buf[u], e1 = buf[u + 1];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const double p0_ = tq[q] * v0[q], p1_ = tq[q] * v1[q];
                e0 = e0 - p0_;
                e1 = e1 - p1_;
            }
            const double w0 = v0[NP] * e0, w1 = v1[NP] * e1;
            const bool ok0 = !GD || (row >= r0), ok1 = !GD || (row + 1 >= r0 && row + 1 < mrel);
            if (ok0) s = s + w0;                                        // :653, rows ascending
            if (ok1) s = s + w1;
            if (FLUSH) {
                const unsigned boff = (unsigned)(row >> 3) * ldb, ioff = (unsigned)(row & 7) * 8u;
                if (!GD) {
                    // a lane's eight rows of a block are one 64-byte sector: collect them and store the sector whole
                    // (four 16-byte stores back to back), not as four partial writes spread over the block's arithmetic
                    est[u & 7] = e0;
                    est[(u & 7) + 1] = e1;
                    if ((u & 7) == 6) {
#pragma unroll
                        for (int q2 = 0; q2 < 4; ++q2) {
                            qrx_u32x4 w;
                            w.x = (unsigned)__double2loint(est[2 * q2]); w.y = (unsigned)__double2hiint(est[2 * q2]);
                            w.z = (unsigned)__double2loint(est[2 * q2 + 1]); w.w = (unsigned)__double2hiint(est[2 * q2 + 1]);
                            __builtin_amdgcn_raw_buffer_store_b128(w, rsrc, ko + 16u * q2, boff, QRX_AUX_STORE);
                        }
                    }
                } else {                                                // first / last tile: row by row
                    qrx_u32x2 w;
                    w.x = (unsigned)__double2loint(e0); w.y = (unsigned)__double2hiint(e0);
                    if (ok0) __builtin_amdgcn_raw_buffer_store_b64(w, rsrc, ko + ioff, boff, QRX_AUX_STORE);
                    w.x = (unsigned)__double2loint(e1); w.y = (unsigned)__double2hiint(e1);
                    if (ok1) __builtin_amdgcn_raw_buffer_store_b64(w, rsrc, ko + ioff + 8u, boff, QRX_AUX_STORE);
                }
            }
            // keep a pair's work together: with many pending updates the scheduler otherwise hoists the LDS reads of
            // many row pairs (NP + 1 values each) and runs out of registers
            if (NP >= 6) __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int ntile = (mrel + TR - 1) / TR;
    // Unconditional read-ahead of the reflector rows: rows past the last one belong to the next slot or to the padding
    // behind the banks and are never used.
    vfetch(0);
    vstore(0);
    load(a0, 0);
    tile_barrier();
    std::false_type plain_t; std::true_type guard_t;
    // A tile whose rows are all live runs unguarded and fully unrolled; the tile of row j (unless j is a multiple of 8) and the last, partial one run guarded.  The three are separate loops, not branches of one loop body: a
    // merge of the two forms at the loop's back edge makes the compiler copy the registers of the load group just issued
    // for the next tile, i.e. wait for it -- one full memory latency per 64 rows.  For the same reason the next tile's
    // reflector rows go to LDS BEFORE the tile's last load group is issued (the memory counter retires in order).
    auto tile_plain = [&](int t) __attribute__((always_inline)) {
        const double *tile = vt[t & 1];
        const int rb = t * TR;
        vfetch(t + 1);
#pragma unroll
        for (int g = 0; g < TR / U; g += 2) {
            load(a1, rb + (g + 1) * U);
            compute(plain_t, a0, rb + g * U, tile, g);
            if (g + 2 >= TR / U) vstore(t + 1);
            load(a0, rb + (g + 2) * U);
            compute(plain_t, a1, rb + (g + 1) * U, tile, g + 1);
        }
        tile_barrier();
    };
    auto tile_guard = [&](int t) __attribute__((always_inline)) {
        const double *tile = vt[t & 1];
        const int rb = t * TR;
        vfetch(t + 1);
#pragma unroll 1
        for (int g = 0; g < TR / U; g += 2) {
            load(a1, rb + (g + 1) * U);
            compute(guard_t, a0, rb + g * U, tile, g);
            load(a0, rb + (g + 2) * U);
            compute(guard_t, a1, rb + (g + 1) * U, tile, g + 1);
        }
        vstore(t + 1);
        tile_barrier();
    };
    const int tfull = mrel / TR;                                        // tiles 0 .. tfull-1 end at or before the last row
    int t = 0;
    if (r0 != 0) { tile_guard(0); t = 1; }
#pragma unroll 1
    for (; t < tfull; ++t) tile_plain(t);
#pragma unroll 1
    for (; t < ntile; ++t) tile_guard(t);

    if (!act) return;
    qrx_pass_tail<NP, FLUSH>(p, j, k, col, coff + k, m, n, ld, cur, vst, s, rowj, rk0, wa0, refl, ajj, tq, Tp, vc, vo, tpall,
                             rdall, waall, Rall, qtfall);
}

// The same pass for launches too small to fill the chip (a handful of problems still iterating, one problem alone, the
// last narrow steps): there a lone wave per 64 columns is bound by its own instruction stream -- (2 NP + 2) fp64
// operations per row and lane, 80 us (NP = 0) to 225 us (flush) per 4096 rows -- while three SIMDs of its CU idle.
// ROW-PARALLEL form: a workgroup of W waves per (problem, window); waves 1 .. W-1 (producers) walk disjoint
// 16-row groups, apply the pending updates, multiply by the reflector entry and hand the PRODUCTS over through LDS;
// wave 0 (the adder) does nothing but the ordered sum s = s + w_i (:652-653), rows ascending -- the only part of the step
// that is serial by definition (a dependent fp64 add with register operands issues every ~2 ns: profiles/ubench/dp_issue.hip).  One barrier per round of (W-1) x 16 rows; the adder
// works one round behind the producers.  Same arithmetic on the same operands in the same order: bit-identical to
// k_qrx_pass.
#ifndef QRX_RP_MAX_WG
#define QRX_RP_MAX_WG 512               // launches of at most this many (problem, window) pairs take the row-parallel pass
                                        // (re-swept at the end of round 3, ms per solve at 0 / 512 / 1024: 256 x 4096x256 431 / 410 / 428,
                                        // 1024 x 2048x128 196 / 196 / 200, 1024 x 1024x64 47.6 / 47.2 / 50.5, 64 x 4096x256 289 / 192 / 192)
#endif
#define QRX_RP_G 16                     // rows per producer and round (two 8-row blocks)
#define QRX_RP_D 4                      // rounds per producer tile = row groups in flight per producer
// Workgroup barrier that orders LDS traffic only: __syncthreads() would also drain the matrix loads in flight for the
// coming rounds (s_waitcnt vmcnt(0)) -- a full memory latency per round.
__device__ __forceinline__ void qrx_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NP, bool FLUSH, int W>
__global__ void __launch_bounds__(64 * W)
k_qrx_pass_rp(int p0, int nprob, int nwin, int lo, int m, int n, int ld, int coff, size_t tst, size_t vst, int j, int cur,
              double *__restrict__ T, const double *__restrict__ Vall, double *__restrict__ tpall,
              int32_t *__restrict__ srcall, int32_t *__restrict__ slotall, double *__restrict__ rdall,
              double *__restrict__ waall, const QrxStep *__restrict__ stepall, double *__restrict__ Rall,
              double *__restrict__ qtfall, const LmState *__restrict__ st, const int32_t *__restrict__ plist)
{
    constexpr int NPR = W - 1, G = QRX_RP_G, D = QRX_RP_D, RR = NPR * G, LP = 8;
    static_assert(NP < 8, "the row-parallel pass stages at most eight entries per row");
    constexpr int NPI = NP < QRX_C ? NP : 0;
    __shared__ double vt[NPR][2][D * G * LP];
    __shared__ __attribute__((aligned(16))) double pb[2][RR / 2][64][2];
    const int b_ = blockIdx.x, grp = b_ / (8 * nwin), r_ = b_ % (8 * nwin);
    const int pl = grp * 8 + (r_ & 7), win = r_ >> 3;
    if (pl >= nprob) return;
    const int p = plist ? plist[pl] : p0 + pl;                           // (plist: nprob counts its entries, k_qrx_list)
    if (p < 0 || (st && st[p].stage != ST_NEED_QR)) return;
    const int lane = threadIdx.x & 63, ldp = n + 1;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: block offsets stay scalar
    const bool adder = (wv == 0);
    const int pw = adder ? 0 : wv - 1;                                   // producer index
    const QrxStep step = stepall[p];
    const bool refl = step.ajnorm != 0.0;
    const double ajj = step.ajj;
    int32_t *slotp = slotall + (size_t)p * ld;
    double *tpc = tpall + ((size_t)p * 2 + cur) * QRX_C * ldp;
    double *Tp = T + (size_t)p * tst;
    const int jb = j & ~7, r0 = j & 7, mrel = m - jb;
    const double *vc = Vall + ((size_t)p * 2 + cur) * QRX_C * vst + jb;
    const double *vo = Vall + ((size_t)p * 2 + (cur ^ 1)) * QRX_C * vst + jb;
    const int col = ld - 64 * (win + 1) + lane;
    const int kslot = (col >= coff + lo) ? slotp[col] : -1;
    const bool act = kslot > j;
    const int k = act ? kslot : n;
    double tq[NP > 0 ? NP : 1];
#pragma unroll
    for (int q = 0; q < NP; ++q) tq[q] = tpc[(size_t)q * ldp + k];
    const double rk0 = (adder && act && k < n) ? rdall[(size_t)p * n + k] : 0.0;
    const double wa0 = (adder && act && k < n) ? waall[(size_t)p * n + k] : 1.0;
    double rowj = 0.0;
    if (adder) {                                                        // row j with its pending updates (becomes final in the tail)
        double e = Tp[qrx_at(j, col, ld)];
#pragma unroll
        for (int q = 0; q < NP; ++q) e = e - tq[q] * vc[(size_t)q * vst + r0];
        rowj = e;
    }
    const int nblk = (mrel + 7) >> 3;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(Tp + (size_t)(jb >> 3) * ld * 8, 0, (int)((size_t)nblk * ld * 64), 0x00020000);
    const unsigned so = act ? (unsigned)col * 64u : 0x80000000u;
    const unsigned ko = act ? (unsigned)(coff + k) * 64u : 0x80000000u;
    const unsigned ldb = (unsigned)ld * 64u;
    const int nround = (mrel + RR - 1) / RR;                            // producers: rounds 0 .. nround-1, adder: 1 .. nround
    const int ntile = (nround + 1 + D - 1) / D;

    // producer side -------------------------------------------------------------------------------------------------
    // reflector tile kt of producer pw: the D row groups it handles in rounds kt*D .. kt*D + D-1; lane l fetches the
    // entries of group l / 16, row l % 16 (clamped to the last row: entries past it are never used)
    double sv[NP + 1];
    auto vfetch = [&](int kt) __attribute__((always_inline)) {
        const int row = min((kt * D + (lane >> 4)) * RR + pw * G + (lane & 15), mrel - 1);
#pragma unroll
        for (int q = 0; q < NP; ++q) sv[q] = vc[(size_t)q * vst + row];
        sv[NP] = FLUSH ? vo[row] : vc[(size_t)NPI * vst + row];
    };
    auto vstore = [&](int buf) __attribute__((always_inline)) {
        double *d = &vt[pw][buf][lane * LP];
#pragma unroll
        for (int q = 0; q <= NP; ++q) d[q] = sv[q];
    };
    double a[D][G];
    auto load = [&](double (&buf)[G], int t) __attribute__((always_inline)) {                          // the producer's group of round t
        const int rbase = t * RR + pw * G;
#pragma unroll
        for (int b4 = 0; b4 < G / 8; ++b4) {
            const unsigned boff = (unsigned)((rbase >> 3) + b4) * ldb;
#pragma unroll
            for (int q2 = 0; q2 < 4; ++q2) {
                const qrx_u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, so + 16u * q2, boff, QRX_AUX_LOAD);
                buf[b4 * 8 + 2 * q2] = __hiloint2double((int)w.y, (int)w.x);
                buf[b4 * 8 + 2 * q2 + 1] = __hiloint2double((int)w.w, (int)w.z);
            }
        }
    };
    auto produce = [&](const double (&buf)[G], int t, const double *tile, int i) __attribute__((always_inline)) {
        const int rbase = t * RR + pw * G;
        // (Measured and dropped.  Four rows at a time with the chains pinned in step and the LDS reads a group ahead: no
        // faster for a lone problem, 3 % slower for a full batch.  Reflector entries kept across the lanes and pulled out
        // with a DPP64 row broadcast instead of the LDS broadcast read: equal for a lone problem -- 7 extra VALU moves per
        // row replace 7 LDS reads --, 3 % slower for a full batch.  A product ring three rounds deep with LDS counters
        // instead of the barrier per round: no faster.  Cycle counters in the kernel: at NP = 6 a producer spends ~100 cycles
        // per row, i.e. its 13 fp64 operations + LDS reads at the VALU's 4 cycles per wave instruction; a fifth and sixth wave
        // share a SIMD with another and finish last.  The split over three producers is what this form can give.)
        double est[8];
#pragma unroll
        for (int u = 0; u < G; u += 2) {
            double v0[NP + 1], v1[NP + 1];
#pragma unroll
            for (int q = 0; q <= NP; ++q) { v0[q] = tile[(i * G + u) * LP + q]; v1[q] = tile[(i * G + u + 1) * LP + q]; }
            double e0 = buf[u], e1 = buf[u + 1];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const double p0_ = tq[q] * v0[q], p1_ = tq[q] * v1[q];
                e0 = e0 - p0_;
                e1 = e1 - p1_;
            }
            double2 ww;                                                 // rows outside the live range: garbage the adder skips
            ww.x = v0[NP] * e0;
            ww.y = v1[NP] * e1;
            *reinterpret_cast<double2 *>(&pb[t & 1][(pw * G + u) >> 1][lane][0]) = ww;
            if (FLUSH) {
                est[u & 7] = e0;
                est[(u & 7) + 1] = e1;
                if ((u & 7) == 6) {
                    const int rb8 = rbase + (u & ~7);
                    const unsigned boff = (unsigned)(rb8 >> 3) * ldb;
                    if (rb8 >= r0 && rb8 + 8 <= mrel) {                  // uniform: a whole 64-byte sector per lane
#pragma unroll
                        for (int q2 = 0; q2 < 4; ++q2) {
                            qrx_u32x4 w;
                            w.x = (unsigned)__double2loint(est[2 * q2]); w.y = (unsigned)__double2hiint(est[2 * q2]);
                            w.z = (unsigned)__double2loint(est[2 * q2 + 1]); w.w = (unsigned)__double2hiint(est[2 * q2 + 1]);
                            __builtin_amdgcn_raw_buffer_store_b128(w, rsrc, ko + 16u * q2, boff, QRX_AUX_STORE);
                        }
                    } else {                                            // the block of row j, the last block: row by row
#pragma unroll
                        for (int q2 = 0; q2 < 8; ++q2) {
                            qrx_u32x2 w;
                            w.x = (unsigned)__double2loint(est[q2]); w.y = (unsigned)__double2hiint(est[q2]);
                            if (rb8 + q2 >= r0 && rb8 + q2 < mrel)
                                __builtin_amdgcn_raw_buffer_store_b64(w, rsrc, ko + 8u * q2, boff, QRX_AUX_STORE);
                        }
                    }
                }
            }
        }
    };
    // adder side ----------------------------------------------------------------------------------------------------
    double s = 0.0;
    auto consume = [&](int half) __attribute__((always_inline)) {
        double2 ww[RR / 2];
#pragma unroll
        for (int pr = 0; pr < RR / 2; ++pr) ww[pr] = *reinterpret_cast<const double2 *>(&pb[half][pr][lane][0]);
#pragma unroll
        for (int pr = 0; pr < RR / 2; ++pr) {
            s = s + ww[pr].x;                                           // :653, rows ascending
            s = s + ww[pr].y;
        }
    };
    auto consume_edge = [&](int half, int rbase) __attribute__((always_inline)) {    // the round of row j, the last round
#pragma unroll 1
        for (int pr = 0; pr < RR / 2; ++pr) {
            const double2 ww = *reinterpret_cast<const double2 *>(&pb[half][pr][lane][0]);
            const int row = rbase + 2 * pr;
            if (row >= r0 && row < mrel) s = s + ww.x;                  // uniform
            if (row + 1 >= r0 && row + 1 < mrel) s = s + ww.y;
        }
    };

    // The two roles run separate loops with the same number of barriers (s_barrier counts arrivals, not program
    // counters): a shared loop body would merge the roles' register states at every round and make the compiler copy --
    // hence wait for -- the load groups in flight.  Producers run every round unguarded; the adder skips the rows outside the live range.
    if (adder) {
        qrx_lds_barrier();
#pragma unroll 1
        for (int t = 0; t < ntile * D; ++t) {
            if (t >= 1 && t <= nround) {
                const int rbase = (t - 1) * RR;
                if (rbase >= r0 && rbase + RR <= mrel) consume((t - 1) & 1);
                else consume_edge((t - 1) & 1, rbase);
            }
            qrx_lds_barrier();
        }
    } else {
        vfetch(0);
        vstore(0);
#pragma unroll
        for (int i = 0; i < D - 1; ++i) load(a[i], i);
        qrx_lds_barrier();
#pragma unroll 1
        for (int kt = 0; kt < ntile; ++kt) {
            vfetch(kt + 1);
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const int t = kt * D + i;
                load(a[(i + D - 1) % D], t + D - 1);
                produce(a[i], t, vt[pw][kt & 1], i);
                if (i == D - 1) vstore((kt + 1) & 1);
                qrx_lds_barrier();
            }
        }
    }
    if (!adder || !act) return;
    qrx_pass_tail<NP, FLUSH>(p, j, k, col, coff + k, m, n, ld, cur, vst, s, rowj, rk0, wa0, refl, ajj, tq, Tp, vc, vo, tpall,
                             rdall, waall, Rall, qtfall);
}

// The row-parallel pass, WIDE form: launches of at most QRX_RPW16_MAX_WG (problem, window) pairs -- the straggler rounds of
// a big batch, small per-rank shares of a sharded batch, the halves of a 32-255-problem batch.  There every workgroup has
// a CU to itself and k_qrx_pass_rp is bound by that CU: its producers read their reflector entries back as BROADCAST
// ds_reads ((NP + 1) / 2 sixteen-byte reads per row that each occupy the LDS pipe like a full 1 KB read) and a lone wave
// per SIMD pays every instruction's latency itself.  Here: sixteen waves per workgroup -- wave 0 the ADDER (the ordered sum
// of :652-653, one column per lane), twelve PRODUCERS (load, apply the pending updates, multiply with the reflector, hand
// the products over through LDS, store on a flush), one STAGER and two waves that retire at once: the adder's chain of
// dependent adds wants a SIMD to itself, so roles are dealt from the hardware's SIMD id at run time and only the stager,
// which sleeps at barriers, shares the adder's SIMD.  The adder's LDS reads are issued by hand three producers' rows
// ahead of its adds.  Same arithmetic on the same operands in the same order: bit-identical to k_qrx_pass.
// History of the form (round 4, measured with in-kernel clocks at 32 x 4096x256, us per pass of ~4000 rows): producers that
// own a COLUMN per lane (16 bytes at a 64-byte stride) with the reflector entries as scalar operands (s_load_dwordx16 per
// slot and round: no LDS traffic, no VALU work for them) ran a plain pass in 45 us and a flushing one in 100-118; entries
// through v_readlane out of a register tile had the producers at 26 us of produce time against 8; the compiler's own
// schedule of the adder ("twelve reads, wait for all, twenty-four adds") 30 us of consume against 26; a branch around the
// flush's stores made the compiler's wait counts assume the path without stores.  Round 5 replaced those producers by
// the lane-quad ones below (34 / 78 us) and raised the form's range from 128 to 256 pairs.
#ifndef QRX_RPW16_MAX_WG
#define QRX_RPW16_MAX_WG 256            // launches of at most this many (problem, window) pairs take the wide form (128 until
                                        // its producers read whole sectors per lane quad; ms per solve at 128 / 256 / 384 with them:
                                        // 44 x 4096x256 105.8 / 97.0 / 97.0, 64: 159 / 156 / 156, 128: 225 / 227 / 229, 256 x 2048x128 66.5 / 64.4 / 64.4)
#endif
#define QRX_RP_HALF 1632                // pass-form code: the wide form on 32-column half windows
#ifndef QRX_RPWH_MAX_WG
#define QRX_RPWH_MAX_WG 256             // ... while a launch has at most this many (problem, half window) pairs (ms per solve, off / 128 /
                                        // 256 / 512: 13 x 4096x256 68.9 / 62.8 / 62.8 / 62.8, 24: 71.1 / 67.5 / 66.1 / 66.1, 32: 78.4 / 75.6 / 71.9 / 71.9,
                                        // 44: 91.9 / 91.0 / 88.0 / 91.9, 32 x 2048x128 23.2 / 21.6 / 21.6 / 21.6, 128: 34.6 / 34.2 / 34.1 / 36.6).  The gain is
                                        // small because the adder, not LDS bandwidth, sets the pace: ~7 ns per row whatever the column count
                                        // (in-kernel clocks: consume 28.5 -> 26.5 us per 4000 rows with half the columns)
#endif
#ifndef QRX_PIV32_MAXM
#define QRX_PIV32_MAXM 2048
#endif
#ifndef QRX_FEW_MAX
#define QRX_FEW_MAX 256                 // batches of at most this many active problems take the pivot kernel's FEW instance
#endif                                  // (32 x 4096x256: 111 instead of 122 ms per solve)
#ifndef QRX_RPW_MAXNP
#define QRX_RPW_MAXNP 4                 // pending reflectors the wide form keeps: flush period MAXNP + 1.  (Round 4, producers with the entries
                                        // in scalar registers, ms per solve at 3 / 4: 32 x 4096x256 104 / 99, 47: 153 / 152, 64: 164 / 160; 5 did
                                        // not fit.  The lane-quad producers hold four columns' multipliers per slot: 128 VGPRs at 4.)
#endif
#define QRX_RPW_G 8                     // rows per producer and round: one 64-byte sector per lane
#ifndef QRX_RPW_AH
#define QRX_RPW_AH 3                    // row groups in flight per producer (ms per solve at 2 / 3 / 4: 16 x 4096x256 74.6 / 75.6 / 78.9,
                                        // 44: 97.7 / 95.0 / 99.9; tcp_pattern2: two, three and four groups run alike)
#endif
#ifndef QRX_RPW_QSTR
#define QRX_RPW_QSTR 132                // doubles between two row pairs of a product buffer (64 x 2 + 4: lane (c4, r) writes
                                        // pair r of column c4 -- eight lanes, eight different 16-byte groups of the 32 banks;
                                        // ms per solve at 132 / 136: 16 x 4096x256 69.8 / 71.7, 44: 92.3 / 93.3, 128 x 2048x128 34.6 / 34.9)
#endif
#define QRX_RPW_NVB 4                   // rounds of staged reflector entries in LDS
template <int NP, bool FLUSH, int W, int CW = 64>
__global__ void __launch_bounds__(64 * W)
k_qrx_pass_rpw(int p0, int nprob, int nwin, int lo, int m, int n, int ld, int coff, size_t tst, size_t vst, int j, int cur,
               double *__restrict__ T, const double *__restrict__ Vall, double *__restrict__ tpall,
               int32_t *__restrict__ srcall, int32_t *__restrict__ slotall, double *__restrict__ rdall,
               double *__restrict__ waall, const QrxStep *__restrict__ stepall, double *__restrict__ Rall,
               double *__restrict__ qtfall, const LmState *__restrict__ st, const int32_t *__restrict__ plist)
{
    // The adder's chain of dependent adds wants a SIMD to itself (an add every ~4.6 cycles is 87 % of the SIMD's fp64 issue
    // rate; sharing it round-robin with four producers stretched every add to ~20 cycles: 50 us per 4096-row pass whatever
    // NP).  The W waves of the workgroup land W / 4 on each SIMD; the waves that share wave 0's SIMD retire at once and the
    // other 3 W / 4 are the producers.  (Which waves those are is read from the hardware id at run time; if the placement
    // is ever uneven, waves of the adder's SIMD fill in, or surplus ones retire: always exactly NPR producers.)
    constexpr int NPR = 3 * W / 4, G = QRX_RPW_G, AH = QRX_RPW_AH, D = AH, RR = NPR * G;               // D: rounds per trip of the outer loops
    constexpr int PSTR = QRX_RPW_QSTR;                                  // doubles between two row pairs of a product buffer
    // CW: columns per workgroup.  A window's products go through LDS once in and once out -- 16 bytes of LDS traffic per
    // matrix element, ~67 bytes per clock all told -- and that, not memory, bounds a workgroup that has its CU to itself
    // (in-kernel clocks, 32 x 4096x256, a 4000-row pass: the adder busy 29.5 of 34 us, the producers idle 2/3 of the
    // time).  While the chip has CUs to spare a window is therefore dealt to TWO workgroups of 32 columns (`nwin` then
    // counts 32-column windows); the adder runs with its upper 32 lanes off.
    static_assert(CW == 64 || CW == 32, "whole or half windows");
    static_assert(G == 8, "a row group is one sector per lane");
    __shared__ int simd_of[W];
    constexpr int NPI = NP < QRX_C ? NP : 0;
    extern __shared__ __attribute__((aligned(16))) double pbw[];         // [2][RR / 2][PSTR]: the products of a round, row pairs x 64 columns x 2
                                                                         // then [QRX_RPW_NVB][NP + 1][RR], the staged reflector entries
    const int b_ = blockIdx.x, grp = b_ / (8 * nwin), r_ = b_ % (8 * nwin);
    const int pl = grp * 8 + (r_ & 7), win = r_ >> 3;
    if (pl >= nprob) return;
    const int p = plist ? plist[pl] : p0 + pl;                           // (plist: nprob counts its entries, k_qrx_list)
    if (p < 0 || (st && st[p].stage != ST_NEED_QR)) return;
    const int lane = threadIdx.x & 63, ldp = n + 1;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool adder = (wv == 0);
#ifdef QRX_DBG_CLK
    const long long ckk = wall_clock64();
#endif
    // roles: HW_REG_HW_ID bits 5:4 = the SIMD this wave runs on
    if (lane == 0) simd_of[wv] = (int)__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));
    __syncthreads();
    int pw = 0;
    {
        const int as = simd_of[0];
        int other = 0, same = 0, mine_other = 0, mine_same = 0;          // waves 1 .. W-1 off / on the adder's SIMD, and how many before this one
#pragma unroll
        for (int w2 = 1; w2 < W; ++w2) {
            const bool o = simd_of[w2] != as;
            if (w2 < wv) { mine_other += o ? 1 : 0; mine_same += o ? 0 : 1; }
            other += o ? 1 : 0; same += o ? 0 : 1;
        }
        if (!adder) {
            const bool o = simd_of[wv] != as;
            pw = o ? mine_other : other + mine_same;                     // off-SIMD waves first, then fill-ins
            pw = __builtin_amdgcn_readfirstlane(pw);                     // (wave-uniform by construction: keeps the loads' block offsets scalar)
            if (pw >= NPR + 1) return;                                   // shares the adder's SIMD (or surplus): retire -- but for one, the stager
        }
    }
    if (adder) __builtin_amdgcn_s_setprio(3);
    const QrxStep step = stepall[p];
    const bool refl = step.ajnorm != 0.0;
    const double ajj = step.ajj;
    int32_t *slotp = slotall + (size_t)p * ld;
    double *tpc = tpall + ((size_t)p * 2 + cur) * QRX_C * ldp;
    double *Tp = T + (size_t)p * tst;
    const int jb = j & ~7, r0 = j & 7, mrel = m - jb;
    const double *vc = Vall + ((size_t)p * 2 + cur) * QRX_C * vst + jb;
    const double *vo = Vall + ((size_t)p * 2 + (cur ^ 1)) * QRX_C * vst + jb;
    const bool inw = CW == 64 || lane < CW;
    const int col = inw ? ld - CW * (win + 1) + lane : ld - 1;
    const int kslot = (inw && col >= coff + lo) ? slotp[col] : -1;
    const bool act = kslot > j;
    const int k = act ? kslot : n;
    double tq[NP > 0 ? NP : 1];
#pragma unroll
    for (int q = 0; q < NP; ++q) tq[q] = tpc[(size_t)q * ldp + k];
    const double rk0 = (adder && act && k < n) ? rdall[(size_t)p * n + k] : 0.0;
    const double wa0 = (adder && act && k < n) ? waall[(size_t)p * n + k] : 1.0;
    double rowj = 0.0;
    if (adder) {                                                        // row j with its pending updates (becomes final in the tail)
        double e = Tp[qrx_at(j, col, ld)];
#pragma unroll
        for (int q = 0; q < NP; ++q) e = e - tq[q] * vc[(size_t)q * vst + r0];
        rowj = e;
    }
    const int nblk = (mrel + 7) >> 3;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(Tp + (size_t)(jb >> 3) * ld * 8, 0, (int)((size_t)nblk * ld * 64), 0x00020000);
    const unsigned so = act ? (unsigned)col * 64u : 0x80000000u;
    const unsigned ko = act ? (unsigned)(coff + k) * 64u : 0x80000000u;
    const unsigned ldb = (unsigned)ld * 64u;
    const int nround = (mrel + RR - 1) / RR;                            // producers: rounds 0 .. nround-1, adder: 1 .. nround
    const int ntile = (nround + 1 + D - 1) / D;
    double *pb0 = pbw, *pb1 = pbw + (size_t)(RR / 2) * PSTR;

    if (adder) {
        // ------------------------------------------------------------------------------------------------------------
        // the ordered sum (:652-653): the products of round t - 1 while the producers form those of round t; a producer's
        // eight rows at a time, the reads of the next two producers' rows in flight
        double s = 0.0;
        // The reads are issued by hand, three producers' rows (12 reads: the counter holds 15) ahead of the adds, and waited
        // for chunk by chunk (LDS returns in order).  Left to the compiler the round became "twelve reads, wait for all of
        // them, twenty-four adds": the LDS latency in full once per 24 rows, 16-20 cycles per row instead of the chain's 4.6.
        typedef double qrx_v2d __attribute__((ext_vector_type(2)));
        auto consume = [&](const double *half) __attribute__((always_inline)) {
            const unsigned addr = (unsigned)(size_t)(half + 2 * lane);                 // pair pr at addr + pr * 1024 bytes
            qrx_v2d w4[4][4];
#define QRX_RD4(c)                                                                                                  \
            asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\t"                       \
                         "ds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"                           \
                         : "=&v"(w4[(c) & 3][0]), "=&v"(w4[(c) & 3][1]), "=&v"(w4[(c) & 3][2]), "=&v"(w4[(c) & 3][3])  \
                         : "v"(addr), "n"(((c) * 4) * PSTR * 8), "n"(((c) * 4 + 1) * PSTR * 8), "n"(((c) * 4 + 2) * PSTR * 8),              \
                           "n"(((c) * 4 + 3) * PSTR * 8) : "memory")
#define QRX_WAIT4(c, N)                                                                                             \
            asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(w4[(c) & 3][0]), "+v"(w4[(c) & 3][1]), "+v"(w4[(c) & 3][2]), "+v"(w4[(c) & 3][3]))
            static_assert(NPR >= 3 && NPR * 4 * PSTR * 8 <= 65536, "three chunks in flight; immediate offsets of the reads");
            QRX_RD4(0); QRX_RD4(1); QRX_RD4(2);
#pragma unroll
            for (int c = 0; c < NPR; ++c) {
                // chunk c complete (at most two younger chunks outstanding), then the chunk three ahead is requested
                if (c + 2 < NPR) { QRX_WAIT4(c, 8); }
                else if (c + 1 < NPR) { QRX_WAIT4(c, 4); }
                else { QRX_WAIT4(c, 0); }
                if (c + 3 < NPR) {
                    switch ((c + 3) % 16) {                               // (the chunk index must be a literal for the immediates)
#define QRX_CASE(k) case k: QRX_RD4(k); break;
                    QRX_CASE(3) QRX_CASE(4) QRX_CASE(5) QRX_CASE(6) QRX_CASE(7) QRX_CASE(8) QRX_CASE(9) QRX_CASE(10) QRX_CASE(11)
                    QRX_CASE(12) QRX_CASE(13) QRX_CASE(14) QRX_CASE(15)
#undef QRX_CASE
                    default: break;
                    }
                }
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    s = s + w4[c & 3][h].x;                             // :653, rows ascending
                    s = s + w4[c & 3][h].y;
                }
            }
#undef QRX_RD4
#undef QRX_WAIT4
        };
        auto consume_edge = [&](const double *half, int rbase) __attribute__((always_inline)) {    // the round of row j, the last round
            const double2 *src2 = reinterpret_cast<const double2 *>(half) + lane;
#pragma unroll 1
            for (int pr = 0; pr < RR / 2; ++pr) {
                const double2 ww = src2[(size_t)pr * (PSTR / 2)];
                const int row = rbase + 2 * pr;
                if (row >= r0 && row < mrel) s = s + ww.x;              // uniform
                if (row + 1 >= r0 && row + 1 < mrel) s = s + ww.y;
            }
        };
#ifdef QRX_DBG_CLK
        long long ck0 = wall_clock64(), cwork = 0, cwait = 0;
#endif
        qrx_lds_barrier();
#ifdef QRX_DBG_CLK
        long long ck1 = wall_clock64();
#endif
#pragma unroll 1
        for (int t = 0; t < ntile * D; ++t) {
#ifdef QRX_DBG_CLK
            const long long ca = wall_clock64();
#endif
            if (t >= 1 && t <= nround) {
                const int rbase = (t - 1) * RR;
                const double *half = ((t - 1) & 1) ? pb1 : pb0;
                if (inw) {                                              // (half windows: the upper 32 lanes stay off, LDS reads included)
                    if (rbase >= r0 && rbase + RR <= mrel) consume(half);
                    else consume_edge(half, rbase);
                }
            }
#ifdef QRX_DBG_CLK
            const long long cb = wall_clock64();
#endif
            qrx_lds_barrier();
#ifdef QRX_DBG_CLK
            cwork += cb - ca; cwait += wall_clock64() - cb;
#endif
        }
#ifdef QRX_DBG_CLK
        if (lane == 0 && j >= 99 && j <= 101 && blockIdx.x < 1)
            printf("rpw adder wg %d j=100 W=%d NP=%d: head %lld first-barrier %lld consume %lld barrier-wait %lld rounds %d (x10 ns) simd %d\n",
                   blockIdx.x, W, NP, ck0 - ckk, ck1 - ck0, cwork, cwait, nround, simd_of[0]);
#endif
        if (!act) return;
        qrx_pass_tail<NP, FLUSH>(p, j, k, col, coff + k, m, n, ld, cur, vst, s, rowj, rk0, wa0, refl, ajj, tq, Tp, vc, vo, tpall,
                                 rdall, waall, Rall, qtfall);
        return;
    }
    // ------------------------------------------------------------------------------------------------------------
    // Producers that read WHOLE SECTORS PER LANE QUAD.  In the lane-per-column shape every 16-byte load instruction
    // touches 64 sectors for a quarter of each, and that is what bounds a CU that has the pass to itself
    // (profiles/ubench/tcp_pattern2.hip, us per 4096 x 64 window at 32 / 88 / 128 / 188 / 256 workgroups: 49 / 52 / 74 /
    // 91 / 103, and 123 / 133 / 207 / 262 / 285 with the flush's stores).  Here lanes 4c .. 4c+3 read column c's sector
    // (1 KB contiguous per instruction, sixteen columns; 26 / 30 / 43 / 62 / 91 us, and 48 / 53 / 131 / 147 / 216 with
    // stores: whole sectors written by one instruction): lane (c4, r) holds rows 2r, 2r+1 of the four columns 16 q + c4.
    // The reflector entries are then no longer wave-uniform; as extra vector loads they cost as much as the matrix loads
    // (+ 60 % with five slots: the CU's limit is load INSTRUCTIONS), so one otherwise idle wave on the adder's SIMD -- the
    // stager -- copies each round's entries into LDS three rounds ahead and the producers read their row pair back
    // (one 16-byte LDS read per slot and group).  Same operands, same operations, same order per element: bit-identical.
    constexpr int NVB = QRX_RPW_NVB;
    double *vsb = pbw + (size_t)2 * (RR / 2) * PSTR;                 // [NVB][NP + 1][RR]
    auto vsrc = [&](int q, int t) __attribute__((always_inline)) {  // rounds past the last re-read the last (never used)
        const int row = min(t, nround - 1) * RR;
        return ((q < NP) ? vc + (size_t)q * vst : (FLUSH ? vo : vc + (size_t)NPI * vst)) + row;
    };
    if (pw == NPR) {
        // the stager: entries of round t + 3 into LDS during round t (loaded two rounds before that)
        const bool on = lane < RR / 2;                               // 48 lanes x 16 bytes = the 96 rows of a round
        double2 hold[3][NP + 1];
        auto fetch = [&](double2 (&h)[NP + 1], int t) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q <= NP; ++q) h[q] = on ? *reinterpret_cast<const double2 *>(vsrc(q, t) + 2 * lane) : make_double2(0.0, 0.0);
        };
        auto stage = [&](const double2 (&h)[NP + 1], int t) __attribute__((always_inline)) {
            if (on) {
#pragma unroll
                for (int q = 0; q <= NP; ++q)
                    *reinterpret_cast<double2 *>(vsb + ((size_t)(t % NVB) * (NP + 1) + q) * RR + 2 * lane) = h[q];
            }
        };
        fetch(hold[0], 0); fetch(hold[1], 1); fetch(hold[2], 2);     // one trip to memory before the first barrier
        stage(hold[0], 0); stage(hold[1], 1); stage(hold[2], 2);
        fetch(hold[1], 3); fetch(hold[0], 4);
        qrx_lds_barrier();
        int t = 0;
#pragma unroll 1
        for (; t + 1 < ntile * D; t += 2) {
            stage(hold[1], t + 3); fetch(hold[1], t + 5);
            qrx_lds_barrier();
            stage(hold[0], t + 4); fetch(hold[0], t + 6);
            qrx_lds_barrier();
        }
        if (t < ntile * D) { stage(hold[1], t + 3); qrx_lds_barrier(); }
        return;
    }
    constexpr int NQ = CW / 16;                                      // load instructions per 8-row group
    const int c4 = lane >> 2, rq = lane & 3;
    unsigned soq[NQ], koq[NQ];
    double tqq[NP > 0 ? NP : 1][NQ];
#pragma unroll
    for (int q2 = 0; q2 < NQ; ++q2) {
        const int colq = ld - CW * (win + 1) + 16 * q2 + c4;
        const int ks = (colq >= coff + lo) ? slotp[colq] : -1;
        const bool aq = ks > j;
        const int kq = aq ? ks : n;
        soq[q2] = aq ? (unsigned)colq * 64u + 16u * rq : 0x80000000u;
        koq[q2] = aq ? (unsigned)(coff + kq) * 64u + 16u * rq : 0x80000000u;
#pragma unroll
        for (int q = 0; q < NP; ++q) tqq[q][q2] = tpc[(size_t)q * ldp + kq];
    }
    double2 aq_[AH][NQ];
    double2 vq[NP + 1];
    auto loadq = [&](double2 (&buf)[NQ], int t) __attribute__((always_inline)) {
        const unsigned boff = (unsigned)((t * RR + pw * G) >> 3) * ldb;
#pragma unroll
        for (int q2 = 0; q2 < NQ; ++q2) {
            const qrx_u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, soq[q2], boff, QRX_AUX_LOAD);
            buf[q2].x = __hiloint2double((int)w.y, (int)w.x);
            buf[q2].y = __hiloint2double((int)w.w, (int)w.z);
        }
    };
    auto vread = [&](int t) __attribute__((always_inline)) {
        const double *src = vsb + (size_t)(t % NVB) * (NP + 1) * RR + pw * G + 2 * rq;
#pragma unroll
        for (int q = 0; q <= NP; ++q) vq[q] = *reinterpret_cast<const double2 *>(src + (size_t)q * RR);
    };
    auto produceq = [&](const double2 (&buf)[NQ], int t) __attribute__((always_inline)) {
        const int rbase = t * RR + pw * G;
        double *dst = ((t & 1) ? pb1 : pb0) + (size_t)(pw * (G / 2) + rq) * PSTR + 2 * c4;
        double2 est[NQ];
#pragma unroll
        for (int q2 = 0; q2 < NQ; ++q2) {
            double e0 = buf[q2].x, e1 = buf[q2].y;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const double p0_ = tqq[q][q2] * vq[q].x, p1_ = tqq[q][q2] * vq[q].y;
                e0 = e0 - p0_;
                e1 = e1 - p1_;
            }
            double2 ww;                                             // rows outside the live range: garbage the adder skips
            ww.x = vq[NP].x * e0;
            ww.y = vq[NP].y * e1;
            *reinterpret_cast<double2 *>(dst + 32 * q2) = ww;
            if (FLUSH) { est[q2].x = e0; est[q2].y = e1; }
        }
        if (FLUSH) {
            // whole sectors, unconditionally (see the lane-per-column producer below); a slot's own position is a dead
            // column unless it is the column the slot is read from, so no lane of any workgroup reads what another writes
            const unsigned boff = (unsigned)(rbase >> 3) * ldb;
#pragma unroll
            for (int q2 = 0; q2 < NQ; ++q2) {
                qrx_u32x4 w;
                w.x = (unsigned)__double2loint(est[q2].x); w.y = (unsigned)__double2hiint(est[q2].x);
                w.z = (unsigned)__double2loint(est[q2].y); w.w = (unsigned)__double2hiint(est[q2].y);
                __builtin_amdgcn_raw_buffer_store_b128(w, rsrc, koq[q2], boff, QRX_AUX_STORE);
            }
        }
    };
#pragma unroll
    for (int i = 0; i < AH - 1; ++i) loadq(aq_[i], i);
#ifdef QRX_DBG_CLK
    long long pk0 = wall_clock64(), pwork = 0, pwait = 0, pmem = 0;
#endif
    qrx_lds_barrier();
    vread(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int kt = 0; kt < ntile; ++kt) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const int t = kt * D + i;
#ifdef QRX_DBG_CLK
            const long long pa = wall_clock64();
#endif
            loadq(aq_[(i + AH - 1) % AH], t + AH - 1);
            __builtin_amdgcn_sched_barrier(0);
#ifdef QRX_DBG_CLK
            if constexpr (NQ == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // (the group of this round and its reflector entries have arrived)
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const long long pm = wall_clock64();
            pmem += pm - pa;
#endif
            produceq(aq_[i % AH], t);
            __builtin_amdgcn_sched_barrier(0);
            vread(t + 1);                                           // staged by the barrier before this round
#ifdef QRX_DBG_CLK
            const long long pbk = wall_clock64();
#endif
            qrx_lds_barrier();
#ifdef QRX_DBG_CLK
            pwork += pbk - pa; pwait += wall_clock64() - pbk;
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#ifdef QRX_DBG_CLK
    if (lane == 0 && j >= 99 && j <= 101 && blockIdx.x < 1 && (pw == 0 || pw == NPR - 1))
        printf("rpw producer %d wg %d j=100 NP=%d: head %lld work %lld (of it waiting for loads %lld) barrier-wait %lld (x10 ns) rounds %d\n", pw, blockIdx.x, NP,
               pk0 - ckk, pwork, pmem, pwait, nround);
#endif
    return;
}

// The pass for a HANDFUL of problems (straggler rounds, one problem alone; factorisations of at most QRX_COL_MAX_WG
// (problem, column) pairs): a workgroup per trailing column instead of a lane per column -- n - j workgroups per problem where the other
// forms have ceil((n - j) / 64), so a lone 4096 x 256 problem still puts 256 workgroups on the chip.  The products of a
// column with the reflector are formed by the preparing waves; their ordered sum (:652-653) runs down the lanes of one
// wave exactly as NORM2's recurrence does in the pivot kernel (64 consecutive terms per lane in registers, the running
// sum handed on with a DPP wave shift): ~9.4 us per 4096 rows.  A workgroup writes its own column, its own entries of
// tp / R / qtf / rdiag / wa and nothing else.
// The update of a column stays ONE STEP BEHIND.  A 65536-row column (BASELINE config 5) cannot wait in registers for its
// multiplier, and as a second sweep (read, update, write: every column's workgroup at the same moment) the update cost
// as much again as the chain-bound sweep that forms the dot product.  So the sweep of step j applies step j - 1's
// update (multiplier and reflector are known), writes the column back and multiplies the fresh values with step j's
// reflector -- one read and one write of the trailing matrix per step, spread over the whole chain-bound sweep.  That is
// the deferred-update machinery of k_qrx_pass with a flush at every step and exactly one pending reflector (the pivot
// kernel applies it to the column it gathers, k_qrx_finish to the residual), without the physical move.  The eager form
// this replaced (update from the registers the products were formed from, in the same step) was equal at 1024 rows and
// slower from 2048 on (11.6 / 21.2 us per step at 2048 / 4096 rows against 10.3 / 17.2).
// The sweep is software-pipelined around a wave that does nothing but the chain: three PREPARING waves form the
// products of chunk c + 1 and hand them over while the chain wave adds chunk c (two LDS buffers, ONE LDS-only barrier per
// chunk), the loads of chunk c + 2 already in flight: a chunk costs its 4096-add chain plus the chain wave's LDS
// reads (9.9 us).  The buffers are dynamic LDS behind two pointers; a column of one chunk gets ONE buffer, sized to the
// column (4096 rows: 17.2 us per step against 19.1 with a static buf[2][...] array -- the generated addressing, not the
// allocation: padded back to 68 KB it stays at 17.3, and an empty launch costs the same at any LDS size, lds_launch.hip).
// Same values, same order as the deferred forms: bit-identical.
#ifndef QRX_COL_MAX_WG
#define QRX_COL_MAX_WG 1536             // factorisations of at most this many (problem, column) pairs take this form.  ms per
                                        // solve, this form / the wide row-parallel form on half windows (which moved the crossover down
                                        // from 3072 pairs): 4096 x 256: 4 problems 52.8 / 60.6, 6: 62.3 / 61.2, 8: 71.6 / 61.6, 12: 90.0 / 62.5;
                                        // 2048 x 128: 8: 17.2 / 20.6, 16: 22.6 / 20.9, 24: 27.9 / 21.3; 1024 x 64: 16: 6.7 / 8.1, 48: 11.4 / 10.3
#endif
#define QRX_COL_EL 64                   // terms per lane of the ordered sum (chunks of 4096 rows)
template <bool PEND>
__global__ void __launch_bounds__(256)
k_qrx_pass_col(int m, int n, int ld, size_t tst, size_t vst, int j, int cur, double *__restrict__ T,
                    const double *__restrict__ Vall, double *__restrict__ tpall, const int32_t *__restrict__ srcall,
                    double *__restrict__ rdall, double *__restrict__ waall, const QrxStep *__restrict__ stepall,
                    double *__restrict__ Rall, double *__restrict__ qtfall, const LmState *__restrict__ st, const int32_t *__restrict__ plist)
{
    constexpr int EL = QRX_COL_EL, CAP = 64 * EL, NPREP = 192, NPAIR = CAP / 2, PPT = (NPAIR + NPREP - 1) / NPREP;   // 2048 row pairs, 11 per thread
    // dynamic LDS: two product buffers of CAP + 128 doubles -- one, sized to the column, when the column is a single chunk
    extern __shared__ __attribute__((aligned(16))) double bufs[];
    double *buf[2] = {bufs, bufs + ((m - (j & ~7)) > CAP ? CAP + 128 : 0)};
    __shared__ double xch[2];
    const int p = plist ? plist[blockIdx.y] : (int)blockIdx.y;
    if (p < 0 || (st && st[p].stage != ST_NEED_QR)) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const bool chain = (wid == NPREP / 64);                             // (wave-uniform)
    const int k = j + 1 + blockIdx.x;
    const int ldp = n + 1, jb = j & ~7;
    const QrxStep step = stepall[p];
    const bool refl = step.ajnorm != 0.0;
    const double ajj = step.ajj;
    const int col = srcall[(size_t)p * ldp + k];
    double *Tp = T + (size_t)p * tst;
    const double *vc = Vall + ((size_t)p * 2 + cur) * QRX_C * vst + jb;                       // slot 0: the pending reflector (PEND) or the new one
    const double *vnew = PEND ? Vall + ((size_t)p * 2 + (cur ^ 1)) * QRX_C * vst + jb : vc;   // step j's reflector
    const double tq0 = PEND ? tpall[((size_t)p * 2 + cur) * QRX_C * ldp + k] : 0.0;           // step j - 1's multiplier for this column
    const int r0 = j & 7, len = m - jb;
    double *colp = Tp + qrx_at(jb, col, ld);
    double rk0 = 0.0, wa0 = 1.0;
    if (tid == 0 && k < n) { rk0 = rdall[(size_t)p * n + k]; wa0 = waall[(size_t)p * n + k]; }
#ifdef QRX_DBG_CLK
    const long long c0 = wall_clock64();
    long long cw = 0, cr = 0, cc = 0;                                   // chain wave: barrier waits, LDS reads, the chains
#endif
    const size_t blk = (size_t)ld * 8;
    const int nch = (len + CAP - 1) / CAP;
    // A lane takes a PAIR of rows (16 bytes), four adjacent lanes a sector of the row-blocked matrix: a load or store
    // instruction of the wave covers 16 whole sectors of the column and 1 KB of each reflector.
    double2 av[PPT], vn[PPT], vp[PEND ? PPT : 1];
    // (the reflector entries first: their addresses do not depend on the column index, so for chunk 0 they are on their
    // way while that index is still being fetched)
    auto loadchunk = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            const int pr = tid + u * NPREP, rb = c * CAP + 2 * pr;      // rel row of the pair
            const bool in = pr < NPAIR && rb < len;
            vn[u] = in ? *reinterpret_cast<const double2 *>(vnew + rb) : make_double2(0.0, 0.0);
            if (PEND) vp[u] = in ? *reinterpret_cast<const double2 *>(vc + rb) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            const int pr = tid + u * NPREP, rb = c * CAP + 2 * pr;
            const bool in = pr < NPAIR && rb < len;
            av[u] = in ? *reinterpret_cast<const double2 *>(colp + (size_t)(rb >> 3) * blk + (rb & 7)) : make_double2(0.0, 0.0);
        }
    };
    double s = 0.0, rowj = 0.0;                                         // the running sum lives in the chain wave, row j in thread 0
    // (Two loops, not one with two branches: the registers of the chunk in flight and the chain wave's 64 terms would
    // otherwise be live together -- 293 registers, one workgroup per CU instead of two.  Both sides pass the same barriers.)
    if (!chain) {
        loadchunk(0);
        for (int c = 0; c < nch; ++c) {
            const int base = c * CAP;
            // the runs the chain wave reads: whole trips of eight (zeros behind the column's end); a single-chunk
            // column's buffer is sized to exactly these
            const int wl = ((((min(CAP, len - base) + EL - 1) / EL) + 7) & ~7) * EL;
            // buf[c & 1] was last read by the chain of chunk c - 2, which ended before the chain wave joined the barrier
            // of chunk c - 1
#pragma unroll
            for (int u = 0; u < PPT; ++u) {
                const int pr = tid + u * NPREP, rel = 2 * pr;
                if (pr < NPAIR && rel < wl) {
                    const int ra = base + rel, rbb = ra + 1;
                    const bool la = ra >= r0 && jb + ra < m, lb = rbb >= r0 && jb + rbb < m;
                    double2 e = av[u];
                    if (PEND) {                                         // :655 of step j - 1, rows j and below
                        if (la) e.x = e.x - tq0 * vp[u].x;
                        if (lb) e.y = e.y - tq0 * vp[u].y;
                    }
                    if (ra == r0) rowj = e.x;                           // (rel row r0 = row j: chunk 0, one of threads 0 .. 3)
                    if (rbb == r0) rowj = e.y;
                    double2 w;
                    w.x = la ? vn[u].x * e.x : 0.0;                     // :653
                    w.y = lb ? vn[u].y * e.y : 0.0;
                    *reinterpret_cast<double2 *>(buf[c & 1] + rel + 2 * (rel / EL)) = w;
                    if (PEND) {
                        double *wp = colp + (size_t)(ra >> 3) * blk + (ra & 7);
                        if (la && lb) *reinterpret_cast<double2 *>(wp) = e;
                        else {
                            if (la) wp[0] = e.x;
                            if (lb) wp[1] = e.y;
                        }
                    }
                }
            }
            if (c + 1 < nch) loadchunk(c + 1);                          // in flight during the barrier and the chain
            qrx_lds_barrier();                                          // orders LDS traffic only: the loads stay in flight
        }
    } else {
        __builtin_amdgcn_s_setprio(3);                                   // the chain wave goes first on its SIMD
        for (int c = 0; c < nch; ++c) {
            const int base = c * CAP;
#ifdef QRX_DBG_CLK
            const long long q0 = wall_clock64();
#endif
            qrx_lds_barrier();
#ifdef QRX_DBG_CLK
            const long long q1 = wall_clock64();
#endif
            const int cl = min(CAP, len - base), nl = (cl + EL - 1) / EL;
            double d[EL];
            const double2 *mine = reinterpret_cast<const double2 *>(buf[c & 1] + lane * (EL + 2));
#pragma unroll
            for (int u = 0; u < EL / 2; ++u) { const double2 v2 = mine[u]; d[2 * u] = v2.x; d[2 * u + 1] = v2.y; }
#ifdef QRX_DBG_CLK
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const long long q2 = wall_clock64();
            cw += q1 - q0; cr += q2 - q1;
#endif
            double t = s;
            // eight runs per trip (a taken branch costs ~40 cycles here; t is the same in every lane on entry, so the shift
            // before run 0 changes nothing; the terms behind the column's end are +0.0 and the sum is never -0.0, so the
            // runs that fill up the last trip hand it on unchanged: the result is in the last lane of the last trip)
            const int ntrip = (nl + 7) >> 3;
#pragma unroll 1
            for (int g = 0; g < ntrip; ++g) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    t = nlh_wave_shr1(t);
#pragma unroll
                    for (int u = 0; u < EL; ++u) t = t + d[u];         // :653, rows ascending (terms outside the rows: +0.0)
                }
            }
            const int lo_ = __builtin_amdgcn_readlane(__double2loint(t), 8 * ntrip - 1);
            const int hi_ = __builtin_amdgcn_readlane(__double2hiint(t), 8 * ntrip - 1);
            s = __hiloint2double(hi_, lo_);
#ifdef QRX_DBG_CLK
            cc += wall_clock64() - q2;
#endif
        }
    }
    if (chain && lane == 0) xch[0] = s;
    if (tid == (r0 >> 1)) xch[1] = rowj;                                // (the thread that held row j's pair)
    __syncthreads();                                                    // (also: the column as written above is visible to thread 0)
    s = xch[0];
    rowj = xch[1];
#ifdef QRX_DBG_CLK
    const long long c1 = wall_clock64();
#endif
    if (tid == 0) {
        const double tq1[1] = {tq0};
        qrx_pass_tail<PEND ? 1 : 0, PEND>(p, j, k, col, col, m, n, ld, cur, vst, s, rowj, rk0, wa0, refl, ajj, tq1, Tp, vc,
                                           Vall + ((size_t)p * 2 + (cur ^ 1)) * QRX_C * vst + jb, tpall, rdall, waall, Rall, qtfall);
    }
#ifdef QRX_DBG_CLK
    if (tid == 0 && (j == 100 || j == 400) && (blockIdx.x == 0 || blockIdx.x == 100))
        printf("pass j=%d wg %d: sweep %lld tail %lld (x10 ns)\n", j, blockIdx.x, c1 - c0, wall_clock64() - c1);
    if (chain && lane == 0 && (j == 100 || j == 400) && (blockIdx.x == 0 || blockIdx.x == 100))
        printf("pass_col chain wave j=%d wg %d: %d chunks, barrier wait %lld, lds read %lld, chain %lld (x10 ns)\n", j, blockIdx.x, nch, cw, cr, cc);
#endif
}

// After the last step: wa4 = Q^T f (:241-253; rows < n are the qtf entries, the rest carries the pending
// updates of the residual column), the diagonal of R (:251), then the outer-loop head.
__global__ void __launch_bounds__(256)
k_qrx_finish(int m, int n, int ld, int coff, size_t tst, size_t vst, int cur, int np, const double *__restrict__ T, QrxWs w, double *__restrict__ Rall,
             LmVecs v, double *__restrict__ wa4all, double *__restrict__ scratch_all, const double *__restrict__ xall,
             LmState *__restrict__ st, double factor, double gtol)
{
    __shared__ double red[64];
    __shared__ double scratch[3 * NLH_NCH + 8];
    const int p = blockIdx.x;
    LmState *s = st ? st + p : nullptr;
    if (s && s->stage != ST_NEED_QR) return;
    const int tid = threadIdx.x, BS = blockDim.x, ldp = n + 1;
    const bool first = (!s) || (s->inner_pass == 0);
    double *w4 = first ? (wa4all + (size_t)p * m) : (scratch_all + (size_t)p * m);
    const double *a = T + (size_t)p * tst;
    const double *Vc = w.V + ((size_t)p * 2 + cur) * QRX_C * vst;
    const double *tpc = w.tp + ((size_t)p * 2 + cur) * QRX_C * ldp;
    const double *rdiag = w.rdiag + (size_t)p * n;
    double *qtf = v.qtf + (size_t)p * n;
    double *R = Rall + (size_t)p * n * n;
    double tk[QRX_C];
#pragma unroll
    for (int q = 0; q < QRX_C; ++q) tk[q] = (q < np) ? tpc[(size_t)q * ldp + n] : 0.0;
    for (int i = tid; i < m; i += BS) {
        double e;
        if (i < n) {
            e = qtf[i];
        } else {
            e = a[qrx_at(i, coff + n, ld)];
#pragma unroll
            for (int q = 0; q < QRX_C; ++q)
                if (q < np) e = e - tk[q] * Vc[(size_t)q * vst + i];
        }
        w4[i] = e;
    }
    for (int k = tid; k < n; k += BS) { R[(size_t)k * n + k] = rdiag[k]; v.rdiag[(size_t)p * n + k] = rdiag[k]; }
    __syncthreads();
    if (!s) return;
    if (tid == 0) { s->factor_kind = 1; s->qr_count += 1; }
    if (first) {
        lm_head<true>(n, R, n, v.ipvt + (size_t)p * n, v.acnorm + (size_t)p * n, qtf, xall + (size_t)p * n,
                      v.diag + (size_t)p * n, v.diag_prev + (size_t)p * n, s, factor, gtol, ST_QR_READY, red, scratch);
    } else {
        if (tid == 0) s->stage = ST_QR_READY;
    }
}

// two product buffers of 3 W / 4 * 8 rows x 64 lanes x 8 bytes (row pairs QRX_RPW_QSTR doubles apart), + the staged reflector entries
static constexpr size_t qrx_rpw_lds(int W)
{
    return sizeof(double) * ((size_t)(3 * W / 4) * QRX_RPW_G * QRX_RPW_QSTR + (size_t)QRX_RPW_NVB * (QRX_RPW_MAXNP + 1) * (3 * W / 4) * QRX_RPW_G);
}

template <int NP, bool FLUSH>
static void launch_pass(int rp, hipStream_t stream, int p0, int nprob, int lo, int m, int n, int ld, int coff, size_t tst, size_t vst, int j, int cur, double *T, const QrxWs &w,
                        double *R, double *qtf, const LmState *st)
{
    // One column per lane: measured against two and four columns per lane (fewer waves, the LDS row shared by more
    // elements) on 512 x 4096x256, 1024 x 2048x128 and a single problem, more waves won every time.
    const int nwin = (n + 1 - lo + 63) / 64;                            // live physical columns coff + lo .. coff + n
    const dim3 grid((unsigned)(((nprob + 7) / 8) * 8 * nwin));
    // the row-parallel forms over the compacted problems when the factorisation has a list (qrx_factor)
    const int npl = w.plist ? w.ny : nprob;
    const int32_t *pls = w.plist;
    const dim3 gridl((unsigned)(((npl + 7) / 8) * 8 * nwin));
    if constexpr (NP >= 8) rp = 0;
    // the wide row-parallel form holds at most QRX_RPW_MAXNP pending updates (registers); a launch that inherits more from the
    // form before it takes the four-wave form until the next flush
    if constexpr (NP > QRX_RPW_MAXNP) { if (rp == 16 || rp == QRX_RP_HALF || rp == 8) rp = 4; }
    if constexpr (NP < 8) {
    if constexpr (NP <= QRX_RPW_MAXNP) {
    if (rp == 8)                                                        // eight waves (adder, stager, six producers), two workgroups per CU
        hipLaunchKernelGGL((k_qrx_pass_rpw<NP, FLUSH, 8>), gridl, dim3(64 * 8), qrx_rpw_lds(8), stream, p0, npl, nwin, lo, m, n, ld, coff, tst, vst, j, cur,
                           T, (const double *)w.V, w.tp, w.src, w.slotof, w.rdiag, w.wa, (const QrxStep *)w.step, R, qtf, st, pls);
    else if (rp == QRX_RP_HALF) {                                       // the wide form on half windows
        const int nsw = (n + 1 - lo + 31) / 32;
        hipLaunchKernelGGL((k_qrx_pass_rpw<NP, FLUSH, 16, 32>), dim3((unsigned)(((npl + 7) / 8) * 8 * nsw)), dim3(64 * 16), qrx_rpw_lds(16), stream, p0, npl,
                           nsw, lo, m, n, ld, coff, tst, vst, j, cur, T, (const double *)w.V, w.tp, w.src, w.slotof, w.rdiag, w.wa, (const QrxStep *)w.step, R, qtf, st, pls);
    } else if (rp == 16)
        hipLaunchKernelGGL((k_qrx_pass_rpw<NP, FLUSH, 16>), gridl, dim3(64 * 16), qrx_rpw_lds(16), stream, p0, npl, nwin, lo, m, n, ld, coff, tst, vst, j, cur,
                           T, (const double *)w.V, w.tp, w.src, w.slotof, w.rdiag, w.wa, (const QrxStep *)w.step, R, qtf, st, pls);
    }
    if (rp == 4)
        hipLaunchKernelGGL((k_qrx_pass_rp<NP, FLUSH, 4>), gridl, dim3(64 * 4), 0, stream, p0, npl, nwin, lo, m, n, ld, coff, tst, vst, j, cur,
                           T, (const double *)w.V, w.tp, w.src, w.slotof, w.rdiag, w.wa, (const QrxStep *)w.step, R, qtf, st, pls);
    }
    static const int share_env = [] { const char *e = getenv("NLH_QRX_SHARE"); return e ? atoi(e) : 1; }();
    if (rp == 0 && share_env && nwin >= 2 && nwin <= QRX_SHARE_MAXWIN)
        hipLaunchKernelGGL((k_qrx_pass<NP, FLUSH, true>), dim3((unsigned)nprob), dim3(64 * nwin), 0, stream, p0, nprob, nwin, lo, m, n, ld, coff,
                           tst, vst, j, cur, T, (const double *)w.V, w.tp, w.src, w.slotof, w.rdiag, w.wa, (const QrxStep *)w.step, R, qtf, st);
    else if (rp == 0)
        hipLaunchKernelGGL((k_qrx_pass<NP, FLUSH, false>), grid, dim3(64), 0, stream, p0, nprob, nwin, lo, m, n, ld, coff, tst, vst, j, cur,
                           T, (const double *)w.V, w.tp, w.src, w.slotof, w.rdiag, w.wa, (const QrxStep *)w.step, R, qtf, st);
}

// A pass with np pending updates; flushing ones exist for np = 1, 3 and QRX_C - 1 (flush periods 2, 4 and QRX_C).
static constexpr bool qrx_can_flush(int np) { return np == 1 || np == 3 || np == QRX_RPW_MAXNP || (QRX_C > 8 && np == 7) || np == QRX_C - 1; }

template <int NP>
static void dispatch_pass(int np, bool flush, int rp, hipStream_t stream, int p0, int nprob, int lo, int m, int n, int ld, int coff, size_t tst, size_t vst,
                          int j, int cur, double *T, const QrxWs &w, double *R, double *qtf, const LmState *st)
{
    if (np == NP) {
        if (flush) {
            if constexpr (qrx_can_flush(NP)) launch_pass<NP, true>(rp, stream, p0, nprob, lo, m, n, ld, coff, tst, vst, j, cur, T, w, R, qtf, st);
        } else {
            if constexpr (NP < QRX_C - 1) launch_pass<NP, false>(rp, stream, p0, nprob, lo, m, n, ld, coff, tst, vst, j, cur, T, w, R, qtf, st);
        }
    } else if constexpr (NP < QRX_C - 1) {
        dispatch_pass<NP + 1>(np, flush, rp, stream, p0, nprob, lo, m, n, ld, coff, tst, vst, j, cur, T, w, R, qtf, st);
    }
}

// Per device (called when a handle is created on it): the column sweep of long columns asks for more than 64 KB of
// dynamic LDS (two product buffers), which has to be allowed on every device the kernel is launched on.
template <int NP>
static void qrx_rpw_attr()
{
    const int lim = (int)qrx_rpw_lds(16);
    hipFuncSetAttribute((const void *)k_qrx_pass_rpw<NP, false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    hipFuncSetAttribute((const void *)k_qrx_pass_rpw<NP, false, 16, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    if constexpr (qrx_can_flush(NP)) {
        hipFuncSetAttribute((const void *)k_qrx_pass_rpw<NP, true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
        hipFuncSetAttribute((const void *)k_qrx_pass_rpw<NP, true, 16, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    }
    if constexpr (NP < QRX_RPW_MAXNP) qrx_rpw_attr<NP + 1>();
}

void qrx_init_device()
{
    const int lim = (int)(sizeof(double) * 2 * (64 * QRX_COL_EL + 128));
    hipFuncSetAttribute((const void *)k_qrx_pass_col<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    hipFuncSetAttribute((const void *)k_qrx_pass_col<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    qrx_rpw_attr<0>();                                           // the sixteen-wave row-parallel pass: 120 KB
}

void qrx_factor(hipStream_t stream, int nprob, int m, int n, const double *J, double *T, const double *fvec,
                double *R, LmVecs v, double *wa4, double *scratch, const double *x, LmState *st, double factor,
                double gtol, void *ws, const QrxTimer *tm, int nact)
{
    QrxWs w;
    qrx_carve(ws, nprob, m, n, &w);
    const int ld = qrx_ld(n), coff = qrx_coff(n);
    const size_t tst = qrx_tstride(m, n), vst = qrx_vstride(m);
    auto tb = [&](int which, hipStream_t s) { if (tm) tm->begin(tm->ctx, which, s); };
    auto te = [&](int which, hipStream_t s) { if (tm) tm->end(tm->ctx, which, s); };
    tb(2, stream);
    if (J) {
        const size_t total = (size_t)((m + 7) >> 3) * n;
        const unsigned gx = (unsigned)std::min<size_t>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(k_qrx_transpose, dim3(gx, nprob), dim3(256), 0, stream, m, n, ld, coff, tst, J, T, (const LmState *)st);
    }
    if (nact <= 0 || nact > nprob) nact = nprob;
    // a workgroup per column for the initial norms while the launch's chains would otherwise leave the chip idle (a thread
    // per column walks its 4096 rows alone: 800 us for 47 x 4096x256, whatever the count; k_qrx_init<true> is HBM-bound
    // only from ~1000 problems on)
    static const long initn_env = [] { const char *e = getenv("NLH_QRX_INITN"); return e ? atol(e) : 32768L; }();
    // launches with (columns x problems) grids cover the problems that work, not the batch (k_qrx_list)
    const bool use_list = st != nullptr && nact < nprob && (long)nact * n <= std::max<long>(QRX_COL_MAX_WG, initn_env);
    const int ny = use_list ? nact : nprob;
    if (use_list) hipLaunchKernelGGL(k_qrx_list, dim3(1), dim3(256), 0, stream, nprob, nact, st, w.plist);
    else w.plist = nullptr;
    w.ny = ny;
    if ((long)nact * n <= std::max<long>(QRX_COL_MAX_WG, initn_env)) {
        hipLaunchKernelGGL(k_qrx_init<false>, dim3(nprob), dim3(256), 0, stream, m, n, ld, coff, tst, T, fvec, w, v, (const LmState *)st);
        hipLaunchKernelGGL(k_qrx_init_norms, dim3(n, ny), dim3(256), 0, stream, m, n, ld, coff, tst, (const double *)T, w, v,
                           (const LmState *)st);
    } else {
        hipLaunchKernelGGL(k_qrx_init<true>, dim3(nprob), dim3(256), 0, stream, m, n, ld, coff, tst, T, fvec, w, v, (const LmState *)st);
    }
    te(2, stream);
    // (Measured and dropped: the two halves of the batch on two streams, half B's pivot kernel under half A's pass, with
    // events keeping the passes from overlapping each other -- 1035 ms instead of 999 ms per 512 x 4096x256 solve; the
    // cross-stream event waits cost more than the pivot latency they hide.  Sub-batches on host threads, which need no
    // cross-stream ordering, do hide it: nlh_lm.hip, lm_sub_batches.)
    static const int forced_period = [] { const char *e = getenv("NLH_QRX_PERIOD"); return e ? atoi(e) : 0; }();
    static const long rp_env = [] { const char *e = getenv("NLH_QRX_RP"); return e ? atol(e) : -1L; }();
    const long rp_max = rp_env >= 0 ? rp_env : QRX_RP_MAX_WG;
    static const long rpw16_env = [] { const char *e = getenv("NLH_QRX_RPW16"); return e ? atol(e) : -1L; }();
    const long rpw16_max = std::min(rp_max, rpw16_env >= 0 ? rpw16_env : (long)QRX_RPW16_MAX_WG);
    static const int col_env = [] { const char *e = getenv("NLH_QRX_COL"); return e ? atoi(e) : -1; }();
    if (col_env >= 0 ? nact <= col_env : (long)nact * n <= QRX_COL_MAX_WG) {
        // two product buffers for columns of several chunks, one sized to the column otherwise
        auto coll_lds = [](int m_, int j_) {
            const int len = m_ - (j_ & ~7);
            const int nrun = (((len + QRX_COL_EL - 1) / QRX_COL_EL) + 7) & ~7;        // the chain wave adds whole trips of eight runs
            return sizeof(double) * (size_t)(len > 64 * QRX_COL_EL ? 2 * (64 * QRX_COL_EL + 128) : nrun * (QRX_COL_EL + 2));
        };
        auto pivot = [&](int j, int cur, int np, int pf) {
            if (m <= 2048)
                hipLaunchKernelGGL((k_qrx_pivot<32, false, true>), dim3(nprob), dim3(256), 0, stream, 0, m, n, ld, coff, tst, vst, j, cur, np, pf, T, w, R, v,
                                   (const LmState *)st);
            else if (m - j > 64 * 64 && m - j <= 64 * QRX_LONG_EL * QRX_LONG_MAXCH) {
                // long column: the scaling of the reflector as a launch of its own over the whole chip (flush bit 2), and
                // from step 1 on (no physical interchange, at most one update pending) the gather too: search and
                // bookkeeping (flush bit 3), gather, NORM2 as three launches -- 218 -> ~200 us per 65536-row step
                if (j >= 1 && np <= 1) {
                    hipLaunchKernelGGL((k_qrx_pivot<64, true>), dim3(nprob), dim3(QRX_LONG_THREADS), 0, stream, 0, m, n, ld, coff, tst, vst, j, cur, np,
                                       pf | 4 | 8, T, w, R, v, (const LmState *)st);
                    hipLaunchKernelGGL(k_qrx_gather_long, dim3((unsigned)(((m - (j & ~7)) / 2 + 1023) / 1024 + 1), nprob), dim3(256), 0, stream, m, ld, tst, vst,
                                       j, cur, np, pf, (const double *)T, w, (const LmState *)st);
                    hipLaunchKernelGGL(k_qrx_norm_long, dim3(nprob), dim3(QRX_NORM_THREADS), 0, stream, m, n, vst, j, cur, np, pf, w, (const LmState *)st);
                } else
                hipLaunchKernelGGL((k_qrx_pivot<64, true>), dim3(nprob), dim3(QRX_LONG_THREADS), 0, stream, 0, m, n, ld, coff, tst, vst, j, cur, np, pf | 4, T, w,
                                   R, v, (const LmState *)st);
                hipLaunchKernelGGL(k_qrx_scale_long, dim3((unsigned)(((m - j) / 2 + 1023) / 1024 + 1), nprob), dim3(256), 0, stream, m, vst, j, cur, np, pf,
                                   w, (const LmState *)st);
            } else
                hipLaunchKernelGGL((k_qrx_pivot<64, false, true>), dim3(nprob), dim3(256), 0, stream, 0, m, n, ld, coff, tst, vst, j, cur, np, pf, T, w, R, v,
                                   (const LmState *)st);
        };
        {
            // a workgroup per trailing column, the update one step behind (k_qrx_pass_col): one pending reflector from
            // step 1 on, a bank switch at every step, no physical move
            int cur = 0;
            for (int j = 0; j < n; ++j) {
                tb(0, stream);
                pivot(j, cur, j > 0 ? 1 : 0, j > 0 ? 1 : 0);
                te(0, stream);
                tb(1, stream);
                if (j == 0)
                    hipLaunchKernelGGL(k_qrx_pass_col<false>, dim3(n - j, ny), dim3(256), coll_lds(m, j), stream, m, n, ld, tst, vst, j, cur, T,
                                       (const double *)w.V, w.tp, (const int32_t *)w.src, w.rdiag, w.wa, (const QrxStep *)w.step, R, v.qtf,
                                       (const LmState *)st, (const int32_t *)w.plist);
                else
                    hipLaunchKernelGGL(k_qrx_pass_col<true>, dim3(n - j, ny), dim3(256), coll_lds(m, j), stream, m, n, ld, tst, vst, j, cur, T,
                                       (const double *)w.V, w.tp, (const int32_t *)w.src, w.rdiag, w.wa, (const QrxStep *)w.step, R, v.qtf,
                                       (const LmState *)st, (const int32_t *)w.plist);
                te(1, stream);
                if (j > 0) cur ^= 1;
            }
            tb(2, stream);
            hipLaunchKernelGGL(k_qrx_finish, dim3(nprob), dim3(256), 0, stream, m, n, ld, coff, tst, vst, cur, 1, (const double *)T, w, R, v,
                               wa4, scratch, x, st, factor, gtol);
            te(2, stream);
            return;
        }
    }
    // the pivot kernel's instance for a handful of problems (NORM2's general runs out of registers, at most two workgroups
    // per CU) also serves every launch whose workgroups are all resident at once anyway
    static const int few_env = [] { const char *e = getenv("NLH_QRX_FEW"); return e ? atoi(e) : -1; }();
    const int few_max = few_env >= 0 ? few_env : QRX_FEW_MAX;
    // rows up to which a BATCH takes the pivot kernel's 32-terms-per-lane instance (128 registers, four workgroups per CU;
    // a column of more than 2048 rows then goes through its reflector slot in memory and two NORM2 chunks).  Measured at
    // 2048 x 4096x256 with the instance for 4096 rows too: 3,458 / 3,428 against 3,458 / 3,500 LM it/s (same box, alternating
    // runs): no gain, the default stays 2048.  (Also built and measured: an
    // instance whose threads gather whole 64-byte sectors -- sixteen consecutive elements each, the column in registers
    // from the gather to the scaling, two NORM2 chunks out of registers -- at three / four workgroups per CU: 96 / 268
    // bytes of spills, 1.4 / 4.4 % SLOWER than the 64-term instance.)
    static const int piv32_env = [] { const char *e = getenv("NLH_QRX_PIV32_MAXM"); return e ? atoi(e) : -1; }();
    const int piv32_maxm = std::max(2048, piv32_env >= 0 ? piv32_env : QRX_PIV32_MAXM);
    bool prev_flushed = false;
    int cur = 0, np = 0, lo = 1;             // lo: first slot position that can still hold live data (step 0 moves physically)
    for (int j = 0; j < n; ++j) {
        const int nwin = (n + 1 - lo + 63) / 64;
        const long nwg = (long)nact * nwin;
        // waves per workgroup of the row-parallel pass (16: the wide form, k_qrx_pass_rpw), 0: one wave per window
        static const long rpwh_env = [] { const char *e = getenv("NLH_QRX_RPWH"); return e ? atol(e) : -1L; }();
        const long nwgh = (long)nact * ((n + 1 - lo + 31) / 32);
        // (NLH_QRX_RP8=<pairs>: the lane-quad form with EIGHT waves -- adder, stager, six producers, two workgroups per CU --
        // instead of the four-wave form up to that many pairs.  Measured, ms per solve off / up to 512: 96 x 4096x256 197.9 /
        // 192.3, 128: 218.1 / 231.5, 192: 321.8 / 317.8, 192 x 2048x128 55.6 / 53.5, 384: 94.8 / 90.9, 512: 104.8 / 104.3 --
        // no consistent gain where launches are HBM-bound anyway; left off.)
        static const long rp8_env = [] { const char *e = getenv("NLH_QRX_RP8"); return e ? atol(e) : 0L; }();
        const int rp = nwg <= rpw16_max ? (nwgh <= (rpwh_env >= 0 ? rpwh_env : (long)QRX_RPWH_MAX_WG) ? QRX_RP_HALF : 16)
                       : nwg <= rp_max ? (nwg <= rp8_env ? 8 : 4) : 0;
        // the wide form keeps at most three pending reflectors (scalar registers): a flush every 4th step; the four-wave
        // form every 8th; full launches every QRX_C-th
        const int period = forced_period ? forced_period : ((rp == 16 || rp == QRX_RP_HALF || rp == 8) ? QRX_RPW_MAXNP + 1 : rp == 4 ? (QRX_C < 8 ? QRX_C : 8) : QRX_C);
        const bool flush = qrx_can_flush(np) && np >= period - 1;
        tb(0, stream);
        const int pf = (flush ? 1 : 0) | (prev_flushed ? 2 : 0);
        // (Measured and dropped: the step as two lean launches -- a 102-register gather kernel, eight workgroups per CU,
        // and a NORM2 kernel with ONE WAVE per problem working out of registers, so that all 2048 chains of a batch are in
        // flight at once instead of 512 per round.  Bit-identical, but no faster: gather 98 us (HBM-bound: the column's
        // sectors and np reflector vectors, ~210 KB per problem) + norm 88 us (46 us loading and preparing coefficients
        // with a lane per 64-element run, 24 us the chains -- two per SIMD --, 24 us scaling: the column makes three more
        // trips through memory than in the fused kernel) against 191 us for this kernel's four rounds.)
        if (m <= 2048 && nact <= few_max)
            hipLaunchKernelGGL((k_qrx_pivot<32, false, true>), dim3(nprob), dim3(256), 0, stream, 0, m, n, ld, coff, tst, vst, j, cur, np, pf,
                               T, w, R, v, (const LmState *)st);
        else if (m <= piv32_maxm && (m <= 2048 || nact > few_max))
            hipLaunchKernelGGL(k_qrx_pivot<32>, dim3(nprob), dim3(256), 0, stream, 0, m, n, ld, coff, tst, vst, j, cur, np, pf,
                               T, w, R, v, (const LmState *)st);
        else if (m - j <= 64 * 64 && nact <= few_max)
            hipLaunchKernelGGL((k_qrx_pivot<64, false, true>), dim3(nprob), dim3(256), 0, stream, 0, m, n, ld, coff, tst, vst, j, cur, np, pf,
                               T, w, R, v, (const LmState *)st);
        else if (m - j > 64 * 64 && m - j <= 64 * QRX_LONG_EL * QRX_LONG_MAXCH)
            hipLaunchKernelGGL((k_qrx_pivot<64, true>), dim3(nprob), dim3(QRX_LONG_THREADS), 0, stream, 0, m, n, ld, coff, tst, vst, j, cur, np, pf,
                               T, w, R, v, (const LmState *)st);
        else
            hipLaunchKernelGGL(k_qrx_pivot<64>, dim3(nprob), dim3(256), 0, stream, 0, m, n, ld, coff, tst, vst, j, cur, np, pf,
                               T, w, R, v, (const LmState *)st);
        te(0, stream);
        tb(1, stream);
        dispatch_pass<0>(np, flush, rp, stream, 0, nprob, lo, m, n, ld, coff, tst, vst, j, cur, T, w, R, v.qtf, st);
        te(1, stream);
        prev_flushed = flush;
        if (flush) { cur ^= 1; np = 1; lo = j + 1; } else { np += 1; }
    }
    tb(2, stream);
    hipLaunchKernelGGL(k_qrx_finish, dim3(nprob), dim3(256), 0, stream, m, n, ld, coff, tst, vst, cur, np, (const double *)T, w, R, v,
                       wa4, scratch, x, st, factor, gtol);
    te(2, stream);
}
