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
//   * the working matrix T is ROW-major (the residual rides along as the last column), one lane owns one
//     trailing column and walks down the rows: a wave reads 64 consecutive doubles per row through a buffer
//     descriptor (scalar row offset + per-lane column offset, no address arithmetic), 32 rows of loads in
//     flight ahead of the arithmetic, one wave per workgroup, no barrier between waves;
//   * rows END on a 64-column boundary (qrx_coff), so the trailing columns of any step fill whole 64-column
//     windows counted from the end: every wave-level access is one aligned 512-byte span and a step launches
//     exactly ceil((n - j) / 64) waves per problem;
//   * reflector entries are wave-uniform: each wave stages a 64-row tile of them in LDS one tile ahead
//     (coalesced reads of the slot vectors) and reads a row's entries back as broadcast ds_reads, issued one
//     row pair ahead of the arithmetic; two rows are processed together so that the dependent mul / sub chains
//     of the pending updates interleave;
//   * column updates are DEFERRED: after step j a trailing column is not rewritten; its multiplier
//     t_k = s_k / a_jj is kept and later passes apply the pending updates on the fly, oldest first --
//     e = ((a - t_0 v_0) - t_1 v_1) ... -- the very roundings of the eager update (:655).  Every 7th step
//     the pass stores e back (flush, non-temporal): 8 B read + 8/7 B written per element and step instead of 24.
//     (Measured: 12 or 16 slots per bank are slower -- the extra multiply / subtract pairs cost more than the
//     flushes they save; a flushing pass runs at 4.8 TB/s, a plain one at 5.7-6.0 TB/s of the 6.3 achievable.)
//   * the column interchange (:626-637) never moves data: slot k of the permuted matrix carries a source
//     column index src[k]; the pivot column is consumed into the reflector and slot kmax simply inherits
//     slot j's source and pending multipliers.  The flush writes every slot to its own position.
// Launches per step: k_qrx_pivot (one workgroup per problem: pivot search, bookkeeping, gather of the pivot
// column with its pending updates, NORM2 -- a serial chain of m - j adds --, scaling -> reflector) and
// k_qrx_pass (lane per trailing column).  The driver keeps several sub-batches in flight on private streams
// (nlh_api.hip, lm_sub_batches) so that one sub-batch's pivot kernels run under another's passes.
#include "nlh_qrx.h"
#include "nlh_common.h"
#include <type_traits>
#include <cstdlib>

#ifndef QRX_C
#define QRX_C 8            // reflector slots per bank = pending updates before a flush + 1
#endif
#define QRX_TR 64           // rows per reflector tile of the pass
#define QRX_PAD_ROWS 160   // read-ahead padding behind the last problem's matrix (a tile + a load group)
#ifndef QRX_AUX_LOAD
#define QRX_AUX_LOAD 0    // cache policy of the matrix stream (2 = non-temporal)
#endif
#ifndef QRX_AUX_STORE
#define QRX_AUX_STORE 2   // the flush: non-temporal, the rewritten columns are not read again before the next step
#endif
#define QRX_NE 8           // NORM2 chunk: elements per thread

typedef unsigned int qrx_u32x2 __attribute__((ext_vector_type(2)));
struct QrxStep { double ajnorm, ajj; int32_t kmax, pad; };

// Physical column of slot k (k = 0 .. n, n = the residual) is k + qrx_coff(n): the row ENDS on a 64-column boundary, so
// that the trailing slots j+1 .. n of any step fill whole 64-column windows counted from the end: every wave-level load
// and store of the pass is one aligned 512-byte span, and the number of waves is exactly ceil((n - j) / 64).
static int qrx_coff(int n) { return (64 - ((n + 1) & 63)) & 63; }
int qrx_ld(int n) { return n + 1 + qrx_coff(n); }

static size_t qrx_tstride(int m, int n) { return (size_t)m * qrx_ld(n); }   // doubles between two problems' matrices
static size_t qrx_vstride(int m)      // doubles between two slots of a reflector bank
{
    return ((size_t)m + 7) & ~(size_t)7;
}

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
    if (w) {
        char *b = (char *)base;
        w->V = (double *)(b + oV); w->tp = (double *)(b + otp); w->rdiag = (double *)(b + ord);
        w->wa = (double *)(b + owa); w->step = (QrxStep *)(b + ost); w->src = (int32_t *)(b + osr);
        w->slotof = (int32_t *)(b + oso);
    }
    return off;
}

size_t qrx_workspace_bytes(int nprob, int m, int n) { return qrx_carve(nullptr, nprob, m, n, nullptr); }

// Column-major m x n  ->  row-major with row stride ld (32 x 32 tiles through LDS).
__global__ void __launch_bounds__(256)
k_qrx_transpose(int m, int n, int ld, int coff, size_t tst, const double *__restrict__ J, double *__restrict__ T,
                const LmState *__restrict__ st)
{
    __shared__ double tile[32][33];
    const int p = blockIdx.z;
    if (st && st[p].stage != ST_NEED_QR) return;
    const double *Jp = J + (size_t)p * m * n;
    double *Tp = T + (size_t)p * tst;
    const int i0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + tx, k = k0 + r;
        tile[r][tx] = (i < m && k < n) ? Jp[(size_t)k * m + i] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, k = k0 + tx;
        if (i < m && k < n) Tp[(size_t)i * ld + coff + k] = tile[tx][r];
    }
}

// wa4 = fvec as column n (:241), initial column norms (:611-616), identity maps.
__global__ void __launch_bounds__(256)
k_qrx_init(int m, int n, int ld, int coff, size_t tst, double *__restrict__ T, const double *__restrict__ fall, QrxWs w, LmVecs v,
           const LmState *__restrict__ st)
{
    const int p = blockIdx.x;
    if (st && st[p].stage != ST_NEED_QR) return;
    const int tid = threadIdx.x, BS = blockDim.x;
    double *a = T + (size_t)p * tst;
    const double *f = fall + (size_t)p * m;
    for (int i = tid; i < m; i += BS) a[(size_t)i * ld + coff + n] = f[i];
    for (int k = tid; k < n; k += BS) {
        const double nr = norm2_flang_serial_strided(a + coff + k, ld, m);
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
__global__ void __launch_bounds__(256)
k_qrx_pivot(int p0, int m, int n, int ld, int coff, size_t tst, size_t vst, int j, int cur, int np, int flush, double *__restrict__ T, QrxWs w,
            double *__restrict__ Rall, LmVecs v, const LmState *__restrict__ st)
{
    __shared__ double cd[2 * QRX_NE * 256];
    __shared__ double aux[40 + 128];
    __shared__ double red[64];
    const int p = p0 + blockIdx.x;
    if (st && st[p].stage != ST_NEED_QR) return;
    const int tid = threadIdx.x, BS = blockDim.x, ldp = n + 1;
    int *redi = reinterpret_cast<int *>(red + 32);
    double *rdiag = w.rdiag + (size_t)p * n, *wa = w.wa + (size_t)p * n;
    int32_t *src = w.src + (size_t)p * ldp;
    int32_t *ipvt = v.ipvt + (size_t)p * n;
    double *tpc = w.tp + ((size_t)p * 2 + cur) * QRX_C * ldp;
    double *R = Rall + (size_t)p * n * n;

    double bv = 0.0;
    int bk = 0x7fffffff;
    for (int k = j + tid; k < n; k += BS) {
        const double d = rdiag[k];
        if (bk == 0x7fffffff || d > bv) { bv = d; bk = k; }
    }
    const int kmax = block_argmax_first(bv, bk, red, redi);
    const int srck = src[kmax];
    int32_t *slotof = w.slotof + (size_t)p * ld;
    double tk[QRX_C];
#pragma unroll
    for (int q = 0; q < QRX_C; ++q) tk[q] = (q < np) ? tpc[(size_t)q * ldp + kmax] : 0.0;
    __syncthreads();
    if (kmax != j) {
        if (tid == 0) {
            rdiag[kmax] = rdiag[j];
            wa[kmax] = wa[j];
            const int32_t t = ipvt[j]; ipvt[j] = ipvt[kmax]; ipvt[kmax] = t;
            if (j == 0) {
                // first step only: a physical interchange (slot 0's column is copied over the consumed pivot column in the
                // gather below), so that the live columns are coff + 1 .. from the start -- with n + 1 = 1 (mod 64) a
                // live column at coff + 0 would cost every pass of the first cycle a whole extra 64-column window
                slotof[src[0]] = -1;
                slotof[srck] = kmax;
            } else {
                src[kmax] = src[j];                              // slot kmax now lives where slot j's data is
                slotof[src[j]] = kmax;
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
    const int src0 = coff;                                       // physical column of slot 0 at step 0
    // The pivot column with its pending updates applied, oldest first.  Four rows per thread are loaded together (the
    // column walk costs a 64-byte sector per element, the pending reflector entries are coalesced) before any is stored.
    const double *__restrict__ Vc = w.V + ((size_t)p * 2 + cur) * QRX_C * vst;          // slot q at Vc + q * vst
    double *__restrict__ Vn = flush ? w.V + ((size_t)p * 2 + (cur ^ 1)) * QRX_C * vst
                                    : w.V + (((size_t)p * 2 + cur) * QRX_C + np) * vst;
    double *col = T + (size_t)p * tst + srck;
    for (int i0 = j + tid; i0 < m; i0 += 4 * BS) {
        double e[4], vq[4][QRX_C - 1], mv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = min(i0 + u * BS, m - 1);
            e[u] = col[(size_t)row * ld];
            mv[u] = move0 ? col[(size_t)row * ld + (src0 - srck)] : 0.0;
#pragma unroll
            for (int q = 0; q < QRX_C - 1; ++q) vq[u][q] = (q < np) ? Vc[(size_t)q * vst + row] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int q = 0; q < QRX_C - 1; ++q)
                if (q < np) e[u] = e[u] - tk[q] * vq[u][q];
            if (i0 + u * BS < m) {
                Vn[i0 + u * BS] = e[u];
                if (move0) col[(size_t)(i0 + u * BS) * ld] = mv[u];
            }
        }
    }
    __syncthreads();
    double ajnorm = norm2_flang_block_wide<QRX_NE>([&](int i) { return Vn[j + i]; }, m - j, cd, QRX_NE * BS, aux);   // :642
    double ajj = 0.0;
    if (ajnorm != 0.0) {
        if (Vn[j] < 0.0) ajnorm = -ajnorm;                       // :644
        __syncthreads();
        for (int i = j + tid; i < m; i += BS) {                   // :645-646
            double t = Vn[i] / ajnorm;
            if (i == j) t = t + 1.0;
            Vn[i] = t;
        }
        __syncthreads();
        ajj = Vn[j];
    }
    if (tid == 0) {
        QrxStep s;
        s.ajnorm = ajnorm; s.ajj = ajj; s.kmax = kmax; s.pad = 0;
        w.step[p] = s;
        rdiag[j] = -ajnorm;                                      // :665
    }
}

// Step j, part 2: every trailing column k = j+1 .. n (n = the residual): pending updates, dot product with
// the reflector in ascending row order (:652-653), multiplier (:654), row j becomes final (R(j,k) / qtf(j)),
// norm down-date (:656-661).
// One wave per workgroup; a lane owns CPT columns (lane, lane + 64, ...) and walks down the rows, so a wave reads
// 64 consecutive doubles per row and column group, with 32 loads per lane in flight ahead of the arithmetic.
// The reflector entries of a row (wave-uniform) come from an LDS tile of 64 rows that the wave stages for itself one
// tile ahead (coalesced 64-byte rows -> broadcast ds_reads); with CPT = 4 a row's nine LDS values serve 256 elements.
template <int NP, bool FLUSH, int CPT>
__global__ void __launch_bounds__(64)
k_qrx_pass(int p0, int nprob, int nwin, int lo, int m, int n, int ld, int coff, size_t tst, size_t vst, int j, int cur,
           double *__restrict__ T, const double *__restrict__ Vall, double *__restrict__ tpall,
           int32_t *__restrict__ srcall, int32_t *__restrict__ slotall, double *__restrict__ rdall,
           double *__restrict__ waall, const QrxStep *__restrict__ stepall, double *__restrict__ Rall,
           double *__restrict__ qtfall, const LmState *__restrict__ st)
{
    constexpr int U = 32 / CPT;            // rows per load group
    constexpr int TR = QRX_TR;             // rows per reflector tile
    constexpr int LP = QRX_C;              // LDS row: the pending entries and the new one (NP + 1 <= QRX_C doubles)
    constexpr int NPI = NP < QRX_C ? NP : 0;
    __shared__ double vt[2][TR * LP];
    // Workgroup -> (problem, window): consecutive workgroup ids go to consecutive XCDs, so the windows of one problem
    // are given ids that agree modulo 8: they share an L2 (reflector tiles, multipliers, the row segment two windows
    // both touch are fetched from the fabric once).
    const int b_ = blockIdx.x, grp = b_ / (8 * nwin), r_ = b_ % (8 * nwin);
    const int pl = grp * 8 + (r_ & 7), win = r_ >> 3;
    if (pl >= nprob) return;
    const int p = p0 + pl;
    if (st && st[p].stage != ST_NEED_QR) return;
    const int lane = threadIdx.x, ldp = n + 1;
    const int wtop = ld - 64 * CPT * win;                               // end (exclusive) of this wave's topmost window
    const QrxStep step = stepall[p];
    const bool refl = step.ajnorm != 0.0;
    const double ajj = step.ajj;
    int32_t *srcp = srcall + (size_t)p * ldp;
    int32_t *slotp = slotall + (size_t)p * ld;
    double *tpc = tpall + ((size_t)p * 2 + cur) * QRX_C * ldp;
    const double *vc = Vall + ((size_t)p * 2 + cur) * QRX_C * vst + j;          // slot q, row j + r at vc[q * vst + r]
    const double *vo = Vall + ((size_t)p * 2 + (cur ^ 1)) * QRX_C * vst + j;   // slot 0 of the other bank
    double *Tj = T + (size_t)p * tst + (size_t)j * ld;                  // row j
    const int nrows = m - j;

    // A lane owns a PHYSICAL column; which slot of the permuted matrix that column currently holds comes from the
    // inverse map (the interchange never moves data).  Since the last flush the live columns are coff + lo .. ld - 1
    // (lo = the step after that flush, 0 before the first one) minus the consumed ones, at most QRX_C - 1 of them,
    // whose lanes idle: every load is the lane's own column, i.e. one aligned 512-byte span per wave and row.
    unsigned kc[CPT], pc[CPT];
    bool act[CPT];
    double tq[CPT][NP > 0 ? NP : 1], s[CPT], rowj[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int col = wtop - 64 * (c + 1) + lane;
        const int k = (col >= coff + lo) ? slotp[col] : -1;
        act[c] = k > j;
        pc[c] = (unsigned)col;
        kc[c] = act[c] ? (unsigned)k : (unsigned)n;                     // idle lanes: any valid slot for the table reads
#pragma unroll
        for (int q = 0; q < NP; ++q) tq[c][q] = tpc[(size_t)q * ldp + kc[c]];
        s[c] = 0.0;
    }
    auto pending = [&](double a, int c, const auto &vr) {
        double e = a;
#pragma unroll
        for (int q = 0; q < NP; ++q) e = e - tq[c][q] * vr[q];
        return e;
    };
    {   // row j with its pending updates (becomes final below)
        double v0[NP + 1];
#pragma unroll
        for (int q = 0; q < NP; ++q) v0[q] = vc[(size_t)q * vst];
#pragma unroll
        for (int c = 0; c < CPT; ++c) rowj[c] = pending(Tj[pc[c]], c, v0);
    }

    // reflector tile t: lane l fetches the entries of row t*TR + l (one coalesced 512-byte read per slot); the staged
    // LDS row is [pending v_0 .. v_NP-1, new v]
    double sv[NP + 1];
    auto vfetch = [&](int t) {
        const int row = t * TR + lane;
#pragma unroll
        for (int q = 0; q < NP; ++q) sv[q] = vc[(size_t)q * vst + row];
        sv[NP] = FLUSH ? vo[row] : vc[(size_t)NPI * vst + row];
    };
    auto vstore = [&](int buf) {
        double *d = &vt[buf][lane * LP];
#pragma unroll
        for (int q = 0; q <= NP; ++q) d[q] = sv[q];
    };
    // Matrix accesses go through a buffer descriptor of this problem's matrix: address = descriptor base + wave-uniform
    // row offset (scalar register) + per-lane column offset (one VGPR per column for the whole pass); reads past the
    // last row return zero and are never used.
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(Tj, 0, (int)((size_t)nrows * ld * sizeof(double)), 0x00020000);
    unsigned so[CPT], ko[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        so[c] = act[c] ? pc[c] * 8u : 0x80000000u;      // idle lanes: past the descriptor's range, nothing is fetched
        ko[c] = (kc[c] + (unsigned)coff) * 8u;
    }
    const unsigned ldb = (unsigned)ld * 8u;
    double a0[U][CPT], a1[U][CPT];
    auto load = [&](double (&buf)[U][CPT], int r0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned roff = (unsigned)(r0 + u) * ldb;
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                const qrx_u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, so[c], roff, QRX_AUX_LOAD);
                buf[u][c] = __hiloint2double((int)w.y, (int)w.x);
            }
        }
    };
    auto compute = [&](auto guarded, const double (&buf)[U][CPT], int r0, const double *tile, int g) {
        constexpr bool GD = decltype(guarded)::value;
        double va[NP + 1], vb[NP + 1];
        auto ldsrow = [&](double (&dst)[NP + 1], int u) {
            const double *vr = tile + (g * U + u) * LP;
#pragma unroll
            for (int q = 0; q <= NP; ++q) dst[q] = vr[q];
        };
        auto flushrow = [&](const double (&e)[CPT], int row) {
            const unsigned roff = (unsigned)row * ldb;
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                if (act[c]) {
                    qrx_u32x2 w;
                    w.x = (unsigned)__double2loint(e[c]); w.y = (unsigned)__double2hiint(e[c]);
                    __builtin_amdgcn_raw_buffer_store_b64(w, rsrc, ko[c], roff, QRX_AUX_STORE);
                }
            }
        };
        // Two rows at a time with the 2 * CPT update chains interleaved (each chain is a dependent mul / sub
        // sequence, one wave per SIMD has nobody else to hide that latency); the LDS reads of a row pair are
        // issued one pair ahead.
        ldsrow(va, 0);
        ldsrow(vb, 1);
#pragma unroll
        for (int u = 0; u < U; u += 2) {
            if (GD && r0 + u >= nrows) break;                           // uniform
            double e0[CPT], e1[CPT], v0[NP + 1], v1[NP + 1];
#pragma unroll
            for (int q = 0; q <= NP; ++q) { v0[q] = va[q]; v1[q] = vb[q]; }
            if (u + 2 < U) { ldsrow(va, u + 2); ldsrow(vb, u + 3); }
#pragma unroll
            for (int c = 0; c < CPT; ++c) { e0[c] = buf[u][c]; e1[c] = buf[u + 1][c]; }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                double p0[CPT], p1[CPT];
#pragma unroll
                for (int c = 0; c < CPT; ++c) { p0[c] = tq[c][q] * v0[q]; p1[c] = tq[c][q] * v1[q]; }
#pragma unroll
                for (int c = 0; c < CPT; ++c) { e0[c] = e0[c] - p0[c]; e1[c] = e1[c] - p1[c]; }
            }
            double w0[CPT], w1[CPT];
#pragma unroll
            for (int c = 0; c < CPT; ++c) { w0[c] = v0[NP] * e0[c]; w1[c] = v1[NP] * e1[c]; }
#pragma unroll
            for (int c = 0; c < CPT; ++c) s[c] = s[c] + w0[c];           // :653, rows ascending
            if (FLUSH) flushrow(e0, r0 + u);
            if (GD && r0 + u + 1 >= nrows) break;
#pragma unroll
            for (int c = 0; c < CPT; ++c) s[c] = s[c] + w1[c];
            if (FLUSH) flushrow(e1, r0 + u + 1);
        }
    };
    const int ntile = (nrows + TR - 1) / TR, nfull = nrows / TR;
    // Unconditional read-ahead of the reflector rows: rows past the last one belong to the next bank or to the padding
    // behind the banks and are never used.
    vfetch(0);
    vstore(0);
    load(a0, 0);
    __syncthreads();
    std::false_type plain_t; std::true_type guard_t;
    for (int t = 0; t < ntile; ++t) {
        const double *tile = vt[t & 1];
        const int rb = t * TR;
        vfetch(t + 1);
        if (t < nfull) {
#pragma unroll 1
            for (int g = 0; g < TR / U; g += 2) {
                load(a1, rb + (g + 1) * U);
                compute(plain_t, a0, rb + g * U, tile, g);
                load(a0, rb + (g + 2) * U);
                compute(plain_t, a1, rb + (g + 1) * U, tile, g + 1);
            }
        } else {
#pragma unroll 1
            for (int g = 0; g < TR / U; g += 2) {
                load(a1, rb + (g + 1) * U);
                compute(guard_t, a0, rb + g * U, tile, g);
                load(a0, rb + (g + 2) * U);
                compute(guard_t, a1, rb + (g + 1) * U, tile, g + 1);
            }
        }
        vstore((t + 1) & 1);
        __syncthreads();
    }

    double *tpn = FLUSH ? tpall + ((size_t)p * 2 + (cur ^ 1)) * QRX_C * ldp : tpc + (size_t)NPI * ldp;
    double *rdiag = rdall + (size_t)p * n, *wa = waall + (size_t)p * n;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        if (!act[c]) continue;
        const int k = (int)kc[c];
        const double temp = refl ? s[c] / ajj : 0.0;                    // :654 (residual: see nlh_kernels_exact.h)
        tpn[k] = temp;
        if (FLUSH) {                                                    // the column now sits at its slot's own position
            srcp[k] = coff + k;
            slotp[coff + k] = k;
            if ((int)pc[c] != coff + k) slotp[pc[c]] = -1;
        }
        const double rjk = refl ? rowj[c] - temp * ajj : rowj[c];       // :655 at i = j: row j is final
        if (k == n) { qtfall[(size_t)p * n + j] = rjk; continue; }
        Rall[(size_t)p * n * n + (size_t)k * n + j] = rjk;
        if (!refl) continue;
        double rk = rdiag[k];
        if (rk != 0.0) {                                                // :656-661
            const double t2 = rjk / rk;
            rk = rk * sqrt(fmax(0.0, 1.0 - t2 * t2));
            const double q = rk / wa[k];
            if (!(5.0e-2 * (q * q) > NLH_EPS)) {
                const double *colp = Tj + pc[c];
                const double *dstp = Tj + coff + kc[c];
                rk = norm2_flang_serial([&](int i2) {
                    const int row = 1 + i2;
                    double vr[NP + 1];
#pragma unroll
                    for (int q = 0; q < NP; ++q) vr[q] = vc[(size_t)q * vst + row];
                    // a flush has just rewritten the column (pending updates applied) at its own position
                    const double e = FLUSH ? dstp[(size_t)row * ld] : pending(colp[(size_t)row * ld], c, vr);
                    const double vn = FLUSH ? vo[row] : vc[(size_t)NPI * vst + row];
                    return e - temp * vn;
                }, nrows - 1);
                wa[k] = rk;
            }
            rdiag[k] = rk;
        }
    }
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
            e = a[(size_t)i * ld + coff + n];
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

template <int NP, bool FLUSH>
static void launch_pass(hipStream_t stream, int p0, int nprob, int lo, int m, int n, int ld, int coff, size_t tst, size_t vst, int j, int cur, double *T, const QrxWs &w,
                        double *R, double *qtf, const LmState *st)
{
    // One column per lane (CPT = 1): measured against two and four columns per lane (fewer waves, the LDS row shared by
    // more elements) on 512 x 4096x256, 1024 x 2048x128 and a single problem, more waves won every time.
    constexpr int CPT = 1;
    const int nwin = (n + 1 - lo + 64 * CPT - 1) / (64 * CPT);         // live physical columns coff + lo .. coff + n
    const dim3 grid((unsigned)(((nprob + 7) / 8) * 8 * nwin));
    hipLaunchKernelGGL((k_qrx_pass<NP, FLUSH, CPT>), grid, dim3(64), 0, stream, p0, nprob, nwin, lo, m, n, ld, coff, tst, vst, j, cur,
                       T, (const double *)w.V, w.tp, w.src, w.slotof, w.rdiag, w.wa, (const QrxStep *)w.step, R, qtf, st);
}

// np = 0 .. QRX_C - 2: plain pass with np pending updates; np = QRX_C - 1: the flushing pass.
template <int NP>
static void dispatch_pass(int np, hipStream_t stream, int p0, int nprob, int lo, int m, int n, int ld, int coff, size_t tst, size_t vst,
                          int j, int cur, double *T, const QrxWs &w, double *R, double *qtf, const LmState *st)
{
    if constexpr (NP == QRX_C - 1) {
        launch_pass<NP, true>(stream, p0, nprob, lo, m, n, ld, coff, tst, vst, j, cur, T, w, R, qtf, st);
    } else {
        if (np == NP) launch_pass<NP, false>(stream, p0, nprob, lo, m, n, ld, coff, tst, vst, j, cur, T, w, R, qtf, st);
        else dispatch_pass<NP + 1>(np, stream, p0, nprob, lo, m, n, ld, coff, tst, vst, j, cur, T, w, R, qtf, st);
    }
}

void qrx_factor(hipStream_t stream, int nprob, int m, int n, const double *J, double *T, const double *fvec,
                double *R, LmVecs v, double *wa4, double *scratch, const double *x, LmState *st, double factor,
                double gtol, void *ws, const QrxTimer *tm)
{
    QrxWs w;
    qrx_carve(ws, nprob, m, n, &w);
    const int ld = qrx_ld(n), coff = qrx_coff(n);
    const size_t tst = qrx_tstride(m, n), vst = qrx_vstride(m);
    auto tb = [&](int which, hipStream_t s) { if (tm) tm->begin(tm->ctx, which, s); };
    auto te = [&](int which, hipStream_t s) { if (tm) tm->end(tm->ctx, which, s); };
    tb(2, stream);
    hipLaunchKernelGGL(k_qrx_transpose, dim3((m + 31) / 32, (n + 31) / 32, nprob), dim3(256), 0, stream, m, n, ld, coff, tst, J, T,
                       (const LmState *)st);
    hipLaunchKernelGGL(k_qrx_init, dim3(nprob), dim3(256), 0, stream, m, n, ld, coff, tst, T, fvec, w, v, (const LmState *)st);
    te(2, stream);
    // (Measured and dropped: the two halves of the batch on two streams, half B's pivot kernel under half A's pass, with
    // events keeping the passes from overlapping each other -- 1035 ms instead of 999 ms per 512 x 4096x256 solve; the
    // cross-stream event waits cost more than the pivot latency they hide.  Sub-batches on host threads, which need no
    // cross-stream ordering, do hide it: nlh_api.hip, lm_sub_batches.)
    int cur = 0, np = 0, lo = 1;             // lo: first slot position that can still hold live data (step 0 moves physically)
    for (int j = 0; j < n; ++j) {
        const bool flush = (np == QRX_C - 1);
        tb(0, stream);
        hipLaunchKernelGGL(k_qrx_pivot, dim3(nprob), dim3(256), 0, stream, 0, m, n, ld, coff, tst, vst, j, cur, np, flush ? 1 : 0,
                           T, w, R, v, (const LmState *)st);
        te(0, stream);
        tb(1, stream);
        dispatch_pass<0>(np, stream, 0, nprob, lo, m, n, ld, coff, tst, vst, j, cur, T, w, R, v.qtf, st);
        te(1, stream);
        if (flush) { cur ^= 1; np = 1; lo = j + 1; } else { np += 1; }
    }
    tb(2, stream);
    hipLaunchKernelGGL(k_qrx_finish, dim3(nprob), dim3(256), 0, stream, m, n, ld, coff, tst, vst, cur, np, (const double *)T, w, R, v,
                       wa4, scratch, x, st, factor, gtol);
    te(2, stream);
}
