// nlh_kernels_exact.h -- lmfactor + Q^T f in the reference's operation order
// (policy NLH_FACTOR_EXACT): bit-identical to the CPU path.
//
// What can be parallel without changing a single rounding: the trailing columns of a
// Householder step are independent (one thread per column, each forming its dot product
// and axpy over the rows in ascending order exactly as src/nonlin_least_squares.f90:652-655),
// every elementwise update, the pivot search, and -- for NORM2 -- the divisions (see
// norm2_flang_block).  What stays serial: the row recurrence inside each dot product.
//
// The Jacobian is processed ROW-major (Jt[i*n + k]) so that the per-column threads of a
// wave read consecutive addresses; k_transpose produces it from the column-major FD result.
// The residual is carried as column index n: applying reflector j to it during step j is
// the same arithmetic as the reference's later Q^T f sweep (:241-253), because column j is
// final after step j and w + v*(-s/a) == w - (s/a)*v bit for bit.
#pragma once
#include "nlh_common.h"
#include "nlh_kernels_factor.h"

// Column-major m-by-n  ->  row-major m-by-n with row stride ld, 32x32 tiles through LDS.
__global__ void __launch_bounds__(256)
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
__global__ void __launch_bounds__(256)
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

#define QX_VC 2048     // rows of the reflector staged in LDS at a time
#define QX_PF 16       // loads each column thread keeps in flight

// Dynamic LDS: (2n + 64) doubles + 3*NLH_NCH + 8 + QX_VC.
__global__ void __launch_bounds__(1024)
k_qr_exact(int m, int n, double *__restrict__ Jt_all, const double *__restrict__ fall,
           double *__restrict__ Rall, LmVecs v, double *__restrict__ wa4all,
           double *__restrict__ scratch_all, const double *__restrict__ xall,
           LmState *__restrict__ st, double factor, double gtol, int standalone)
{
    extern __shared__ double smem[];
    const int p = blockIdx.x;
    LmState *s = st ? st + p : nullptr;
    if (s && s->stage != ST_NEED_QR) return;
    const int tid = threadIdx.x, BS = blockDim.x;
    double *rdiag = smem;           // n
    double *wa = smem + n;          // n
    double *red = smem + 2 * n;     // 64
    int *redi = reinterpret_cast<int *>(red + 32);
    double *scratch = red + 64;     // 3*NLH_NCH + 8
    double *vcol = scratch + 3 * NLH_NCH + 8;   // QX_VC
    double *a = Jt_all + (size_t)p * m * n;
    int32_t *ipvt = v.ipvt + (size_t)p * n;
    double *acnorm = v.acnorm + (size_t)p * n;
    double *qtf = v.qtf + (size_t)p * n;
    const int minmn = m < n ? m : n;
    const double p05 = 5.0e-2;

    // wa4 = fvec (:241); a fallback after a rejected trial must leave the caller's wa4 alone
    const bool first = (!s) || (s->inner_pass == 0);
    double *w4 = first ? (wa4all + (size_t)p * m) : (scratch_all + (size_t)p * m);
    const double *f = fall + (size_t)p * m;
    for (int i = tid; i < m; i += BS) w4[i] = f[i];

    // initial column norms (:611-616): one thread per column, reference-order NORM2
    for (int k = tid; k < n; k += BS) {
        const double nr = norm2_flang_serial_strided(a + k, n, m);
        acnorm[k] = nr; rdiag[k] = nr; wa[k] = nr; ipvt[k] = k;
    }
    __syncthreads();

    for (int j = 0; j < minmn; ++j) {
        double bv = 0.0;                                        // pivot (:622-637)
        int bk = 0x7fffffff;
        for (int k = j + tid; k < n; k += BS) {
            const double d = rdiag[k];
            if (bk == 0x7fffffff || d > bv) { bv = d; bk = k; }
        }
        const int kmax = block_argmax_first(bv, bk, red, redi);
        if (kmax != j) {
            for (int i = tid; i < m; i += BS) {
                double *row = a + (size_t)i * n;
                const double t = row[j]; row[j] = row[kmax]; row[kmax] = t;
            }
            if (tid == 0) {
                rdiag[kmax] = rdiag[j];
                wa[kmax] = wa[j];
                int32_t t = ipvt[j]; ipvt[j] = ipvt[kmax]; ipvt[kmax] = t;
            }
            __syncthreads();
        }
        // reflector (:642-646)
        double ajnorm = norm2_flang_block([&](int i) { return a[(size_t)(j + i) * n + j]; }, m - j, scratch);
        if (ajnorm != 0.0) {
            if (a[(size_t)j * n + j] < 0.0) ajnorm = -ajnorm;
            __syncthreads();
            for (int i = j + tid; i < m; i += BS) {
                double t = a[(size_t)i * n + j] / ajnorm;
                if (i == j) t = t + 1.0;
                a[(size_t)i * n + j] = t;
            }
            __syncthreads();
            const double ajj = a[(size_t)j * n + j];
            // trailing columns (:652-662) and the residual (column index n, :241-253): one thread per
            // column.  The reflector is staged through LDS in chunks of QX_VC rows (shared by all
            // columns); each thread keeps QX_PF of its own loads in flight ahead of the serial recurrence.
            for (int kbase = j + 1; kbase <= n; kbase += BS) {
                const int k = kbase + tid;
                const bool act = k <= n, isf = (k == n);
                double *ck = isf ? w4 : (a + (act ? k : 0));
                const size_t sk = isf ? 1 : (size_t)n;
                double sm = 0.0;
                for (int c0 = j; c0 < m; c0 += QX_VC) {             // pass 1: dot product, rows ascending
                    const int cl = min(QX_VC, m - c0);
                    __syncthreads();
                    for (int i = tid; i < cl; i += BS) vcol[i] = a[(size_t)(c0 + i) * n + j];
                    __syncthreads();
                    if (act) {
                        const double *cp = ck + (size_t)c0 * sk;
                        int i = 0;
                        for (; i + QX_PF <= cl; i += QX_PF) {
                            double av[QX_PF];
#pragma unroll
                            for (int u = 0; u < QX_PF; ++u) av[u] = cp[(size_t)(i + u) * sk];
#pragma unroll
                            for (int u = 0; u < QX_PF; ++u) sm = sm + vcol[i + u] * av[u];
                        }
                        for (; i < cl; ++i) sm = sm + vcol[i] * cp[(size_t)i * sk];
                    }
                }
                const double temp = isf ? (-sm / ajj) : (sm / ajj);     // :654 / :248
                for (int c0 = j; c0 < m; c0 += QX_VC) {             // pass 2: axpy
                    const int cl = min(QX_VC, m - c0);
                    __syncthreads();
                    for (int i = tid; i < cl; i += BS) vcol[i] = a[(size_t)(c0 + i) * n + j];
                    __syncthreads();
                    if (act) {
                        double *cp = ck + (size_t)c0 * sk;
                        int i = 0;
                        for (; i + QX_PF <= cl; i += QX_PF) {
                            double av[QX_PF];
#pragma unroll
                            for (int u = 0; u < QX_PF; ++u) av[u] = cp[(size_t)(i + u) * sk];
                            if (isf) {
#pragma unroll
                                for (int u = 0; u < QX_PF; ++u) cp[(size_t)(i + u) * sk] = av[u] + vcol[i + u] * temp;
                            } else {
#pragma unroll
                                for (int u = 0; u < QX_PF; ++u) cp[(size_t)(i + u) * sk] = av[u] - temp * vcol[i + u];
                            }
                        }
                        for (; i < cl; ++i) {
                            if (isf) cp[(size_t)i * sk] = cp[(size_t)i * sk] + vcol[i] * temp;
                            else cp[(size_t)i * sk] = cp[(size_t)i * sk] - temp * vcol[i];
                        }
                    }
                }
                if (act && !isf) {
                    double rk = rdiag[k];
                    if (rk != 0.0) {
                        const double t2 = a[(size_t)j * n + k] / rk;
                        rk = rk * sqrt(fmax(0.0, 1.0 - t2 * t2));
                        const double q = rk / wa[k];
                        if (!(p05 * (q * q) > NLH_EPS)) {
                            rk = norm2_flang_serial([&](int i2) { return a[(size_t)(j + 1 + i2) * n + k]; }, m - j - 1);
                            wa[k] = rk;
                        }
                        rdiag[k] = rk;
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0) { rdiag[j] = -ajnorm; qtf[j] = w4[j]; }
        __syncthreads();
    }
    for (int j = minmn + tid; j < n; j += BS) qtf[j] = 0.0;      // n <= m always holds for LM

    // R for lmpar: strict upper from the factored rows, diagonal = rdiag (:251)
    double *R = Rall + (size_t)p * n * n;
    for (int e = tid; e < n * n; e += BS) {
        const int i = e % n, c = e / n;
        if (i < c && i < m) R[e] = a[(size_t)i * n + c];
        else if (i == c) R[e] = rdiag[i];
    }
    for (int k = tid; k < n; k += BS) v.rdiag[(size_t)p * n + k] = rdiag[k];
    __syncthreads();
    if (standalone || !s) return;
    if (tid == 0) { s->factor_kind = 1; s->qr_count += 1; }
    if (first) {
        lm_head<true>(n, R, n, ipvt, acnorm, qtf, xall + (size_t)p * n, v.diag + (size_t)p * n,
                      v.diag_prev + (size_t)p * n, s, factor, gtol, ST_QR_READY, red, scratch);
    } else {
        if (tid == 0) s->stage = ST_QR_READY;
    }
}

// ---------------------------------------------------------------------------------------------
// The same factorisation with DEFERRED column updates (what the solver uses for n + 1 <= blockDim).
//
// k_qr_exact above touches every trailing element three times per Householder step (dot pass,
// axpy pass = read + write) and has only the n - j column threads issuing loads, so one step
// costs two exposed memory latencies per 16 rows.  Here
//   * the working matrix is row-major with the residual as column n (row stride ld = n + 1);
//   * a trailing column is NOT rewritten after a step: its multiplier temp_k = s_k / a_jj is kept
//     (tp, LDS) together with the reflector (V, global), and the next step applies the pending
//     updates on the fly, oldest first -- e = ((a - t_0 v_0) - t_1 v_1) ... -- which is the very
//     sequence of roundings the eager update performs.  Every B-1 steps the pass stores e back
//     (flush), so a step reads 8 B per element and writes 8/(B-1) B instead of 24 B;
//   * all threads of the workgroup load, update and multiply (elementwise = order-free); the
//     products go to an LDS tile and ONE thread per column adds them in ascending row order, which
//     is the only serial part of the reference's dot product (:652-653).  Tiles are double-buffered
//     (one barrier each) and three tiles of loads are in flight per thread;
//   * the interchange moves only the displaced column j into slot kmax (the pivot column is
//     consumed into V), and R(j, k) / qtf(j) are written as each row becomes final.
// The residual uses the same formula: w + v*(-s/a) == w - (s/a)*v bit for bit.
#define QL_DEPTH 3      // tiles of loads in flight
#define QX_B 4          // reflector ring: a column is rewritten every QX_B - 1 steps

// MAXT = largest workgroup (register budget: 512 threads -> 256 VGPRs); QL_RPT = rows per thread and tile.
template <int B, int MAXT> struct QlLds {
    static constexpr int RPT = 4096 / MAXT;
    // doubles of dynamic LDS for an n-column problem and bs threads
    static __host__ __device__ size_t doubles(int n, int bs)
    {
        return (size_t)2 * n + 64 + 3 * NLH_NCH + 8 + (size_t)B * (n + 1) + bs + 2 * (size_t)B * (bs / 64) * RPT
               + 2 * (size_t)RPT * bs;
    }
};

template <int B, int MAXT>
__global__ void __launch_bounds__(MAXT)
k_qr_exact_lazy(int m, int n, double *__restrict__ At_all, const double *__restrict__ fall,
                double *__restrict__ Rall, LmVecs v, double *__restrict__ wa4all,
                double *__restrict__ scratch_all, const double *__restrict__ xall,
                LmState *__restrict__ st, double factor, double gtol, double *__restrict__ Vall, int stream_nt)
{
    extern __shared__ double smem[];
    const int p = blockIdx.x;
    LmState *s = st ? st + p : nullptr;
    if (s && s->stage != ST_NEED_QR) return;
    constexpr int QL_RPT = QlLds<B, MAXT>::RPT;
    const int tid = threadIdx.x, BS = blockDim.x;
    const int ld = n + 1, TRMAX = (BS >> 6) * QL_RPT;
    double *rdiag = smem;                          // n
    double *wa = smem + n;                         // n
    double *red = smem + 2 * n;                    // 64
    int *redi = reinterpret_cast<int *>(red + 32);
    double *scratch = red + 64;                    // 3*NLH_NCH + 8
    double *tp = scratch + 3 * NLH_NCH + 8;        // B x (n+1): pending multipliers, by slot
    double *rowj = tp + (size_t)B * ld;            // BS: row j of the trailing columns, pending updates applied
    double *vt = rowj + BS;                        // 2 x B x TRMAX: reflector rows of the current tile
    double *tile = vt + 2 * (size_t)B * TRMAX;     // 2 x QL_RPT x BS products
    double *a = At_all + (size_t)p * m * ld;
    double *Vg = Vall + (size_t)p * B * m;
    double *R = Rall + (size_t)p * n * n;
    int32_t *ipvt = v.ipvt + (size_t)p * n;
    double *acnorm = v.acnorm + (size_t)p * n;
    double *qtf = v.qtf + (size_t)p * n;
    const double p05 = 5.0e-2;

    const bool first = (!s) || (s->inner_pass == 0);
    double *w4 = first ? (wa4all + (size_t)p * m) : (scratch_all + (size_t)p * m);
    const double *f = fall + (size_t)p * m;
    for (int i = tid; i < m; i += BS) a[(size_t)i * ld + n] = f[i];          // wa4 = fvec (:241)
    for (int k = tid; k < n; k += BS) {                                       // :611-616
        const double nr = norm2_flang_serial_strided(a + k, ld, m);
        acnorm[k] = nr; rdiag[k] = nr; wa[k] = nr; ipvt[k] = k;
    }
    __syncthreads();

    int s0 = 0, np = 0;                             // pending reflectors live in slots (s0 + q) & (B-1), q < np
#ifdef QX_PROFILE
    unsigned long long tc[6] = {0, 0, 0, 0, 0, 0}, tl0 = wall_clock64(), tstart = tl0;
    unsigned long long tc2[3] = {0, 0, 0}, tl2 = 0;
#define QX_TICK(x) { unsigned long long t_ = wall_clock64(); tc[x] += t_ - tl0; tl0 = t_; tl2 = t_; }
#define QX_TICK2(x) { unsigned long long t_ = wall_clock64(); tc2[x] += t_ - tl2; tl2 = t_; }
#else
#define QX_TICK(x)
#define QX_TICK2(x)
#endif
    for (int j = 0; j < n; ++j) {
        QX_TICK(5)
        double bv = 0.0;                                                      // pivot (:622-637)
        int bk = 0x7fffffff;
        for (int k = j + tid; k < n; k += BS) {
            const double d = rdiag[k];
            if (bk == 0x7fffffff || d > bv) { bv = d; bk = k; }
        }
        const int kmax = block_argmax_first(bv, bk, red, redi);
        const int snew = (s0 + np) & (B - 1);
        double *Vn = Vg + (size_t)snew * m;
        double tk[B];
#pragma unroll
        for (int q = 0; q < B; ++q) tk[q] = (q < np) ? tp[(size_t)((s0 + q) & (B - 1)) * ld + kmax] : 0.0;
        __syncthreads();
        if (kmax != j) {
            if (tid == 0) {
                rdiag[kmax] = rdiag[j];
                wa[kmax] = wa[j];
                int32_t t = ipvt[j]; ipvt[j] = ipvt[kmax]; ipvt[kmax] = t;
            }
            if (tid < np) {
                const int sl = (s0 + tid) & (B - 1);
                tp[(size_t)sl * ld + kmax] = tp[(size_t)sl * ld + j];
            }
            for (int i = tid; i < j; i += BS) {                               // rows of R already final
                const double t = R[(size_t)j * n + i];
                R[(size_t)j * n + i] = R[(size_t)kmax * n + i];
                R[(size_t)kmax * n + i] = t;
            }
        }
        // the pivot column with its pending updates becomes the reflector; column j moves to slot kmax
        for (int i = j + tid; i < m; i += BS) {
            double e = a[(size_t)i * ld + kmax];
#pragma unroll
            for (int q = 0; q < B; ++q)
                if (q < np) e = e - tk[q] * Vg[(size_t)((s0 + q) & (B - 1)) * m + i];
            Vn[i] = e;
            if (kmax != j) a[(size_t)i * ld + kmax] = a[(size_t)i * ld + j];
        }
        __syncthreads();
        QX_TICK(0)
        double ajnorm = norm2_flang_block_wide<QL_RPT>([&](int i) { return Vn[j + i]; }, m - j, tile, QL_RPT * BS, scratch);   // :642
        QX_TICK(1)
        const int ncol = n - j;                                               // columns j+1 .. n (n = residual)
        if (ajnorm != 0.0) {
            if (Vn[j] < 0.0) ajnorm = -ajnorm;
            __syncthreads();
            for (int i = j + tid; i < m; i += BS) {
                double t = Vn[i] / ajnorm;
                if (i == j) t = t + 1.0;
                Vn[i] = t;
            }
            __syncthreads();
            const double ajj = Vn[j];

            // Work split of a tile: a wave owns RPR consecutive rows and all CW column slots, a lane owns the CPT
            // columns lane, lane+64, ... of those rows.  The reflector entries of a row are then read from LDS
            // once per CPT elements (they are wave-uniform), which is what bounds this loop.
            int CW = 64;
            while (CW < ncol) CW <<= 1;                                        // <= BS because n + 1 <= BS
            const int lg = __ffs(CW) - 1;
            const int lane = tid & 63, wv = tid >> 6, NW = BS >> 6;
            const int lcpt = lg - 6;                                           // CPT = CW / 64 in {1, 2, 4, 8}
            const int TR = (NW * QL_RPT) >> lcpt, ltr = __ffs(TR) - 1;         // rows per tile
            const bool flush = (np == B - 1);
            const int nrows = m - j, ntile = (nrows + TR - 1) >> ltr;
            const int vq = tid >> ltr, vr = tid & (TR - 1);
            const bool vact = vq <= np;
            const double *vsrc = Vg + (size_t)((s0 + (vact ? vq : 0)) & (B - 1)) * m + j + vr;
            const size_t tstride = (size_t)TR * ld;                            // elements between consecutive tiles
            const bool summer = tid < ncol;                                    // this thread adds up column j+1+tid
            double sm = 0.0;

            // The tile loop is instantiated for every pending count and column split: straight-line update chains.
            auto run = [&](auto npc, auto lcc, auto ntc) {
                constexpr int NP = decltype(npc)::value;
                constexpr bool NTL = decltype(ntc)::value;                     // stream the working matrix past the caches
                constexpr int LC = decltype(lcc)::value, CPT = 1 << LC, RPR = QL_RPT >> LC;
                constexpr bool FL = (NP == B - 1);
                int kc[CPT];                                                   // my columns (idle slots shadow column j+1)
                double tpk[CPT][NP > 0 ? NP : 1];
#pragma unroll
                for (int c = 0; c < CPT; ++c) {
                    const int kk = lane + 64 * c;
                    kc[c] = j + 1 + (kk < ncol ? kk : 0);
#pragma unroll
                    for (int q = 0; q < NP; ++q) tpk[c][q] = tp[(size_t)((s0 + q) & (B - 1)) * ld + kc[c]];
                }
                double *cbase = a + (size_t)(j + wv * RPR) * ld;               // my first row of tile 0
                double b0[QL_RPT], b1[QL_RPT], b2[QL_RPT], v0 = 0.0, v1 = 0.0, v2 = 0.0;
                const double *fp = cbase;                                      // fetch pointer (runs QL_DEPTH tiles ahead)
                const double *fv = vsrc;
                // Unconditional: the same number of loads on every path keeps the compiler's s_waitcnt counts at
                // the full prefetch depth.  Tiles past the end read (and ignore) whatever follows -- the next
                // problem's rows or the padding the host adds behind the last one.
                auto fetch = [&](double (&buf)[QL_RPT], double &vv) {
#pragma unroll
                    for (int r = 0; r < RPR; ++r)
#pragma unroll
                        for (int c = 0; c < CPT; ++c) {
                            if constexpr (NTL) buf[r * CPT + c] = __builtin_nontemporal_load(fp + (size_t)r * ld + kc[c]);
                            else buf[r * CPT + c] = fp[(size_t)r * ld + kc[c]];
                        }
                    vv = *fv;
                    fp += tstride; fv += TR;
                };
                double *wp = cbase;                                            // write-back pointer of the current tile
                // rows [r, rows) of a finished tile, ascending; reads issued in batches, only the adds are serial
                auto sum_rows = [&](const double *tcol, int r, int rows) {
                    for (; r + 8 <= rows; r += 8) {
                        double x[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) x[u] = tcol[(r + u) << lg];
#pragma unroll
                        for (int u = 0; u < 8; ++u) sm = sm + x[u];
                    }
                    for (; r + 4 <= rows; r += 4) {
                        double x[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) x[u] = tcol[(r + u) << lg];
#pragma unroll
                        for (int u = 0; u < 4; ++u) sm = sm + x[u];
                    }
                    for (; r < rows; ++r) sm = sm + tcol[r << lg];
                };
                // Tile t-1 is summed while tile t is being prepared: its first 8 products are read before the
                // elementwise work of tile t and added after it, so the LDS latency is off the critical path.
                auto process = [&](auto guarded, double (&buf)[QL_RPT], double vnext, int t) {
                    constexpr bool GD = decltype(guarded)::value;              // epilogue: tile may be short or absent
                    if (GD && t >= ntile) return;
                    const int par = t & 1;
                    const double *vtc = vt + (size_t)par * B * TRMAX + wv * RPR;
                    double *tl = tile + (size_t)par * QL_RPT * BS;
                    double *tw = tl + ((wv * RPR) << lg) + lane;
                    const double *tprev = tile + (size_t)(par ^ 1) * QL_RPT * BS + tid;
                    const int rem = nrows - (t << ltr);
                    const bool pipe = summer && t > 0 && TR >= 8;
                    double x[8];
                    if (pipe) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) x[u] = tprev[u << lg];
                    }
                    double e[QL_RPT];
#pragma unroll
                    for (int r = 0; r < RPR; ++r) {
                        double vrow[NP + 1];
#pragma unroll
                        for (int q = 0; q <= NP; ++q) vrow[q] = vtc[q * TRMAX + r];
#pragma unroll
                        for (int c = 0; c < CPT; ++c) {
                            double ev = buf[r * CPT + c];
#pragma unroll
                            for (int q = 0; q < NP; ++q) ev = ev - tpk[c][q] * vrow[q];
                            e[r * CPT + c] = ev;
                        }
                        if (t == 0 && r == 0 && wv == 0) {                     // row j, pending updates applied
#pragma unroll
                            for (int c = 0; c < CPT; ++c) rowj[lane + 64 * c] = e[c];
                        }
#pragma unroll
                        for (int c = 0; c < CPT; ++c) tw[(r << lg) + 64 * c] = vrow[NP] * e[r * CPT + c];   // a(i,j)*a(i,k), :653
                    }
                    if (FL) {                                                  // idle slots rewrite column j+1 with the same values
#pragma unroll
                        for (int r = 0; r < RPR; ++r) {
                            if (!GD || wv * RPR + r < rem) {
#pragma unroll
                                for (int c = 0; c < CPT; ++c) {
                                    if constexpr (NTL) __builtin_nontemporal_store(e[r * CPT + c], wp + (size_t)r * ld + kc[c]);
                                    else wp[(size_t)r * ld + kc[c]] = e[r * CPT + c];
                                }
                            }
                        }
                    }
                    wp += tstride;
                    if (vact) vt[(size_t)(par ^ 1) * B * TRMAX + vq * TRMAX + vr] = vnext;       // reflector rows of tile t+1
                    if (pipe) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) sm = sm + x[u];
                        sum_rows(tprev, 8, TR);
                    } else if (summer && t > 0) {
                        sum_rows(tprev, 0, TR);
                    }
                    QX_TICK2(0)
                    __syncthreads();
                    QX_TICK2(1)
                };
                QX_TICK(2)
                fetch(b0, v0); fetch(b1, v1); fetch(b2, v2);
                if (vact) vt[vq * TRMAX + vr] = v0;
                __syncthreads();
                const int nmain = ((nrows >> ltr) / QL_DEPTH) * QL_DEPTH;      // full tiles handled branch-free
                std::false_type plain_t; std::true_type guard_t;
                int t = 0;
                for (; t < nmain; t += QL_DEPTH) {
                    process(plain_t, b0, v1, t);     fetch(b0, v0);
                    process(plain_t, b1, v2, t + 1); fetch(b1, v1);
                    process(plain_t, b2, v0, t + 2); fetch(b2, v2);
                }
                for (int e3 = 0; e3 < QL_DEPTH; ++e3) {                       // up to three more tiles, the last may be short
                    process(guard_t, b0, v1, t + e3);
#pragma unroll
                    for (int u = 0; u < QL_RPT; ++u) { b0[u] = b1[u]; b1[u] = b2[u]; }
                    v1 = v2; v2 = v0;
                }
                if (summer)                                                    // the last tile (possibly short)
                    sum_rows(tile + (size_t)((ntile - 1) & 1) * QL_RPT * BS + tid, 0, min(TR, nrows - ((ntile - 1) << ltr)));
                __syncthreads();
                QX_TICK(3)
            };
            auto with_np = [&](auto lcc, auto ntc) {
                switch (np) {
                case 0: run(std::integral_constant<int, 0>{}, lcc, ntc); break;
                case 1: run(std::integral_constant<int, 1>{}, lcc, ntc); break;
                case 2: run(std::integral_constant<int, 2>{}, lcc, ntc); break;
                case 3: if constexpr (B > 3) run(std::integral_constant<int, (B > 3 ? 3 : 0)>{}, lcc, ntc); break;
                case 4: if constexpr (B > 4) run(std::integral_constant<int, (B > 4 ? 4 : 0)>{}, lcc, ntc); break;
                case 5: if constexpr (B > 5) run(std::integral_constant<int, (B > 5 ? 5 : 0)>{}, lcc, ntc); break;
                case 6: if constexpr (B > 6) run(std::integral_constant<int, (B > 6 ? 6 : 0)>{}, lcc, ntc); break;
                default: if constexpr (B > 7) run(std::integral_constant<int, (B > 7 ? 7 : 0)>{}, lcc, ntc); break;
                }
            };
            // Working matrices that exceed the Infinity Cache are read and rewritten with non-temporal accesses in the
            // 256-slot steps, which leaves L2 to the reflectors: -6 % with 256 problems of 4096 x 256 in flight, but
            // +5 % with 64 (whose matrices the cache does hold), hence the switch.
            if (stream_nt && lcpt == 2) {
                with_np(std::integral_constant<int, 2>{}, std::true_type{});
            } else {
                switch (lcpt) {
                case 0: with_np(std::integral_constant<int, 0>{}, std::false_type{}); break;
                case 1: with_np(std::integral_constant<int, 1>{}, std::false_type{}); break;
                case 2: with_np(std::integral_constant<int, 2>{}, std::false_type{}); break;
                default: with_np(std::integral_constant<int, 3>{}, std::false_type{}); break;
                }
            }
            if (summer) {
                const int k = j + 1 + tid;
                const double temp = sm / ajj;                                 // :654 (residual: negated, see header)
                tp[(size_t)snew * ld + k] = temp;
                const double rjk = rowj[tid] - temp * ajj;                    // row j is final
                if (k == n) {
                    qtf[j] = rjk;
                } else {
                    R[(size_t)k * n + j] = rjk;
                    double rk = rdiag[k];
                    if (rk != 0.0) {                                          // :656-661
                        const double t2 = rjk / rk;
                        rk = rk * sqrt(fmax(0.0, 1.0 - t2 * t2));
                        const double q = rk / wa[k];
                        if (!(p05 * (q * q) > NLH_EPS)) {
                            const double *cp = a + k;
                            rk = norm2_flang_serial([&](int i2) {
                                const int i = j + 1 + i2;
                                double e = cp[(size_t)i * ld];
                                if (!flush) {
                                    for (int q2 = 0; q2 < np; ++q2) {
                                        const int sl = (s0 + q2) & (B - 1);
                                        e = e - tp[(size_t)sl * ld + k] * Vg[(size_t)sl * m + i];
                                    }
                                }
                                return e - temp * Vn[i];
                            }, m - j - 1);
                            wa[k] = rk;
                        }
                        rdiag[k] = rk;
                    }
                }
            }
            if (flush) { s0 = snew; np = 1; } else { np += 1; }
            QX_TICK(4)
        } else {
            // no reflector: row j of the trailing columns is final as it stands
            for (int kk = tid; kk < ncol; kk += BS) {
                const int k = j + 1 + kk;
                double e = a[(size_t)j * ld + k];
#pragma unroll
                for (int q = 0; q < B; ++q)
                    if (q < np) e = e - tp[(size_t)((s0 + q) & (B - 1)) * ld + k] * Vg[(size_t)((s0 + q) & (B - 1)) * m + j];
                if (k == n) qtf[j] = e; else R[(size_t)k * n + j] = e;
            }
        }
        __syncthreads();
        if (tid == 0) rdiag[j] = -ajnorm;
        __syncthreads();
    }

#ifdef QX_PROFILE
    if (p == 0 && tid == 0)
        printf("qx total in loop %llu; ", wall_clock64() - tstart);
        printf("qx cycles(100MHz): pivot+form %llu norm %llu scale %llu pass %llu post %llu misc %llu | elementwise+sum %llu barrier %llu - %llu\n", tc[0], tc[1], tc[2],
               tc[3], tc[4], tc[5], tc2[0], tc2[1], tc2[2]);
#endif
    // Q^T f (:241-253): rows below n carry the pending updates, rows above are the qtf entries
    for (int i = tid; i < m; i += BS) {
        double e;
        if (i < n) {
            e = qtf[i];
        } else {
            e = a[(size_t)i * ld + n];
#pragma unroll
            for (int q = 0; q < B; ++q)
                if (q < np) e = e - tp[(size_t)((s0 + q) & (B - 1)) * ld + n] * Vg[(size_t)((s0 + q) & (B - 1)) * m + i];
        }
        w4[i] = e;
    }
    for (int k = tid; k < n; k += BS) { R[(size_t)k * n + k] = rdiag[k]; v.rdiag[(size_t)p * n + k] = rdiag[k]; }
    __syncthreads();
    if (!s) return;
    if (tid == 0) { s->factor_kind = 1; s->qr_count += 1; }
    if (first) {
        lm_head<true>(n, R, n, ipvt, acnorm, qtf, xall + (size_t)p * n, v.diag + (size_t)p * n,
                      v.diag_prev + (size_t)p * n, s, factor, gtol, ST_QR_READY, red, scratch);
    } else {
        if (tid == 0) s->stage = ST_QR_READY;
    }
}
