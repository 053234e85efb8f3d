// nlh_kernels_broyden.h -- dense kernels of quasi_newton_solver (qns_solve,
// src/nonlin_solve.f90:156-427): Householder QR with explicit Q (qr_factor, :289), the rank-one
// Broyden update of B (:301-306) and of its QR factors (qr_rank1_update, :307), B^T f and
// Q^T f (:313, :322) and the triangular solve (:327).
//
// The reference takes all of these from the third-party linalg library (LAPACK / qrupdate,
// unpinned); the CPU restatement defines them as the published unblocked algorithms with every
// sum in ascending index order, and the kernels below perform exactly those operations on every
// matrix element (a sequential sum is owned by one thread; rotations are elementwise), so Q, R, B
// and every iterate are bit-identical to the CPU path.
//
// Layout per problem: B and Q column-major n x n; R ROW-major (Rt[i*n + c]) so that a thread
// per column reads consecutive addresses across a wave.  The QR work array [A | E] is stored
// row-major too: A^T-of-B becomes R in place and the transformed identity E = Q^T, read
// row-major, IS Q column-major -- the factorisation writes both results where they are used.
#pragma once
#include "nlh_common.h"

// DLARTG (LAPACK 3.10): c >= 0, r carries the sign of f.
__device__ __forceinline__ void givens_dev(double f, double g, double &c, double &s, double &r)
{
    if (g == 0.0) { c = 1.0; s = 0.0; r = f; return; }
    if (f == 0.0) { c = 0.0; s = 1.0; r = g; return; }
    const double d = sqrt(f * f + g * g);
    c = fabs(f) / d;
    r = copysign(d, f);
    s = g / r;
}

// Q <- I, and the first reflector's column: vbuf[i] = A(i,0).
static __global__ void __launch_bounds__(256)
k_qn_qr_init(int n, const double *__restrict__ Rt, double *__restrict__ Q, double *__restrict__ vbuf,
        const LmState *__restrict__ gst, int gwant)
{
    const int p = blockIdx.y;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    const size_t nn = (size_t)n * n;
    double *Qp = Q + p * nn;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nn; e += (size_t)gridDim.x * 256)
        Qp[e] = (e % n == e / n) ? 1.0 : 0.0;
    if (blockIdx.x == 0) {
        const double *A = Rt + p * nn;
        double *v = vbuf + (size_t)p * 2 * n;
        for (int i = 1 + threadIdx.x; i < n; i += 256) v[i] = A[(size_t)i * n];
    }
}

// Householder step j, first half, on the row-major work array [A | E] (thread k owns column k; k < n: A,
// k >= n: E): the reflector of column j from the copy the previous step left in vbuf (slot j & 1), then
// w_k = tau * (T(j,k) + sum_{i>j} v_i T(i,k)) for every column still in play, summed by one thread in
// ascending i with QN_U loads in flight.  st[p] = {tau, scal, beta} for the second half (tau = 0: H = I).
#define QN_DOT_BS 128
#define QN_U 32
// General shape: `rows` rows, A has ncA columns (row-major, ld ncA), E has ncE columns (row-major, ld ncE).
// Broyden: rows = ncA = ncE = n (E = Q^T accumulator).  Constrained least squares: rows = m, ncA = n,
// ncE = 1 (E = the right-hand side f, which becomes Q^T f).
// GV = true: more rows than LDS holds (rows > 18000: the reflector would not fit) -- the reflector stays in global memory
// and is scaled where it is used: (v_i * scal) * T(i,k) is the stored-then-multiplied value's two roundings exactly, so
// the same bits as the LDS form at any size.  What the reference has no limit for (a 65536-row constrained least-squares
// problem, a polynomial fit through 10^5 points) then works, at the price of L2 instead of LDS reads in the chains.
template <bool GV>
__global__ void __launch_bounds__(QN_DOT_BS)
k_qn_house_dot(int rows, int ncA, int ncE, int j, const double *__restrict__ Aall, const double *__restrict__ Eall,
               const double *__restrict__ vbuf, double *__restrict__ wbuf, double *__restrict__ st,
        const LmState *__restrict__ gst, int gwant)
{
    extern __shared__ double vs_lds[];             // rows: reflector, rows j+1 .. rows-1 (GV: unused)
    __shared__ double sq_sh;
    const int p = blockIdx.y, tid = threadIdx.x;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    const double *A = Aall + (size_t)p * rows * ncA, *E = Eall + (size_t)p * rows * ncE;
    const double *vcur = vbuf + ((size_t)p * 2 + (j & 1)) * rows;
    double *vs = vs_lds;
    if (!GV) {
        for (int i0 = j + 1 + tid; i0 < rows; i0 += 8 * QN_DOT_BS) {      // 8 loads in flight per thread
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * QN_DOT_BS; t[u] = (i < rows) ? vcur[i] : 0.0; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * QN_DOT_BS; if (i < rows) vs[i] = t[u]; }
        }
        __syncthreads();
    }
    const double *vr = GV ? vcur : vs;             // the unscaled reflector, wherever it is
    if (tid == 0) {                                // one ordered sum; reads batched 16 at a time
        double s = 0.0;
        int i = j + 1;
        for (; i + 16 <= rows; i += 16) {
            double t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = vr[i + u];
#pragma unroll
            for (int u = 0; u < 16; ++u) s = s + t[u] * t[u];
        }
        for (; i < rows; ++i) s = s + vr[i] * vr[i];
        sq_sh = s;
    }
    __syncthreads();
    const double sq = sq_sh;
    if (sq == 0.0) {                               // H = I (uniform)
        if (blockIdx.x == 0 && tid == 0) { st[(size_t)p * 4] = 0.0; st[(size_t)p * 4 + 1] = 0.0; st[(size_t)p * 4 + 2] = 0.0; }
        return;
    }
    const double alpha = A[(size_t)j * ncA + j];
    const double beta = -copysign(sqrt(alpha * alpha + sq), alpha);
    const double tau = (beta - alpha) / beta;
    const double scal = 1.0 / (alpha - beta);
    if (blockIdx.x == 0 && tid == 0) { st[(size_t)p * 4] = tau; st[(size_t)p * 4 + 1] = scal; st[(size_t)p * 4 + 2] = beta; }
    if (!GV) {
        for (int i = j + 1 + tid; i < rows; i += QN_DOT_BS) vs[i] = vs[i] * scal;
        __syncthreads();
    }
    const int k = blockIdx.x * QN_DOT_BS + tid;
    if (k >= ncA + ncE || (k < ncA && k <= j)) return;
    const double *T = (k < ncA) ? A + k : E + (k - ncA);
    const size_t ld = (k < ncA) ? ncA : ncE;
    double w = T[(size_t)j * ld];
    for (int i = j + 1; i < rows; i += QN_U) {     // the chain is serial in i; the loads are not
        double t[QN_U], vv[QN_U];
#pragma unroll
        for (int u = 0; u < QN_U; ++u) {
            t[u] = (i + u < rows) ? T[(size_t)(i + u) * ld] : 0.0;
            vv[u] = (i + u < rows) ? (GV ? vcur[i + u] * scal : vs[i + u]) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < QN_U; ++u)
            if (i + u < rows) w = w + vv[u] * t[u];
    }
    wbuf[(size_t)p * (ncA + ncE) + k] = tau * w;
}

// The same first half with the loads spread over a whole workgroup: 16 columns per workgroup, 256 threads =
// 16 columns x 16 row lanes.  A tile of 256 rows x 16 columns is fetched by all threads at once (each thread 16
// independent loads of 128-byte row segments), the products v_i T(i,k) go to LDS, and one thread per column adds
// them in row order -- the sum is still a single ordered chain per column (bit-identical to the kernel above), but
// the memory latency is paid once per 256 rows instead of once per 32.  The loads of tile t+1 are in flight while
// tile t is summed.  Reflector staged in LDS: rows <= QN_DOT2_MAXROWS.
#define QN_DOT2_MAXROWS 8192
#define QN_DOT2_CG 16
#define QN_DOT2_TR 256
static __global__ void __launch_bounds__(256)
k_qn_house_dot2(int rows, int ncA, int ncE, int j, const double *__restrict__ Aall, const double *__restrict__ Eall,
                const double *__restrict__ vbuf, double *__restrict__ wbuf, double *__restrict__ st,
        const LmState *__restrict__ gst, int gwant)
{
    extern __shared__ double sm2[];
    double *vs = sm2;                                   // rows (index = row)
    double *prod = sm2 + rows;                          // [2][QN_DOT2_TR][QN_DOT2_CG]
    __shared__ double sq_sh;
    const int p = blockIdx.y, tid = threadIdx.x;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    const double *A = Aall + (size_t)p * rows * ncA, *E = Eall + (size_t)p * rows * ncE;
    const double *vcur = vbuf + ((size_t)p * 2 + (j & 1)) * rows;
    for (int i0 = j + 1 + tid; i0 < rows; i0 += 8 * 256) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * 256; t[u] = (i < rows) ? vcur[i] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * 256; if (i < rows) vs[i] = t[u]; }
    }
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        int i = j + 1;
        for (; i + 16 <= rows; i += 16) {
            double t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = vs[i + u];
#pragma unroll
            for (int u = 0; u < 16; ++u) s = s + t[u] * t[u];
        }
        for (; i < rows; ++i) s = s + vs[i] * vs[i];
        sq_sh = s;
    }
    __syncthreads();
    const double sq = sq_sh;
    if (sq == 0.0) {                                    // H = I (uniform)
        if (blockIdx.x == 0 && tid == 0) { st[(size_t)p * 4] = 0.0; st[(size_t)p * 4 + 1] = 0.0; st[(size_t)p * 4 + 2] = 0.0; }
        return;
    }
    const double alpha = A[(size_t)j * ncA + j];
    const double beta = -copysign(sqrt(alpha * alpha + sq), alpha);
    const double tau = (beta - alpha) / beta;
    const double scal = 1.0 / (alpha - beta);
    if (blockIdx.x == 0 && tid == 0) { st[(size_t)p * 4] = tau; st[(size_t)p * 4 + 1] = scal; st[(size_t)p * 4 + 2] = beta; }
    for (int i = j + 1 + tid; i < rows; i += 256) vs[i] = vs[i] * scal;
    __syncthreads();

    const int c = tid & (QN_DOT2_CG - 1), r = tid >> 4;            // column within the group, row lane
    const int k = blockIdx.x * QN_DOT2_CG + c;
    const bool live = (k < ncA + ncE) && !(k < ncA && k <= j);
    const double *T = (k < ncA) ? A + k : E + (k - ncA);
    const size_t ld = (k < ncA) ? ncA : ncE;
    double w = (live && r == 0) ? T[(size_t)j * ld] : 0.0;          // the summing thread of column c is (c, r = 0)
    double tl[16];
    const int ibeg = j + 1;
#define QN_DOT2_LOAD(i0)                                                                   \
    _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                       \
        const int i = (i0) + r + 16 * u;                                                   \
        tl[u] = (live && i < rows) ? T[(size_t)i * ld] : 0.0;                              \
    }
    QN_DOT2_LOAD(ibeg)
    int buf = 0;
    for (int i0 = ibeg; i0 < rows; i0 += QN_DOT2_TR) {
        double *pb = prod + (size_t)buf * QN_DOT2_TR * QN_DOT2_CG;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = i0 + r + 16 * u;
            if (i < rows) pb[(r + 16 * u) * QN_DOT2_CG + c] = vs[i] * tl[u];
        }
        __syncthreads();
        if (i0 + QN_DOT2_TR < rows) { QN_DOT2_LOAD(i0 + QN_DOT2_TR) }
        if (r == 0 && live) {
            const int lim = min(QN_DOT2_TR, rows - i0);
            int ii = 0;
            for (; ii + 16 <= lim; ii += 16) {
                double q[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) q[u] = pb[(ii + u) * QN_DOT2_CG + c];
#pragma unroll
                for (int u = 0; u < 16; ++u) w = w + q[u];
            }
            for (; ii < lim; ++ii) w = w + pb[ii * QN_DOT2_CG + c];
        }
        buf ^= 1;
    }
#undef QN_DOT2_LOAD
    if (r == 0 && live) wbuf[(size_t)p * (ncA + ncE) + k] = tau * w;
}

// Step j with the second half of step j-1 folded in: every element of the trailing matrix is read once, receives
// the pending update T(i,k) -= v'_i w'_k of step j-1 (v', w', tau' from the slots of step j-1), is written back, and
// its updated value goes straight into the ordered sums of step j.  The reflector of step j is rebuilt by every
// workgroup from column j (the pending update applied on the fly); workgroup 0 publishes it for step j+1.
// Same operations per element as k_qn_house_dot2 + k_qn_house_apply (bit-identical), one pass and one launch less
// per step.  The last step's update is applied by k_qn_house_apply.  w and st are double-buffered by step parity
// ([problem][slot]); LDS: two reflectors + two product tiles, rows <= QN_FUSED_MAXROWS.
#define QN_FUSED_MAXROWS 4096
// dynamic LDS of k_qn_house_fused: two reflectors, two padded product tiles of 4096 entries (also the padded squares)
static inline size_t qn_fused_lds(int rows, int dbuf)
{
    const size_t tile = 4096 + 4096 / 16, sq = (size_t)rows + rows / 16 + 16;
    return sizeof(double) * (2 * (size_t)rows + std::max((dbuf ? 2 : 1) * tile, sq));
}
// CG = columns per workgroup (16: wide matrices; 4: tall-skinny ones, so that enough workgroups exist); a tile is
// (4096 / CG) rows x CG columns.
template <int CG>
__global__ void __launch_bounds__(CG == 4 ? 512 : 256)
k_qn_house_fused(int rows, int ncA, int ncE, int j, double *__restrict__ Aall, double *__restrict__ Eall,
                 double *__restrict__ vbuf, double *__restrict__ wbuf, double *__restrict__ st,
        const LmState *__restrict__ gst, int gwant, int dbuf /* 1: two product tiles (a lone problem: the next tile's
        products are formed while this one is summed); 0: one (batches: 64 KB of LDS instead of 100, two workgroups per CU) */)
{
    extern __shared__ double sm3[];
    double *vsp = sm3;                                  // previous reflector, scaled (rows >= j)
    double *vs = sm3 + rows;                            // this step's reflector (rows >= j+1)
    constexpr int RL = 256 / CG, TR = RL * 16;          // row lanes, rows per tile
    constexpr int TRP = TR + TR / 16;                   // (tile rows + padding: QN_FUSED_LDS)
    double *prod = sm3 + 2 * (size_t)rows;              // [2][TR][CG]
    __shared__ double sq_sh, alpha_sh;
    const int p = blockIdx.y, tid = threadIdx.x, nc = ncA + ncE, jp = j - 1;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    double *A = Aall + (size_t)p * rows * ncA, *E = Eall + (size_t)p * rows * ncE;
    const double *vprev_g = vbuf + ((size_t)p * 2 + (jp & 1)) * rows;
    double *vcur_g = vbuf + ((size_t)p * 2 + (j & 1)) * rows;
    const double *stp = st + ((size_t)p * 2 + (jp & 1)) * 4;
    double *stc = st + ((size_t)p * 2 + (j & 1)) * 4;
    const double *wprev = wbuf + ((size_t)p * 2 + (jp & 1)) * nc;
    double *wcur = wbuf + ((size_t)p * 2 + (j & 1)) * nc;
    const double tau_p = (j > 0) ? stp[0] : 0.0, scal_p = (j > 0) ? stp[1] : 0.0;
    const bool pend = tau_p != 0.0;                     // step j-1 was a real reflector whose update is outstanding
    // Column j-1 is finished here: (diagonal, zeros below).  Its rows >= j-1 were deliberately NOT written during
    // step j-1 (every workgroup of that launch was still reading them to rebuild the reflector), so this is where
    // they get their final values: beta of step j-1, or the plain diagonal value if that step was H = I.
    const double dval_p = (j > 0) ? (pend ? stp[2] : stp[3]) : 0.0;

    // (Round 4: every load of this prologue is issued before any is used, with clamped indices -- a load under a
    // condition, or in a loop that consumes it at once, is waited for before the next one is issued: sixteen trips to
    // L2 per loop instead of one.)
    constexpr int PR = QN_FUSED_MAXROWS / 256;          // rows per thread
#ifdef QN_DBG_CLK
    long long ck[8]; ck[0] = wall_clock64();
#endif
    // CG == 4 runs 512 threads: waves 0-3 PRODUCE (everything below but the column sums), waves 4-7 only SUM -- wave 4 + c
    // adds column c's products of tile t down its lanes while the producers form tile t + 1 (two product tiles, one LDS-only
    // barrier per tile); with four waves doing both in turn a tile cost 6 us, of which 2.7 are the sum.  CG == 16: 256 threads.
    const bool prodw = (CG != 4) || tid < 256;
    const int c = tid % CG, r = (tid & 255) / CG;
    const int k = blockIdx.x * CG + c;
    double *T = (k < ncA) ? A + k : E + (k - ncA);
    const size_t ld = (k < ncA) ? ncA : ncE;
    // (tile loads: unconditional, clamped row -- Tl points at a valid column -- and pinned where the values are needed: the
    // compiler would put the loads back under their condition, where each is waited for before the next is issued)
    const double *Tl = (k < nc) ? T : A;
    const size_t ldl = (k < nc) ? ld : (size_t)ncA;
    double tl[16] = {0.0}, ejp0 = 0.0, tj0 = 0.0;
    const int ibeg = j + 1;
#define QN_F_LOAD(i0)                                                                      \
    _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                       \
        const int i = (i0) + r + RL * u;                                                   \
        tl[u] = Tl[(size_t)(i < rows ? i : rows - 1) * ldl];                               \
    }
#define QN_F_PIN _Pragma("unroll") for (int u = 0; u < 16; ++u) asm volatile("" : "+v"(tl[u]));
    if (prodw) {
        double vp[PR], cj[PR];
        const double wpj = pend ? wprev[j] : 0.0;
#pragma unroll
        for (int u = 0; u < PR; ++u) {
            const int i = j + tid + 256 * u, ic = i < rows ? i : rows - 1;
            vp[u] = vprev_g[ic];
            cj[u] = A[(size_t)ic * ncA + j];
        }
        // ... and this thread's entries of the first tile, of row j and of row j - 1: they do not depend on the reflector,
        // so they travel while it is being rebuilt and its squares are summed
        QN_F_LOAD(ibeg)
        ejp0 = Tl[(size_t)(j > 0 ? jp : 0) * ldl];
        tj0 = Tl[(size_t)j * ldl];
#pragma unroll
        for (int u = 0; u < PR; ++u) asm volatile("" : "+v"(vp[u]), "+v"(cj[u]));
        // column j with the pending update applied: alpha (row j) and the unscaled reflector (rows > j)
#pragma unroll
        for (int u = 0; u < PR; ++u) {
            const int i = j + tid + 256 * u;
            if (i < rows) {
                double t = cj[u];
                if (pend) {
                    const double vsi = vp[u] * scal_p;
                    vsp[i] = vsi;
                    t = t - vsi * wpj;
                }
                if (i == j) alpha_sh = t; else vs[i] = t;
            }
        }
    }
    __syncthreads();
#ifdef QN_DBG_CLK
    ck[1] = wall_clock64();
#endif
    if (prodw && blockIdx.x == 0)
        for (int i = j + 1 + tid; i < rows; i += 256) vcur_g[i] = vs[i];
    // sum of squares in row order: the squares are formed by everybody (the same product the one-thread loop forms),
    // one wave adds them down its lanes (round 4; the one-thread loop with the multiply inside ran the LDS reads, the
    // multiplies and the adds of a group one after the other: 43 us at 4096 rows)
    double *sqb = prod;                                 // (the product tiles are not in use yet)
    // (term q at sqb[q + q/16]: a lane's run of sixteen starts 17 doubles after its neighbour's -- no bank conflicts)
    if (prodw)
        for (int i = j + 1 + tid; i < rows; i += 256) { const double t = vs[i]; const int q = i - j - 1; sqb[q + (q >> 4)] = t * t; }
    __syncthreads();
    if (tid < 64) {                                     // one wave, down the lanes (nlh_common.h)
        const double s = ordered_sum_wave<64>([&](int q) { return sqb[q + (q >> 4)]; }, rows - j - 1, 0.0);
        if (tid == 0) sq_sh = s;
    }
    __syncthreads();
#ifdef QN_DBG_CLK
    ck[2] = wall_clock64();
#endif
    const double sq = sq_sh, alpha = alpha_sh;
    const bool refl = sq != 0.0;                        // H_j != I
    double tau = 0.0, scal = 0.0, beta = 0.0;
    if (refl) {
        beta = -copysign(sqrt(alpha * alpha + sq), alpha);
        tau = (beta - alpha) / beta;
        scal = 1.0 / (alpha - beta);
    }
    if (blockIdx.x == 0 && tid == 0) { stc[0] = tau; stc[1] = scal; stc[2] = beta; stc[3] = alpha; }
    const bool hasjp = (j > 0) && (jp / CG == (int)blockIdx.x);     // this workgroup finishes column j-1
    if (!refl && !pend && !hasjp) return;               // nothing to sum, nothing outstanding (uniform)
    if (refl && prodw)
        for (int i = j + 1 + tid; i < rows; i += 256) vs[i] = vs[i] * scal;
    __syncthreads();

    const bool inr = k < nc;
    const bool isjp = (j > 0) && (k < ncA) && (k == jp);
    const bool isj = (k < ncA) && (k == j);             // never stored in this launch (other workgroups read it)
    const bool upd = inr && (isjp || (pend && !(k < ncA && k < jp)));  // columns written back: the update, or the finish of j-1
    const bool live = refl && inr && !(k < ncA && k <= j);           // columns that take part in step j's sums
    const double wp = (upd && !isjp && pend) ? wprev[k] : 0.0;
    double w = 0.0;
    asm volatile("" : "+v"(ejp0), "+v"(tj0));
    if (prodw && r == 0 && inr) {
        if (upd) {                                      // row j-1: the pending update, or the diagonal of column j-1
            double *e = T + (size_t)jp * ld;
            *e = isjp ? dval_p : ejp0 - wp;
        }
        double t = tj0;                                 // row j
        if (upd) { t = isjp ? 0.0 : t - vsp[j] * wp; if (!isj) T[(size_t)j * ld] = t; }
        w = t;
    }
    // CG == 4: wave wv sums column wv of the workgroup (its start value -- row j -- comes from the thread that formed it)
    __shared__ double w0_sh[4];
    const int wv = (tid >> 6) & 3, kw = blockIdx.x * CG + wv;
    const bool wlive = (CG == 4) && refl && kw < nc && !(kw < ncA && kw <= j);
    double ww = 0.0;
    if constexpr (CG == 4) {
        if (prodw && r == 0) w0_sh[c] = w;
        __syncthreads();
        ww = w0_sh[wv];
    }
#ifdef QN_DBG_CLK
    ck[3] = wall_clock64();
#endif
#ifdef QN_DBG_CLK
    ck[4] = ck[3];
#endif
    if constexpr (CG == 4) {
        if (prodw) {
            QN_F_PIN                                    // (the first tile was requested in the prologue)
            int buf = 0;
            for (int i0 = ibeg; i0 < rows; i0 += TR, buf ^= 1) {
                // tile buffer buf was last read by the sums of two tiles ago, which ended before their waves joined the
                // barrier of the previous tile
                double *pb = prod + (size_t)buf * TRP * CG;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int i = i0 + r + RL * u;
                    if (i < rows) {
                        double t = tl[u];
                        if (upd) { t = isjp ? 0.0 : t - vsp[i] * wp; if (!isj) T[(size_t)i * ld] = t; }
                        if (refl) pb[c * TRP + (r + RL * u) + ((r + RL * u) >> 4)] = vs[i] * t;      // column-major, padded
                    }
                }
                QN_F_LOAD(i0 + TR)                      // (clamped: the tile after the last reads row rows-1 again)
                nlh_lds_barrier();                      // publishes the tile; orders LDS traffic only, the loads stay in flight
                QN_F_PIN
            }
        } else {
            int buf = 0;
            for (int i0 = ibeg; i0 < rows; i0 += TR, buf ^= 1) {
                nlh_lds_barrier();
                if (wlive) {
                    const double *pc = prod + (size_t)buf * TRP * CG + wv * TRP;
                    ww = ordered_sum_wave<32>([&](int i) { return pc[i + (i >> 4)]; }, min(TR, rows - i0), ww);
                }
            }
        }
    } else {
    QN_F_PIN                                            // (the first tile was requested in the prologue)
#ifdef QN_DBG_CLK
    ck[4] = wall_clock64();
#endif
    int buf = 0;
    for (int i0 = ibeg; i0 < rows; i0 += TR) {
        double *pb = prod + (size_t)(dbuf ? buf : 0) * TRP * CG;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = i0 + r + RL * u;
            if (i < rows) {
                double t = tl[u];
                if (upd) { t = isjp ? 0.0 : t - vsp[i] * wp; if (!isj) T[(size_t)i * ld] = t; }
                if (refl) pb[(r + RL * u) * CG + c] = vs[i] * t;
            }
        }
        __syncthreads();
        QN_F_LOAD(i0 + TR)                              // (clamped: the tile after the last reads row rows-1 again)
        if (r == 0 && live) {
            const int lim = min(TR, rows - i0);
            int ii = 0;
            for (; ii + 16 <= lim; ii += 16) {
                double q[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) q[u] = pb[(ii + u) * CG + c];
#pragma unroll
                for (int u = 0; u < 16; ++u) w = w + q[u];
            }
            for (; ii < lim; ++ii) w = w + pb[ii * CG + c];
        }
        QN_F_PIN
        buf ^= 1;
        if (!dbuf) __syncthreads();                     // (uniform) the sums have read the tile before it is written again
    }
    }
#undef QN_F_LOAD
#undef QN_F_PIN
    if constexpr (CG == 4) {
        if (!prodw && wlive && (tid & 63) == 0) wcur[blockIdx.x * CG + wv] = tau * ww;
    } else if (r == 0 && live) {
        wcur[k] = tau * w;
    }
#ifdef QN_DBG_CLK
    ck[5] = wall_clock64();
    if (tid == 0 && p == 0 && blockIdx.x == gridDim.x - 2 && j == 10)
        printf("k_qn_house_fused<%d> rows %d nc %d: prologue %lld, sumsq %lld, scale %lld, first tile load %lld, tiles %lld (x10 ns)\n", CG, rows, nc,
               ck[1] - ck[0], ck[2] - ck[1], ck[3] - ck[2], ck[4] - ck[3], ck[5] - ck[4]);
#endif
}

// Second half: T(j,k) -= w_k, T(i,k) -= v_i w_k (elementwise, one thread per column and QN_RC rows),
// column j becomes (beta, 0, ..., 0), and the updated column j+1 is copied to the other vbuf slot.
#define QN_RC 16
static __global__ void __launch_bounds__(256)
k_qn_house_apply(int rows, int ncA, int ncE, int j, double *__restrict__ Aall, double *__restrict__ Eall,
                 double *__restrict__ vbuf, const double *__restrict__ wbuf, const double *__restrict__ st,
                 int wps /* w doubles per problem */, int sps /* st doubles per problem */,
                 int fixcol /* 1 after k_qn_house_fused: column j was never stored during step j, finish it here */,
        const LmState *__restrict__ gst, int gwant)
{
    const int p = blockIdx.z;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    double *A = Aall + (size_t)p * rows * ncA, *E = Eall + (size_t)p * rows * ncE;
    const double *vcur = vbuf + ((size_t)p * 2 + (j & 1)) * rows;
    double *vnext = vbuf + ((size_t)p * 2 + ((j + 1) & 1)) * rows;
    const double tau = st[(size_t)p * sps], scal = st[(size_t)p * sps + 1], beta = st[(size_t)p * sps + 2];
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int i0 = j + blockIdx.y * QN_RC, i1 = min(rows, i0 + QN_RC);
    if (k >= ncA + ncE || (k < ncA && k < j)) return;
    const bool next_owner = (k == j + 1) && (k < ncA);
    double *T = (k < ncA) ? A + k : E + (k - ncA);
    const size_t ld = (k < ncA) ? ncA : ncE;
    if (tau == 0.0) {
        if (next_owner)
            for (int i = max(i0, j + 2); i < i1; ++i) vnext[i] = T[(size_t)i * ld];
        if (fixcol && k == j && k < ncA) {                  // H = I: (diagonal value, zeros below)
            const double alpha = st[(size_t)p * sps + 3];
            for (int i = i0; i < i1; ++i) T[(size_t)i * ld] = (i == j) ? alpha : 0.0;
        }
        return;
    }
    if (k == j) {
        for (int i = i0; i < i1; ++i) T[(size_t)i * ld] = (i == j) ? beta : 0.0;
        return;
    }
    const double w = wbuf[(size_t)p * wps + k];
    double t[QN_RC];
#pragma unroll
    for (int u = 0; u < QN_RC; ++u) t[u] = (i0 + u < i1) ? T[(size_t)(i0 + u) * ld] : 0.0;
#pragma unroll
    for (int u = 0; u < QN_RC; ++u) {
        const int i = i0 + u;
        if (i < i1) {
            const double a = (i == j) ? t[u] - w : t[u] - (vcur[i] * scal) * w;
            T[(size_t)i * ld] = a;
            if (next_owner && i >= j + 2) vnext[i] = a;
        }
    }
}

// vbuf slot 0 <- column 0 of the row-major A (the first reflector's column).
static __global__ void __launch_bounds__(256)
k_qn_col0(int rows, int ncA, const double *__restrict__ Aall, double *__restrict__ vbuf)
{
    const int p = blockIdx.y;
    const double *A = Aall + (size_t)p * rows * ncA;
    double *v = vbuf + (size_t)p * 2 * rows;
    const int i = 1 + blockIdx.x * 256 + threadIdx.x;
    if (i < rows) v[i] = A[(size_t)i * ncA];
}

// out_i = sum_j v_j J(i,j), J column-major m x n, j ascending from an accumulator of zero (DGEMV 'N').
static __global__ void __launch_bounds__(256)
k_matvec_cm(int m, int n, const double *__restrict__ J, const double *__restrict__ v, double *__restrict__ out,
            const LmState *__restrict__ gst, int gwant)
{
    extern __shared__ double xs[];
    const int p = blockIdx.y;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    for (int k = threadIdx.x; k < n; k += 256) xs[k] = v[(size_t)p * n + k];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const double *a = J + (size_t)p * m * n + i;
    double t = 0.0;
    int j = 0;
    for (; j + 8 <= n; j += 8) {
        double c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = a[(size_t)(j + u) * m];
#pragma unroll
        for (int u = 0; u < 8; ++u) t = t + xs[j + u] * c[u];
    }
    for (; j < n; ++j) t = t + xs[j] * a[(size_t)j * m];
    out[(size_t)p * m + i] = t;
}

// s = (df - B dx) / x2   (:301-302): thread per row, sum over columns ascending.
static __global__ void __launch_bounds__(256)
k_qn_resid(int n, const double *__restrict__ B, const double *__restrict__ dx, const double *__restrict__ df,
           double x2, const double *__restrict__ x2all /* per problem, or NULL: x2 */, double *__restrict__ s,
        const LmState *__restrict__ gst, int gwant)
{
    extern __shared__ double xs[];
    const int p = blockIdx.y;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    for (int k = threadIdx.x; k < n; k += 256) xs[k] = dx[(size_t)p * n + k];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double *b = B + (size_t)p * n * n + i;
    double t = 0.0;
    int j = 0;
    for (; j + 8 <= n; j += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = b[(size_t)(j + u) * n];
#pragma unroll
        for (int u = 0; u < 8; ++u) t = t + v[u] * xs[j + u];
    }
    for (; j < n; ++j) t = t + b[(size_t)j * n] * xs[j];
    s[(size_t)p * n + i] = (df[(size_t)p * n + i] - t) / (x2all ? x2all[p] : x2);
}

// B += s dx^T  (rank1_update, :306)
static __global__ void __launch_bounds__(256)
k_qn_rank1(int n, double *__restrict__ B, const double *__restrict__ s, const double *__restrict__ dx,
        const LmState *__restrict__ gst, int gwant)
{
    const int p = blockIdx.z;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i >= n) return;
    double *b = B + (size_t)p * n * n + (size_t)j * n + i;
    *b = *b + s[(size_t)p * n + i] * dx[(size_t)p * n + j];
}

// out_k = sign * sum_i M(i,k) f_i, M column-major m x n, sum over i ascending (grad = B^T f, -Q^T f, Q^T u, J^T f).
// 16 columns per workgroup; a tile of 256 rows x 16 columns is fetched by all 256 threads (coalesced along the
// rows), the products go to LDS and one thread per column adds them in row order; the loads of the next tile are
// in flight during the sums.
static __global__ void __launch_bounds__(256)
k_qn_colsdot(int m, int n, const double *__restrict__ M, const double *__restrict__ f, double *__restrict__ out,
             double sign,
        const LmState *__restrict__ gst, int gwant)
{
    __shared__ double prod[2][16 * 257];
    const int p = blockIdx.y, t = threadIdx.x;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    const int k0 = blockIdx.x * 16;
    const double *Mp = M + (size_t)p * m * n;
    const double *fp = f + (size_t)p * m;
    double acc = 0.0;
    double tl[16], fi = 0.0;
#define CD_LOAD(i0)                                                                        \
    {                                                                                      \
        const int i = (i0) + t;                                                            \
        fi = (i < m) ? fp[i] : 0.0;                                                        \
        _Pragma("unroll") for (int u = 0; u < 16; ++u)                                     \
            tl[u] = (i < m && k0 + u < n) ? Mp[(size_t)(k0 + u) * m + i] : 0.0;            \
    }
    CD_LOAD(0)
    int buf = 0;
    for (int i0 = 0; i0 < m; i0 += 256) {
#pragma unroll
        for (int u = 0; u < 16; ++u) prod[buf][u * 257 + t] = tl[u] * fi;
        __syncthreads();
        if (i0 + 256 < m) CD_LOAD(i0 + 256)
        if (t < 16) {
            const int lim = min(256, m - i0);
            const double *pc = &prod[buf][t * 257];
            int ii = 0;
            for (; ii + 16 <= lim; ii += 16) {
                double q[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) q[u] = pc[ii + u];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc = acc + q[u];
            }
            for (; ii < lim; ++ii) acc = acc + pc[ii];
        }
        buf ^= 1;
    }
#undef CD_LOAD
    if (t < 16 && k0 + t < n) out[(size_t)p * n + k0 + t] = sign * acc;
}

// DQRTV1: rotations folding w into w(0), generated from the bottom (one thread per problem).
static __global__ void k_qn_fold(int n, double *__restrict__ w, double *__restrict__ c, double *__restrict__ s,
        const LmState *__restrict__ gst, int gwant)
{
    const int p = blockIdx.x;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    if (threadIdx.x != 0) return;
    double *wp = w + (size_t)p * n, *cp = c + (size_t)p * n, *sp = s + (size_t)p * n;
    double rr = wp[n - 1];
    for (int i = n - 2; i >= 0; --i) {
        double ci, si, t;
        givens_dev(wp[i], rr, ci, si, t);
        cp[i] = ci; sp[i] = si;
        rr = t;
    }
    wp[0] = rr;
}

// DQROT: rotations on adjacent columns of Q, thread per row.  backward: pairs n-2 .. 0, else 0 .. n-2.
static __global__ void __launch_bounds__(256)
k_qn_rot_q(int n, double *__restrict__ Q, const double *__restrict__ c, const double *__restrict__ s, int backward,
        const LmState *__restrict__ gst, int gwant)
{
    extern __shared__ double cs[];                 // c[n], s[n]
    const int p = blockIdx.y;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    for (int k = threadIdx.x; k < n; k += 256) { cs[k] = c[(size_t)p * n + k]; cs[n + k] = s[(size_t)p * n + k]; }
    __syncthreads();
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= n || n < 2) return;
    double *q = Q + (size_t)p * n * n + row;
    // (Round 4: the row's entries are fetched eight at a time -- unconditional, clamped -- instead of one trip to L2 per
    // rotation: n = 1024 272 -> 50 us.  A store of one step never touches an entry a later step still has to read.)
    if (backward) {
        double y = q[(size_t)(n - 1) * n];
        for (int i0 = n - 2; i0 >= 0; i0 -= 8) {
            double xs[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) xs[u] = q[(size_t)(i0 - u >= 0 ? i0 - u : 0) * n];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 - u;
                asm volatile("" : "+v"(xs[u]));
                if (i >= 0) {
                    const double x = xs[u];
                    const double ci = cs[i], si = cs[n + i];
                    q[(size_t)(i + 1) * n] = ci * y - si * x;
                    y = ci * x + si * y;
                }
            }
        }
        q[0] = y;
    } else {
        double x = q[0];
        for (int i0 = 0; i0 < n - 1; i0 += 8) {
            double ys[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) ys[u] = q[(size_t)(i0 + 1 + u < n ? i0 + 1 + u : n - 1) * n];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u;
                asm volatile("" : "+v"(ys[u]));
                if (i < n - 1) {
                    const double y = ys[u];
                    const double ci = cs[i], si = cs[n + i];
                    q[(size_t)i * n] = ci * x + si * y;
                    x = ci * y - si * x;
                }
            }
        }
        q[(size_t)(n - 1) * n] = x;
    }
}

// DQRQH + the first-row update: R -> upper Hessenberg, then R(0,:) += w0 v^T.  Thread per column.
static __global__ void __launch_bounds__(256)
k_qn_hess_r(int n, double *__restrict__ Rt, const double *__restrict__ c, const double *__restrict__ s,
            const double *__restrict__ w, const double *__restrict__ v,
        const LmState *__restrict__ gst, int gwant)
{
    extern __shared__ double cs[];
    const int p = blockIdx.y;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    for (int k = threadIdx.x; k < n; k += 256) { cs[k] = c[(size_t)p * n + k]; cs[n + k] = s[(size_t)p * n + k]; }
    __syncthreads();
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= n) return;
    double *r = Rt + (size_t)p * n * n + col;
    const int ii = col < n - 2 ? col : n - 2;
    double top;
    if (ii >= 0) {
        double t = r[(size_t)(ii + 1) * n];
        for (int j0 = ii; j0 >= 0; j0 -= 8) {       // (eight entries of the column per trip to L2, as in k_qn_rot_q)
            double rs[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) rs[u] = r[(size_t)(j0 - u >= 0 ? j0 - u : 0) * n];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 - u;
                asm volatile("" : "+v"(rs[u]));
                if (j >= 0) {
                    const double rj = rs[u];
                    const double cj = cs[j], sj = cs[n + j];
                    r[(size_t)(j + 1) * n] = cj * t - sj * rj;
                    t = cj * rj + sj * t;
                }
            }
        }
        top = t;
    } else {
        top = r[0];
    }
    r[0] = top + w[(size_t)p * n] * v[(size_t)p * n + col];
}

// DQHQR: back to upper triangular.  One workgroup per problem, thread per column (NC columns per thread
// when n > blockDim): step j, the owner of column j turns (t, R(j+1,j)) into rotation j and publishes it,
// then every later column applies it.  c, s receive the rotations (dynamic LDS: 2n doubles).
template <int NC>
__global__ void __launch_bounds__(1024)
k_qn_retri(int n, double *__restrict__ Rt, double *__restrict__ c, double *__restrict__ s,
        const LmState *__restrict__ gst, int gwant)
{
    extern __shared__ double cs[];
    const int p = blockIdx.x, tid = threadIdx.x, BS = blockDim.x;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    double *R = Rt + (size_t)p * n * n;
    // Round 4: a thread's "entry below" of step j is simply the next entry of its own column, so the column is streamed
    // eight rows ahead (unconditional, clamped loads), and the barrier of a step waits for LDS only -- the rotations are
    // the only thing threads exchange; the form before paid a trip to L2 in every one of the n - 1 steps (the barrier
    // waited for the prefetch it had just issued: 0.65 us a step).
    constexpr int D = NC == 1 ? 8 : 2;                         // rows ahead (registers: 2 * NC * D doubles)
    double t[NC], cur[NC][D], nxt[NC][D];
    int colc[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int col = tid + q * BS;
        colc[q] = col < n ? col : n - 1;
        t[q] = R[colc[q]];                                        // R(0, col)
#pragma unroll
        for (int u = 0; u < D; ++u) cur[q][u] = R[(size_t)(1 + u < n ? 1 + u : n - 1) * n + colc[q]];     // R(1 .. D, col)
    }
    for (int j0 = 0; j0 < n - 1; j0 += D) {
#pragma unroll
        for (int q = 0; q < NC; ++q)
#pragma unroll
            for (int u = 0; u < D; ++u) nxt[q][u] = R[(size_t)(j0 + D + 1 + u < n ? j0 + D + 1 + u : n - 1) * n + colc[q]];
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int j = j0 + u;
            if (j < n - 1) {                                      // uniform
                const int oq = j / BS, ot = j - oq * BS;          // owner of column j
                if (tid == ot) {
#pragma unroll
                    for (int q = 0; q < NC; ++q)
                        if (q == oq) {
                            double cj, sj, rd;
                            givens_dev(t[q], cur[q][u], cj, sj, rd);
                            cs[j] = cj; cs[n + j] = sj;
                            R[(size_t)j * n + j] = rd;
                            R[(size_t)(j + 1) * n + j] = 0.0;
                        }
                }
                nlh_lds_barrier();
                const double cj = cs[j], sj = cs[n + j];
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    const int col = tid + q * BS;
                    if (col > j && col < n) {
                        const double below = cur[q][u];
                        R[(size_t)j * n + col] = cj * t[q] + sj * below;
                        t[q] = cj * below - sj * t[q];
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NC; ++q)
#pragma unroll
            for (int u = 0; u < D; ++u) { asm volatile("" : "+v"(nxt[q][u])); cur[q][u] = nxt[q][u]; }
    }
    {                                                             // last column keeps its carried value
        const int col = n - 1, oq = col / BS, ot = col - oq * BS;
        if (tid == ot) {
#pragma unroll
            for (int q = 0; q < NC; ++q)
                if (q == oq) R[(size_t)col * n + col] = t[q];
        }
    }
    __syncthreads();
    for (int k = tid; k < n; k += BS) { c[(size_t)p * n + k] = cs[k]; s[(size_t)p * n + k] = cs[n + k]; }
}

// x <- R^-1 x, column oriented (DTRSV 'U','N','N'); R row-major.  Dynamic LDS: n doubles.
// Round 4: BLOCKED like k_lu_solve's back substitution (nlh_kernels_lu.h): columns sixteen at a time, one wave solves the
// 16 x 16 triangle out of LDS (the solved entry of a step reaches the other lanes by v_readlane, no barrier), every thread
// then applies the block's sixteen solved entries to its own row, whose entries (contiguous in the row-major R) were
// fetched a block ahead; barriers wait for LDS only.  The column-at-a-time form paid a trip to L2 and two barriers per
// column (n = 256: 96 us).  Every x(i) still receives x(i) - x(j) r(i,j) for j descending, skipped where x(j) is exactly
// zero, and x(j) = x(j) / r(j,j) is the same division: the same bits.
static __global__ void __launch_bounds__(1024)
k_qn_solve_upper(int n, const double *__restrict__ Rt, double *__restrict__ xall, size_t stride_r, size_t stride_x,
        const LmState *__restrict__ gst, int gwant)
{
    constexpr int W = 16;
    extern __shared__ double bs[];
    __shared__ __attribute__((aligned(16))) double tri[2][W * W];     // tri[.][j * W + l] = r(jb + l, jb + j)
    __shared__ int flag;
    const int p = blockIdx.x, tid = threadIdx.x, BS = blockDim.x, lane = tid & 63, wid = tid >> 6;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    const double *R = Rt + (size_t)p * stride_r;   // leading n x n block, row-major with leading dimension n
    double *x = xall + (size_t)p * stride_x;
    const int nblk = (n + W - 1) / W;
    const unsigned rc = tid < n ? tid : n - 1;     // this thread's first row (clamped)
    double pf[W], pf1[W], t1[4];
    auto issue = [&](int kb) {                      // loads of block kb (columns kb * W ...): unconditional, clamped
        const int kk = kb < 0 ? 0 : kb, jb = kk * W, w = n - jb < W ? n - jb : W;
        const double *row = R + (size_t)rc * n + jb;
#pragma unroll
        for (int j = 0; j < W; ++j) pf1[j] = row[j < w ? j : w - 1];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = (tid + u * BS) & (W * W - 1), j = e / W, l = e % W;
            t1[u] = R[(size_t)(jb + (l < w ? l : w - 1)) * n + jb + (j < w ? j : w - 1)];
        }
    };
    auto land = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("" : "+v"(t1[u]));
            if (tid + u * BS < W * W) tri[buf][tid + u * BS] = t1[u];
        }
#pragma unroll
        for (int j = 0; j < W; ++j) { asm volatile("" : "+v"(pf1[j])); pf[j] = pf1[j]; }
    };
    issue(nblk - 1);
    for (int i = tid; i < n; i += BS) bs[i] = x[i];
    land(0);
    int buf = 0;
    for (int kb = nblk - 1; kb >= 0; --kb, buf ^= 1) {
        const int jb = kb * W, w = n - jb < W ? n - jb : W;
        issue(kb - 1);
        nlh_lds_barrier();
        const double *tr = tri[buf];
        if (wid == 0) {
            const int l = lane & (W - 1);
            double bl = (lane < w) ? bs[jb + lane] : 0.0;
            const double dl = tr[l * W + l];
            double trv[W];                                       // (all LDS reads of the triangle in flight together)
#pragma unroll
            for (int j = 0; j < W; ++j) trv[j] = tr[j * W + l];
            unsigned ran = 0u;
#pragma unroll
            for (int j = W - 1; j >= 0; --j) {
                const double raw = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(bl), j), __builtin_amdgcn_readlane(__double2loint(bl), j));
                const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(dl), j), __builtin_amdgcn_readlane(__double2loint(dl), j));
                const bool go = raw != 0.0;                       // (uniform) the column loop skips an exactly zero x(j)
                const double xj = raw / d;
                const double t = bl - xj * trv[j];
                bl = (go && lane < j) ? t : ((go && lane == j) ? xj : bl);
                ran |= go ? (1u << j) : 0u;
            }
            if (lane < w) bs[jb + lane] = bl;
            if (lane == 0) flag = (int)ran;
        }
        nlh_lds_barrier();
        const unsigned ran = (unsigned)flag;
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
                    if ((ran >> j) & 1u) bi = bi - bs[jb + j] * R[(size_t)i * n + jb + j];
            }
            bs[i] = bi;
        }
        land(buf ^ 1);
    }
    __syncthreads();
    for (int i = tid; i < n; i += BS) x[i] = bs[i];
}

// Vandermonde panel of polynomial%fit (src/nonlin_polynomials.f90:177-184, :222-225), row-major npts x ncols:
// column c = column c-1 * x, one thread per point (the products chain along the row).  Also copies y to rhs.
static __global__ void __launch_bounds__(256)
k_vandermonde(int npts, int ncols, int thru_zero, const double *__restrict__ x, const double *__restrict__ y,
              double *__restrict__ A, double *__restrict__ rhs)
{
    const int p = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= npts) return;
    const double xj = x[(size_t)p * npts + j];
    double *row = A + ((size_t)p * npts + j) * ncols;
    double a;
    int c;
    if (thru_zero) { a = xj; row[0] = a; c = 1; }
    else { row[0] = 1.0; a = xj; if (ncols > 1) row[1] = a; c = 2; }
    for (; c < ncols; ++c) { a = a * xj; row[c] = a; }
    rhs[(size_t)p * npts + j] = y[(size_t)p * npts + j];
}


// Dynamic-LDS limits of this header's kernels, for the copies of the translation unit that includes it (called once per
// device from that unit's init function).
static void broyden_kernel_attrs(int lds_max)
{
    hipFuncSetAttribute((const void *)k_qn_house_dot<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_house_dot2, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_house_fused<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_house_fused<16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_matvec_cm, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_rot_q, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_hess_r, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_retri<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_retri<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_retri<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_solve_upper, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    hipFuncSetAttribute((const void *)k_qn_resid, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
}
