// nlh_kernels_gram.h -- G = J^T J with v_mfma_f64_16x16x4_f64 and g = J^T f.
//
// Replaces the O(m n^2) part of lmfactor (src/nonlin_least_squares.f90:569-667) and the
// Q^T f sweep (:241-253) on the normal-equations path: with P^T G P = R^T R,
// R equals lmfactor's R up to row signs and qtf = R^-T P^T g.
//
// Layout: J column-major m-by-n, so a G entry contracts two contiguous columns.  A
// workgroup owns one 64x64 block of the lower block-triangle of G and one K-split of the
// rows; J tiles (32 rows x 64 columns) are read column-coalesced and staged in LDS with a
// 34-double column pitch (34 = 2 mod 32), which makes the MFMA
// operand reads (ds_read_b64, lanes 0-15 = 16 columns, lane>>4 = k) bank-conflict-free.
// Split-K partials go to a slab and are summed in a fixed order: bitwise reproducible.
#pragma once
#include "nlh_common.h"

typedef double v4d __attribute__((ext_vector_type(4)));

#define GRAM_BT 64          // block edge in columns of J
#define GRAM_KT 32          // rows of J per LDS tile
#define GRAM_LD (GRAM_KT + 2)

__device__ __forceinline__ void gram_block_index(int idx, int &bi, int &bj)
{
    // idx enumerates the lower block triangle row by row: (0,0),(1,0),(1,1),(2,0)...
    int r = 0;
    while ((r + 1) * (r + 2) / 2 <= idx) ++r;
    bi = r;
    bj = idx - r * (r + 1) / 2;
}

static __global__ void __launch_bounds__(256, 3)
k_gram_mfma(int m, int n, int rows_per_split, const double *__restrict__ J,
            double *__restrict__ Gpart /* [nprob][nsplit][n*n] */,
            const double *__restrict__ f /* [nprob][m] or null */,
            double *__restrict__ gpart /* [nprob][nsplit][n] */,
            const LmState *__restrict__ st, int want_stage, int nblk_arg, int nsplit_arg, int nprob_arg)
{
    __shared__ double tA[GRAM_BT * GRAM_LD];
    __shared__ double tB[GRAM_BT * GRAM_LD];
    __shared__ double fs[GRAM_KT];
    // XCD-aware mapping.  Workgroups are dealt round-robin over the 8 XCDs (linear id % 8), each with
    // its own L2.  The nblk blocks of one (problem, K-split) item read the same rows of J, so they
    // are given ids  g*8*nblk + blk*8 + xcd  (item = g*8 + xcd): same XCD, adjacent in dispatch
    // order, and J comes from HBM once per item instead of once per block.  Placement is a
    // performance hint only; results do not depend on it.
    const int nblk = nblk_arg, nsplit = nsplit_arg;
    const long L = blockIdx.x;
    const long grp = L / (8L * nblk);
    const int within = (int)(L % (8L * nblk));
    const long item = grp * 8 + (within & 7);
    const int blk = within >> 3;
    if (item >= (long)nsplit * nprob_arg) return;
    const int p = (int)(item / nsplit);
    const int split = (int)(item % nsplit);
    if (st && st[p].stage != want_stage) return;
    int bi, bj;
    gram_block_index(blk, bi, bj);
    const int kbeg = split * rows_per_split;
    const int kend = min(m, kbeg + rows_per_split);
    const double *Jp = J + (size_t)p * m * n;
    const bool diag = (bi == bj);
    const double *tBp = diag ? tA : tB;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;          // 2x2 waves, 32x32 outputs each
    const int lr = tid % GRAM_KT, lc0 = tid / GRAM_KT;   // tile loader: row lr, columns lc0 + LSTEP*cc
    constexpr int LSTEP = 256 / GRAM_KT;

    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = (v4d){0.0, 0.0, 0.0, 0.0};

    // 16 x 16 MFMA tiles of this wave: A-side tile rows ai[0..1], B-side tile columns bi_[0..1] (in units of 16
    // columns of the block), products acc[x][y] = tile (ai[x], bi_[y]).  Off the diagonal a wave owns a 2 x 2 patch.
    // In a diagonal block only tiles on or below the diagonal are needed (k_gram_reduce mirrors elementwise), and they
    // are dealt so that no wave has more than three: the block costs 3/4 of the MFMAs.
    int ai0 = 2 * wr, ai1 = 2 * wr + 1, bt0 = 2 * wc, bt1 = 2 * wc + 1;
    bool m00 = true, m01 = true, m10 = true, m11 = true;
    if (diag) {
        if (w == 0) { ai0 = 0; ai1 = 1; bt0 = 0; bt1 = 1; m01 = false; }                       // (0,0) (1,0) (1,1)
        else if (w == 1) { ai0 = 2; ai1 = 2; bt0 = 0; bt1 = 1; m10 = false; m11 = false; }      // (2,0) (2,1)
        else if (w == 2) { ai0 = 3; ai1 = 3; bt0 = 0; bt1 = 1; m10 = false; m11 = false; }      // (3,0) (3,1)
        else { ai0 = 2; ai1 = 3; bt0 = 2; bt1 = 3; m01 = false; }                               // (2,2) (3,2) (3,3)
    }
    // LDS byte addresses of this lane's operands (the low 32 bits of a generic LDS pointer)
    const int lrow = (lane & 15) * GRAM_LD + (lane >> 4);
    const unsigned aaddr0 = (unsigned)(size_t)(tA + ai0 * 16 * GRAM_LD + lrow);
    const unsigned aaddr1 = (unsigned)(size_t)(tA + ai1 * 16 * GRAM_LD + lrow);
    const unsigned baddr0 = (unsigned)(size_t)(tBp + bt0 * 16 * GRAM_LD + lrow);
    const unsigned baddr1 = (unsigned)(size_t)(tBp + bt1 * 16 * GRAM_LD + lrow);
#define GRAM_LDS_READ4(x0, x1, y0, y1, OFF)                                                          \
    asm volatile("ds_read_b64 %0, %4 offset:%8\n\tds_read_b64 %1, %5 offset:%8\n\t"                 \
                 "ds_read_b64 %2, %6 offset:%8\n\tds_read_b64 %3, %7 offset:%8"                      \
                 : "=&v"(x0), "=&v"(x1), "=&v"(y0), "=&v"(y1)                                       \
                 : "v"(aaddr0), "v"(aaddr1), "v"(baddr0), "v"(baddr1), "n"(OFF)                     \
                 : "memory");
#define GRAM_LDS_WAIT4(x0, x1, y0, y1)                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(y0), "+v"(y1));

    // Software pipeline: the global loads of tile t+1 are issued into registers right after the
    // barrier that publishes tile t in LDS, so their latency overlaps the MFMA phase of tile t.
    constexpr int NLD = GRAM_BT / LSTEP;        // loads per thread per tile
    double ra[NLD], rb[NLD];
    auto load_tile = [&](int k0) {
        const int row = k0 + lr;
        const bool rok = row < kend;
#pragma unroll
        for (int cc = 0; cc < NLD; ++cc) {
            const int col = lc0 + LSTEP * cc;
            const int ja = bi * GRAM_BT + col;
            ra[cc] = (rok && ja < n) ? Jp[(size_t)ja * m + row] : 0.0;
            if (!diag) {
                const int jb = bj * GRAM_BT + col;
                rb[cc] = (rok && jb < n) ? Jp[(size_t)jb * m + row] : 0.0;
            }
        }
    };
    // g = J^T f rides along in the diagonal blocks (VALU beside the MFMA pipe): thread (column
    // tid & 63, k-quarter tid >> 6) accumulates its share of every tile; fixed-order combine at the end.
    const bool dog = diag && f != nullptr;
    const double *fp = dog ? f + (size_t)p * m : nullptr;
    double gacc = 0.0;
    if (kbeg < kend) load_tile(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += GRAM_KT) {
#pragma unroll
        for (int cc = 0; cc < NLD; ++cc) {
            const int col = lc0 + LSTEP * cc;
            tA[col * GRAM_LD + lr] = ra[cc];
            if (!diag) tB[col * GRAM_LD + lr] = rb[cc];
        }
        if (dog && tid < GRAM_KT) fs[tid] = (k0 + tid < kend) ? fp[k0 + tid] : 0.0;
        __syncthreads();
        if (k0 + GRAM_KT < kend) load_tile(k0 + GRAM_KT);
        if (dog) {
            const int gc = tid & 63, gq = (tid >> 6) * (GRAM_KT / 4);
#pragma unroll
            for (int i = 0; i < GRAM_KT / 4; ++i) gacc = gacc + tA[gc * GRAM_LD + gq + i] * fs[gq + i];
        }
        // Operand reads are issued as ds_read_b64 by hand (two 32-lane groups over 64 banks: conflict-free with
        // this pitch, 256 B/clk).  Left to the compiler the reads of neighbouring k-steps are paired into
        // ds_read2_b64, which is serviced in 16-lane groups over 32 banks -- 2-way conflicts here and half the
        // bandwidth (PMC: 42 % of the LDS cycles were conflict cycles).  The operands of step ks+1 are requested
        // before the MFMAs of step ks are issued.
        double a0, a1, b0, b1;
        GRAM_LDS_READ4(a0, a1, b0, b1, 0)
        GRAM_LDS_WAIT4(a0, a1, b0, b1)
#pragma unroll
        for (int ks = 0; ks < GRAM_KT / 4; ++ks) {
            double na0 = 0.0, na1 = 0.0, nb0 = 0.0, nb1 = 0.0;
            if (ks + 1 < GRAM_KT / 4) GRAM_LDS_READ4(na0, na1, nb0, nb1, (ks + 1) * 32)
            if (m00) acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            if (m01) acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            if (m10) acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            if (m11) acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
            if (ks + 1 < GRAM_KT / 4) {
                GRAM_LDS_WAIT4(na0, na1, nb0, nb1)
                a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
            }
        }
        __syncthreads();
    }

#undef GRAM_LDS_READ4
#undef GRAM_LDS_WAIT4
    if (dog) {                                  // tB is unused by diagonal blocks: combine the four k-quarters
        tB[tid] = gacc;
        __syncthreads();
        if (tid < 64) {
            const int jg = bi * GRAM_BT + tid;
            if (jg < n)
                gpart[((size_t)p * nsplit + split) * n + jg] = ((tB[tid] + tB[tid + 64]) + tB[tid + 128]) + tB[tid + 192];
        }
    }
    // f64 16x16x4 C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg.
    double *Gp = Gpart + ((size_t)p * nsplit + split) * (size_t)n * n;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const bool on = (a == 0) ? (c == 0 ? m00 : m01) : (c == 0 ? m10 : m11);
            if (!on) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = bi * GRAM_BT + (a == 0 ? ai0 : ai1) * 16 + (lane >> 4) + 4 * r;   // A-side column of J
                const int gc = bj * GRAM_BT + (c == 0 ? bt0 : bt1) * 16 + (lane & 15);           // B-side column of J
                if (gr < n && gc < n) Gp[(size_t)gr * n + gc] = acc[a][c][r];     // ROW-major slab: see k_gram_reduce
            }
        }
}

// ---------------------------------------------------------------------------------------------
// 224 < n <= 256: one 512-thread workgroup per (problem, K-split) item forms the WHOLE lower triangle.
// k_gram_mfma gives every 64x64 block of G its own workgroup, so the rows of J are staged ~4 times per item (PMC:
// 26.8 MB fetched per 4096x256 problem against 8.4 MB of J) and a barrier pair covers only 32 MFMAs per wave.
// Here the 32-row tile of all 256 columns is staged once (69.6 KB of LDS) and the 136 MFMA tiles of the lower
// triangle are dealt 17 to a wave: wave W takes tile rows 15-W and W (16-W and W+1 tiles, whose column operands
// coincide), i.e. 2 + 16-W operand reads for 17 MFMAs per k-step and 136 MFMAs per wave between barriers.
// Same accumulation order as k_gram_mfma (rows ascending inside a split, splits summed by k_gram_reduce), same
// output slabs: bitwise the same G and g.
#define GRAM_TN 256
// NT = tile rows of the triangle: 16 (n <= 256, 8 waves) or 8 (n <= 128, 4 waves, same scheme at half the size).
template <int W, int NT>
__device__ __forceinline__ void gram_tri_wave(double *tA, double *fs, int kbeg, int kend, int m, int n,
                                              const double *Jp, const double *fp, double *Gp, double *gout, double *gscr, bool direct,
                                              int ldg = 0)
{   // n: columns of THIS panel (Jp points at its first column); ldg: leading dimension of G when the panel is a diagonal
    // block of a wider matrix (Gp then points at the block's (0, 0) entry), 0: n
    if (ldg == 0) ldg = n;
    constexpr int NL = NT - W, NS = W + 1, RL = NT - 1 - W, RS = W;  // long / short tile row of this wave
    constexpr int NTH = 32 * NT, TN = 16 * NT, HALF = NTH / 2;       // threads, columns, columns again
    const int tid = threadIdx.x, lane = tid & 63;
    const int lr = tid % GRAM_KT, lc0 = tid / GRAM_KT;                // loader: row lr, columns lc0 + 16*cc
    constexpr int LSTEP = NTH / GRAM_KT, NLD = TN / LSTEP;
    v4d accL[NL], accS[NS];
#pragma unroll
    for (int c = 0; c < NL; ++c) accL[c] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < NS; ++c) accS[c] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int lrow = (lane & 15) * GRAM_LD + (lane >> 4);
    const unsigned baddr = (unsigned)(size_t)(tA + lrow);
    const unsigned aaddrL = (unsigned)(size_t)(tA + RL * 16 * GRAM_LD + lrow);
    const unsigned aaddrS = (unsigned)(size_t)(tA + RS * 16 * GRAM_LD + lrow);
    double ra[NLD];
    auto load_tile = [&](int k0) {
        const int row = k0 + lr;
        const bool rok = row < kend;
#pragma unroll
        for (int cc = 0; cc < NLD; ++cc) {
            const int col = lc0 + LSTEP * cc;
            ra[cc] = (rok && col < n) ? Jp[(size_t)col * m + row] : 0.0;
        }
    };
    // g = J^T f: column tid & 255, two quarters of the tile per thread (the same four partial sums per column as
    // k_gram_mfma keeps in four threads)
    const int gc = tid & (HALF - 1), gh = tid / HALF;
    double gq0 = 0.0, gq1 = 0.0;
    if (kbeg < kend) load_tile(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += GRAM_KT) {
#pragma unroll
        for (int cc = 0; cc < NLD; ++cc) tA[(lc0 + LSTEP * cc) * GRAM_LD + lr] = ra[cc];
        if (fp && tid < GRAM_KT) fs[tid] = (k0 + tid < kend) ? fp[k0 + tid] : 0.0;
        __syncthreads();
        if (k0 + GRAM_KT < kend) load_tile(k0 + GRAM_KT);
        if (fp) {
            const double *col = tA + gc * GRAM_LD + gh * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) gq0 = gq0 + col[i] * fs[gh * 16 + i];
#pragma unroll
            for (int i = 0; i < 8; ++i) gq1 = gq1 + col[8 + i] * fs[gh * 16 + 8 + i];
        }
#pragma unroll
        for (int ks = 0; ks < GRAM_KT / 4; ++ks) {
            double aL, aS, b[NL];
            asm volatile("ds_read_b64 %0, %2 offset:%4\n\tds_read_b64 %1, %3 offset:%4"
                         : "=&v"(aL), "=&v"(aS) : "v"(aaddrL), "v"(aaddrS), "n"(ks * 32) : "memory");
#pragma unroll
            for (int c = 0; c < NL; ++c)
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(b[c]) : "v"(baddr), "n"(c * 16 * GRAM_LD * 8 + ks * 32) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(aL), "+v"(aS));
#pragma unroll
            for (int c = 0; c < NL; ++c) asm volatile("" : "+v"(b[c]));
#pragma unroll
            for (int c = 0; c < NL; ++c) {
                accL[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aL, b[c], accL[c], 0, 0, 0);
                if (c < NS) accS[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aS, b[c], accS[c], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (fp) {
        gscr[tid] = gq0;
        gscr[NTH + tid] = gq1;
        __syncthreads();
        if (tid < HALF && tid < n)
            gout[tid] = ((gscr[tid] + gscr[NTH + tid]) + gscr[HALF + tid]) + gscr[NTH + HALF + tid];
    }
    // f64 16x16x4 C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg.
#pragma unroll
    for (int c = 0; c < NL; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = RL * 16 + (lane >> 4) + 4 * r, gcol = c * 16 + (lane & 15);
            if (gr < n && gcol < n) {
                if (!direct) Gp[(size_t)gr * ldg + gcol] = accL[c][r];           // (slab: row-major, k_gram_reduce)
                else if (gr >= gcol) { Gp[(size_t)gcol * ldg + gr] = accL[c][r]; Gp[(size_t)gr * ldg + gcol] = accL[c][r]; }
            }
        }
#pragma unroll
    for (int c = 0; c < NS; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = RS * 16 + (lane >> 4) + 4 * r, gcol = c * 16 + (lane & 15);
            if (gr < n && gcol < n) {
                if (!direct) Gp[(size_t)gr * ldg + gcol] = accS[c][r];
                else if (gr >= gcol) { Gp[(size_t)gcol * ldg + gr] = accS[c][r]; Gp[(size_t)gr * ldg + gcol] = accS[c][r]; }
            }
        }
}

template <int NT>
__global__ void __launch_bounds__(32 * NT)
k_gram_tri(int m, int n, int rows_per_split, const double *__restrict__ J, double *__restrict__ Gpart,
           const double *__restrict__ f, double *__restrict__ gpart, const LmState *__restrict__ st, int want_stage,
           int nsplit, double *__restrict__ Gdirect, double *__restrict__ gdirect)
{
    // Gdirect != null (only with nsplit == 1): there is nothing to sum, so G (lower triangle mirrored, exactly what
    // k_gram_reduce would produce) and g are written in place and the reduce launch is skipped.
    extern __shared__ double gsm[];
    double *tA = gsm;                                   // 16 NT * GRAM_LD
    double *fs = tA + 16 * NT * GRAM_LD;                // GRAM_KT
    double *gscr = fs + GRAM_KT;                        // 64 NT
    const int item = blockIdx.x, p = item / nsplit, split = item % nsplit;
    if (st && st[p].stage != want_stage) return;
    const int kbeg = split * rows_per_split, kend = min(m, kbeg + rows_per_split);
    const double *Jp = J + (size_t)p * m * n;
    const double *fp = f ? f + (size_t)p * m : nullptr;
    const bool direct = Gdirect != nullptr;
    double *Gp = direct ? Gdirect + (size_t)p * n * n : Gpart + ((size_t)p * nsplit + split) * (size_t)n * n;
    double *gout = direct ? (gdirect ? gdirect + (size_t)p * n : nullptr) : gpart + ((size_t)p * nsplit + split) * n;
    const int wv = threadIdx.x >> 6;
    if constexpr (NT == 16) {
        switch (wv) {
        case 0: gram_tri_wave<0, 16>(tA, fs, kbeg, kend, m, n, Jp, fp, Gp, gout, gscr, direct); break;
        case 1: gram_tri_wave<1, 16>(tA, fs, kbeg, kend, m, n, Jp, fp, Gp, gout, gscr, direct); break;
        case 2: gram_tri_wave<2, 16>(tA, fs, kbeg, kend, m, n, Jp, fp, Gp, gout, gscr, direct); break;
        case 3: gram_tri_wave<3, 16>(tA, fs, kbeg, kend, m, n, Jp, fp, Gp, gout, gscr, direct); break;
        case 4: gram_tri_wave<4, 16>(tA, fs, kbeg, kend, m, n, Jp, fp, Gp, gout, gscr, direct); break;
        case 5: gram_tri_wave<5, 16>(tA, fs, kbeg, kend, m, n, Jp, fp, Gp, gout, gscr, direct); break;
        case 6: gram_tri_wave<6, 16>(tA, fs, kbeg, kend, m, n, Jp, fp, Gp, gout, gscr, direct); break;
        default: gram_tri_wave<7, 16>(tA, fs, kbeg, kend, m, n, Jp, fp, Gp, gout, gscr, direct); break;
        }
    } else {
        switch (wv) {
        case 0: gram_tri_wave<0, 8>(tA, fs, kbeg, kend, m, n, Jp, fp, Gp, gout, gscr, direct); break;
        case 1: gram_tri_wave<1, 8>(tA, fs, kbeg, kend, m, n, Jp, fp, Gp, gout, gscr, direct); break;
        case 2: gram_tri_wave<2, 8>(tA, fs, kbeg, kend, m, n, Jp, fp, Gp, gout, gscr, direct); break;
        default: gram_tri_wave<3, 8>(tA, fs, kbeg, kend, m, n, Jp, fp, Gp, gout, gscr, direct); break;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 256 < n <= 512 (BASELINE config 5: 65536 x 512).  The lower triangle of G no longer fits one CU's registers (528 tiles
// of 16 x 16), so a (problem, K-split) item is FOUR workgroups of about equal work, each staging only the columns it
// needs: unit 0 / 1 = the diagonal 256-column blocks (gram_tri_wave on columns 0..255 / 256..n-1: 136 tiles), unit 2 / 3 =
// the upper / lower half of the off-diagonal square G(256.., 0..255) (128 tiles each: a 32-row tile of columns 0..255 and
// of 128 columns of the second panel in LDS, 104 KB; eight waves, a wave two tile rows x eight tile columns: 2 + 8
// operand reads for 16 MFMAs per k-step, 128 MFMAs between barriers).  J is staged 3 times per item where k_gram_mfma's
// 64 x 64 blocks stage it 9 times, and a barrier pair covers 128-136 MFMAs per wave instead of 32.
// Same accumulation order (rows ascending inside a split, splits summed by k_gram_reduce), same slabs: bitwise the same
// G and g as k_gram_mfma.
// ---------------------------------------------------------------------------------------------
// Round 5: the two wave bodies of k_gram_512 with DOUBLE-BUFFERED 16-row tiles and ONE barrier per tile.
// With one 512-thread workgroup per CU (BASELINE config 5: 64 K-splits x 4 units = 256 workgroups) nothing hides a
// barrier: all eight waves -- both waves of every SIMD -- leave it together, write the next tile, issue their loads and
// wait for their first operands at the same moment, and the MFMA pipe idles meanwhile (362 us for 232 us of MFMA time).
// Here the tile being multiplied and the tile being written are different buffers: the LDS writes of tile t + 1 and the
// global loads of tile t + 2 are issued in the middle of the MFMA phase of tile t, and the one barrier at the end of the
// phase both publishes tile t + 1 and retires tile t.  Rows are still accumulated in ascending groups of four inside a
// split and g = J^T f keeps its four partial sums per column (rows mod 32 in [0,8), [8,16), [16,24), [24,32)): the same
// bits as the single-buffered bodies and as k_gram_mfma.
#define GRAM_KT2 16
#define GRAM_LD2 (GRAM_KT2 + 2)
template <int W>
__device__ __forceinline__ void gram_tri_wave_db(double *tA0, double *fs0, int kbeg, int kend, int m, int n, const double *Jp,
                                                 const double *fp, double *Gp, double *gout, double *gscr, int ldg)
{
    constexpr int NT = 16, NL = NT - W, NS = W + 1, RL = NT - 1 - W, RS = W;
    constexpr int NTH = 512, TN = 256;
    constexpr int TSZ = TN * GRAM_LD2;                               // doubles per tile buffer
    double *tA1 = tA0 + TSZ;
    (void)fs0; (void)gscr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int lr = tid % GRAM_KT2, lc0 = tid / GRAM_KT2;             // loader: row lr, columns lc0 + 32 cc
    constexpr int LSTEP = NTH / GRAM_KT2, NLD = TN / LSTEP;
    v4d accL[NL], accS[NS];
#pragma unroll
    for (int c = 0; c < NL; ++c) accL[c] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < NS; ++c) accS[c] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int lrow = (lane & 15) * GRAM_LD2 + (lane >> 4);
    // (Loads stay under their conditions: made unconditional through clamped addresses with the value selected afterwards,
    // the compiler placed each select -- and a wait for its load -- right behind the load: 350 -> 388 us.)
    // Whole tiles and a whole panel (uniform for the workgroup: every split of BASELINE config 5): plain loads, nothing to select.
    double ra[NLD], rf = 0.0;
    const bool whole = ((kend - kbeg) % GRAM_KT2 == 0) && n == TN;
    const double *c0p = Jp + (size_t)lc0 * m + lr;
    auto load_tile = [&](int k0) {
        if (whole) {
#pragma unroll
            for (int cc = 0; cc < NLD; ++cc) ra[cc] = c0p[(size_t)(LSTEP * cc) * m + k0];
            if (fp) rf = fp[k0 + lr];
            return;
        }
        const int row = k0 + lr;
        const bool rok = row < kend;
#pragma unroll
        for (int cc = 0; cc < NLD; ++cc) {
            const int col = lc0 + LSTEP * cc;
            ra[cc] = (rok && col < n) ? Jp[(size_t)col * m + row] : 0.0;
        }
        if (fp) rf = rok ? fp[row] : 0.0;
    };
    auto put_tile = [&](double *t) {
#pragma unroll
        for (int cc = 0; cc < NLD; ++cc) t[(lc0 + LSTEP * cc) * GRAM_LD2 + lr] = ra[cc];
    };
    // g = J^T f out of the loader's registers: thread (row lr, its NLD columns) adds J(row, col) * f(row) for its rows
    // lr, lr + 16, ... of the split in ascending order -- no LDS traffic, eight multiply / add pairs per tile -- and the
    // sixteen row classes of a column are added in ascending lr at the end (sixteen partial sums per column and split where
    // k_gram_mfma keeps four: the same g to rounding, not to the bit; G is unaffected).
    double gacc[NLD];
#pragma unroll
    for (int cc = 0; cc < NLD; ++cc) gacc[cc] = 0.0;
    auto g_tile = [&]() {
#pragma unroll
        for (int cc = 0; cc < NLD; ++cc) gacc[cc] = gacc[cc] + ra[cc] * rf;
    };
    const int nst = (kend - kbeg + GRAM_KT2 - 1) / GRAM_KT2;
    if (nst > 0) { load_tile(kbeg); put_tile(tA0); if (fp) g_tile(); }
    if (nst > 1) load_tile(kbeg + GRAM_KT2);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        double *cur = (st & 1) ? tA1 : tA0, *nxt = (st & 1) ? tA0 : tA1;
        const unsigned baddr = (unsigned)(size_t)(cur + lrow);
        const unsigned aaddrL = (unsigned)(size_t)(cur + RL * 16 * GRAM_LD2 + lrow);
        const unsigned aaddrS = (unsigned)(size_t)(cur + RS * 16 * GRAM_LD2 + lrow);
#pragma unroll
        for (int ks = 0; ks < GRAM_KT2 / 4; ++ks) {
            double aL, aS, b[NL];
            asm volatile("ds_read_b64 %0, %2 offset:%4\n\tds_read_b64 %1, %3 offset:%4"
                         : "=&v"(aL), "=&v"(aS) : "v"(aaddrL), "v"(aaddrS), "n"(ks * 32) : "memory");
#pragma unroll
            for (int c = 0; c < NL; ++c)
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(b[c]) : "v"(baddr), "n"(c * 16 * GRAM_LD2 * 8 + ks * 32) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(aL), "+v"(aS));
#pragma unroll
            for (int c = 0; c < NL; ++c) asm volatile("" : "+v"(b[c]));
#pragma unroll
            for (int c = 0; c < NL; ++c) {
                accL[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aL, b[c], accL[c], 0, 0, 0);
                if (c < NS) accS[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aS, b[c], accS[c], 0, 0, 0);
            }
#ifndef GRAM_DBG_NOLOAD
            if (ks == 0 && st + 1 < nst) {                          // in the shadow of this tile's MFMAs: the next tile ...
                put_tile(nxt);
#ifndef GRAM_DBG_NOG
                if (fp) g_tile();
#endif
            }
            if (ks == 1 && st + 2 < nst) load_tile(kbeg + (st + 2) * GRAM_KT2);   // ... and the loads of the one after
#endif
        }
#ifndef GRAM_DBG_NOBAR
        __syncthreads();
#endif
    }
    if (fp) {                                                       // the sixteen row classes of every column, ascending
        double *gs = tA0;                                           // (both tile buffers are free: the loop ended on a barrier)
#pragma unroll
        for (int cc = 0; cc < NLD; ++cc) gs[(lc0 + LSTEP * cc) * GRAM_KT2 + lr] = gacc[cc];
        __syncthreads();
        if (tid < TN && tid < n) {
            double sg = 0.0;
#pragma unroll
            for (int i = 0; i < GRAM_KT2; ++i) sg = sg + gs[tid * GRAM_KT2 + i];
            gout[tid] = sg;
        }
    }
#pragma unroll
    for (int c = 0; c < NL; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = RL * 16 + (lane >> 4) + 4 * r, gcol = c * 16 + (lane & 15);
            if (gr < n && gcol < n) Gp[(size_t)gr * ldg + gcol] = accL[c][r];
        }
#pragma unroll
    for (int c = 0; c < NS; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = RS * 16 + (lane >> 4) + 4 * r, gcol = c * 16 + (lane & 15);
            if (gr < n && gcol < n) Gp[(size_t)gr * ldg + gcol] = accS[c][r];
        }
}

template <int W>
__device__ __forceinline__ void gram_sq_wave_db(double *tB0, int kbeg, int kend, int m, int nA, const double *JpB, const double *JpA,
                                                double *Gp, int ldg, const double *fp, double *goutB, double *goutA)
{   // fp != null: g = J^T f rides along HERE (the square units have 128 tiles to the triangle units' 136: the slack pays for it)
    // out of the loader's registers -- goutB != null: for panel 0's 256 columns, goutA: for this unit's own columns
    // one buffer = panel 0's 256 columns followed by this unit's 128 columns of panel 1
    constexpr int RP = W >> 1, CH = W & 1;
    constexpr int TSZ = 384 * GRAM_LD2;
    double *tB1 = tB0 + TSZ;
    const int tid = threadIdx.x, lane = tid & 63;
    const int lr = tid % GRAM_KT2, lc0 = tid / GRAM_KT2;
    constexpr int LSTEP = 512 / GRAM_KT2, NLB = 256 / LSTEP, NLA = 128 / LSTEP;
    v4d acc[2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[a][c] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int lrow = (lane & 15) * GRAM_LD2 + (lane >> 4);
    double rb[NLB], ra[NLA];
    const bool whole = ((kend - kbeg) % GRAM_KT2 == 0) && nA == 128;
    const double *b0p = JpB + (size_t)lc0 * m + lr, *a0p = JpA + (size_t)lc0 * m + lr;
    double rf = 0.0, gaccB[NLB], gaccA[NLA];
#pragma unroll
    for (int cc = 0; cc < NLB; ++cc) gaccB[cc] = 0.0;
#pragma unroll
    for (int cc = 0; cc < NLA; ++cc) gaccA[cc] = 0.0;
    auto g_tile = [&]() {                                           // thread (row lr, its columns): rows lr, lr + 16, ... ascending
        if (goutB) {
#pragma unroll
            for (int cc = 0; cc < NLB; ++cc) gaccB[cc] = gaccB[cc] + rb[cc] * rf;
        }
#pragma unroll
        for (int cc = 0; cc < NLA; ++cc) gaccA[cc] = gaccA[cc] + ra[cc] * rf;
    };
    auto load_tile = [&](int k0) {
        if (whole) {                                                // (see the triangle body)
#pragma unroll
            for (int cc = 0; cc < NLB; ++cc) rb[cc] = b0p[(size_t)(LSTEP * cc) * m + k0];
#pragma unroll
            for (int cc = 0; cc < NLA; ++cc) ra[cc] = a0p[(size_t)(LSTEP * cc) * m + k0];
            if (fp) rf = fp[k0 + lr];
            return;
        }
        const int row = k0 + lr;
        const bool rok = row < kend;
        if (fp) rf = rok ? fp[row] : 0.0;
#pragma unroll
        for (int cc = 0; cc < NLB; ++cc) rb[cc] = rok ? JpB[(size_t)(lc0 + LSTEP * cc) * m + row] : 0.0;
#pragma unroll
        for (int cc = 0; cc < NLA; ++cc) {
            const int col = lc0 + LSTEP * cc;
            ra[cc] = (rok && col < nA) ? JpA[(size_t)col * m + row] : 0.0;
        }
    };
    auto put_tile = [&](double *t) {
#pragma unroll
        for (int cc = 0; cc < NLB; ++cc) t[(lc0 + LSTEP * cc) * GRAM_LD2 + lr] = rb[cc];
#pragma unroll
        for (int cc = 0; cc < NLA; ++cc) t[(256 + lc0 + LSTEP * cc) * GRAM_LD2 + lr] = ra[cc];
    };
    const int nst = (kend - kbeg + GRAM_KT2 - 1) / GRAM_KT2;
    if (nst > 0) { load_tile(kbeg); put_tile(tB0); if (fp) g_tile(); }
    if (nst > 1) load_tile(kbeg + GRAM_KT2);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        double *cur = (st & 1) ? tB1 : tB0, *nxt = (st & 1) ? tB0 : tB1;
        const unsigned baddr = (unsigned)(size_t)(cur + CH * 8 * 16 * GRAM_LD2 + lrow);
        const unsigned aaddr0 = (unsigned)(size_t)(cur + (256 + (2 * RP) * 16) * GRAM_LD2 + lrow);
        const unsigned aaddr1 = (unsigned)(size_t)(cur + (256 + (2 * RP + 1) * 16) * GRAM_LD2 + lrow);
#pragma unroll
        for (int ks = 0; ks < GRAM_KT2 / 4; ++ks) {
            double a0, a1, b[8];
            asm volatile("ds_read_b64 %0, %2 offset:%4\n\tds_read_b64 %1, %3 offset:%4"
                         : "=&v"(a0), "=&v"(a1) : "v"(aaddr0), "v"(aaddr1), "n"(ks * 32) : "memory");
#pragma unroll
            for (int c = 0; c < 8; ++c)
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(b[c]) : "v"(baddr), "n"(c * 16 * GRAM_LD2 * 8 + ks * 32) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1));
#pragma unroll
            for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(b[c]));
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                acc[0][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b[c], acc[0][c], 0, 0, 0);
                acc[1][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b[c], acc[1][c], 0, 0, 0);
            }
#ifndef GRAM_DBG_NOLOAD
            if (ks == 0 && st + 1 < nst) { put_tile(nxt); if (fp) g_tile(); }
            if (ks == 1 && st + 2 < nst) load_tile(kbeg + (st + 2) * GRAM_KT2);
#endif
        }
#ifndef GRAM_DBG_NOBAR
        __syncthreads();
#endif
    }
    if (fp) {                                                       // the sixteen row classes of every column, ascending (tiles are free)
        double *gs = tB0;
#pragma unroll
        for (int cc = 0; cc < NLB; ++cc) gs[(lc0 + LSTEP * cc) * GRAM_KT2 + lr] = gaccB[cc];
#pragma unroll
        for (int cc = 0; cc < NLA; ++cc) gs[(256 + lc0 + LSTEP * cc) * GRAM_KT2 + lr] = gaccA[cc];
        __syncthreads();
        if (tid < 384) {
            const bool isB = tid < 256;
            double *dst = isB ? goutB : goutA;
            const int c = isB ? tid : tid - 256;
            if (dst && (isB || c < nA)) {
                double sg = 0.0;
#pragma unroll
                for (int i = 0; i < GRAM_KT2; ++i) sg = sg + gs[tid * GRAM_KT2 + i];
                dst[c] = sg;
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = (2 * RP + a) * 16 + (lane >> 4) + 4 * r, gcol = (CH * 8 + c) * 16 + (lane & 15);
                if (gr < nA) Gp[(size_t)gr * ldg + gcol] = acc[a][c][r];
            }
}

static __global__ void __launch_bounds__(512)
k_gram_512(int m, int n, int rows_per_split, const double *__restrict__ J, double *__restrict__ Gpart,
           const double *__restrict__ f, double *__restrict__ gpart, const LmState *__restrict__ st, int want_stage,
           int nsplit, int nprob)
{
    extern __shared__ double gsm[];
    // the four units of an item on the same XCD (workgroup ids that agree modulo 8: one L2 serves their reads of J)
    const long L = blockIdx.x;
    const long grp = L / 32;
    const int within = (int)(L % 32);
    const long item = grp * 8 + (within & 7);
    const int unit = within >> 3;
    if (item >= (long)nsplit * nprob) return;
    const int p = (int)(item / nsplit), split = (int)(item % nsplit);
    if (st && st[p].stage != want_stage) return;
    const int kbeg = split * rows_per_split, kend = min(m, kbeg + rows_per_split);
    const double *Jp = J + (size_t)p * m * n;
    double *Gp = Gpart + ((size_t)p * nsplit + split) * (size_t)n * n;
    double *gout = gpart + ((size_t)p * nsplit + split) * n;
    const int wv = threadIdx.x >> 6;
    if (unit < 2) {
        double *tA = gsm;                                   // 2 x 256 * GRAM_LD2 (two tile buffers)
        double *fs = tA + 2 * 256 * GRAM_LD2;               // 2 x GRAM_KT2
        double *gscr = fs + 2 * GRAM_KT2;                   // 1024
        const int c0 = unit * 256, nc = min(256, n - c0);
        const double *fp = nullptr;                         // (g = J^T f is formed by the square units: gram_sq_wave_db)
        const double *Jc = Jp + (size_t)c0 * m;
        double *Gc = Gp + (size_t)c0 * n + c0;
        switch (wv) {
        case 0: gram_tri_wave_db<0>(tA, fs, kbeg, kend, m, nc, Jc, fp, Gc, gout + c0, gscr, n); break;
        case 1: gram_tri_wave_db<1>(tA, fs, kbeg, kend, m, nc, Jc, fp, Gc, gout + c0, gscr, n); break;
        case 2: gram_tri_wave_db<2>(tA, fs, kbeg, kend, m, nc, Jc, fp, Gc, gout + c0, gscr, n); break;
        case 3: gram_tri_wave_db<3>(tA, fs, kbeg, kend, m, nc, Jc, fp, Gc, gout + c0, gscr, n); break;
        case 4: gram_tri_wave_db<4>(tA, fs, kbeg, kend, m, nc, Jc, fp, Gc, gout + c0, gscr, n); break;
        case 5: gram_tri_wave_db<5>(tA, fs, kbeg, kend, m, nc, Jc, fp, Gc, gout + c0, gscr, n); break;
        case 6: gram_tri_wave_db<6>(tA, fs, kbeg, kend, m, nc, Jc, fp, Gc, gout + c0, gscr, n); break;
        default: gram_tri_wave_db<7>(tA, fs, kbeg, kend, m, nc, Jc, fp, Gc, gout + c0, gscr, n); break;
        }
    } else {
        double *tB = gsm;                                   // 2 x 384 * GRAM_LD2: panel 0 + this unit's half of panel 1, two buffers
        const int r0 = 256 + (unit - 2) * 128, nA = max(0, min(128, n - r0));
        if (nA == 0) return;                                // (uniform)
        const double *JA = Jp + (size_t)r0 * m;
        double *Gs = Gp + (size_t)r0 * n;                   // G(r0 + gr, gcol) of the row-major slab
        const double *fq = f ? f + (size_t)p * m : nullptr;
        double *gB = unit == 2 ? gout : nullptr, *gA = gout + r0;    // unit 2: columns 0..255 and 256..383; unit 3: 384..
        switch (wv) {
        case 0: gram_sq_wave_db<0>(tB, kbeg, kend, m, nA, Jp, JA, Gs, n, fq, gB, gA); break;
        case 1: gram_sq_wave_db<1>(tB, kbeg, kend, m, nA, Jp, JA, Gs, n, fq, gB, gA); break;
        case 2: gram_sq_wave_db<2>(tB, kbeg, kend, m, nA, Jp, JA, Gs, n, fq, gB, gA); break;
        case 3: gram_sq_wave_db<3>(tB, kbeg, kend, m, nA, Jp, JA, Gs, n, fq, gB, gA); break;
        case 4: gram_sq_wave_db<4>(tB, kbeg, kend, m, nA, Jp, JA, Gs, n, fq, gB, gA); break;
        case 5: gram_sq_wave_db<5>(tB, kbeg, kend, m, nA, Jp, JA, Gs, n, fq, gB, gA); break;
        case 6: gram_sq_wave_db<6>(tB, kbeg, kend, m, nA, Jp, JA, Gs, n, fq, gB, gA); break;
        default: gram_sq_wave_db<7>(tB, kbeg, kend, m, nA, Jp, JA, Gs, n, fq, gB, gA); break;
        }
    }
}

// Sum the K-split partials in split order and mirror the lower triangle.  A thread per entry of the LOWER triangle (a
// wave: 64 consecutive rows of one column, contiguous in every slab); it writes G(r, c) and G(c, r).  The partials of
// eight splits are requested together and added in split order (a thread per entry of the full matrix with one load in
// flight read every slab entry twice, one dependent load after the other: 90 us for 64 slabs of 512 x 512).
static __global__ void __launch_bounds__(256)
k_gram_reduce(int n, int nsplit, const double *__restrict__ Gpart, double *__restrict__ G,
              const double *__restrict__ gpart, double *__restrict__ g,
              const LmState *__restrict__ st, int want_stage)
{
    const int p = blockIdx.y;
    if (st && st[p].stage != want_stage) return;
    const size_t nn = (size_t)n * n;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g && e < (size_t)n) {
        double sg = 0.0;
        for (int k = 0; k < nsplit; ++k) sg = sg + gpart[((size_t)p * nsplit + k) * n + e];
        g[(size_t)p * n + e] = sg;
    }
    if (e >= nn) return;
    // The partial slabs are ROW-major: entry (row r, column c), r >= c, at [r * n + c].  The MFMA result map puts a tile's
    // sixteen COLUMNS in the lanes' low bits, so a producer's store instruction writes four runs of 128 bytes (column-major
    // slabs made it 64 scattered 8-byte writes: eight times the write requests, ~30 of the 362 us k_gram_512 took for the
    // 64 slabs of one 65536 x 512 problem).  A wave here: 64 consecutive columns of one row, contiguous in every slab.
    const int c = (int)(e % n), r = (int)(e / n);
    if (r < c) return;                             // the slabs hold every entry with row >= column (16 x 16 tile granularity)
    const double *gp = Gpart + (size_t)p * nsplit * nn + e;
    double s = 0.0;
    int k = 0;
    for (; k + 8 <= nsplit; k += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = gp[(size_t)(k + u) * nn];
#pragma unroll
        for (int u = 0; u < 8; ++u) s = s + v[u];
    }
    for (; k < nsplit; ++k) s = s + gp[(size_t)k * nn];
    G[(size_t)p * nn + e] = s;                                  // G(c, r) of the column-major result: contiguous ...
    if (r != c) G[(size_t)p * nn + (size_t)c * n + r] = s;      // ... and its mirror G(r, c)
}

// g = J^T f: one wave per column, lanes stride the rows (coalesced), shuffle reduction.
static __global__ void __launch_bounds__(256)
k_jtf(int m, int n, const double *__restrict__ J, const double *__restrict__ f,
      double *__restrict__ g, const LmState *__restrict__ st, int want_stage)
{
    const int p = blockIdx.y;
    if (st && st[p].stage != want_stage) return;
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (j >= n) return;
    const double *col = J + (size_t)p * m * n + (size_t)j * m;
    const double *fp = f + (size_t)p * m;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int i = lane;
    for (; i + 192 < m; i += 256) {
        s0 = s0 + col[i] * fp[i];
        s1 = s1 + col[i + 64] * fp[i + 64];
        s2 = s2 + col[i + 128] * fp[i + 128];
        s3 = s3 + col[i + 192] * fp[i + 192];
    }
    for (; i < m; i += 64) s0 = s0 + col[i] * fp[i];
    double s = wave_reduce_sum((s0 + s1) + (s2 + s3));
    if (lane == 0) g[(size_t)p * n + j] = s;
}

// grad(i) = dot(jac(:,i), fvec) in the reference's order (src/nonlin_solve.f90:565-567): one thread per column,
// rows ascending, separate multiply and add.  The Newton line search feeds dot(grad, dir) into the backtracking
// formula, so a reordered sum changes the accepted step in its last bits.
static __global__ void __launch_bounds__(256)
k_jtf_exact(int m, int n, const double *__restrict__ Jall, const double *__restrict__ fall, double *__restrict__ gall,
            const LmState *__restrict__ st, int want)
{
    const int p = blockIdx.y;                                     // problems of a batch: J [p][n][m], f [p][m], g [p][n]
    if (st && st[p].stage != want) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const double *J = Jall + (size_t)p * m * n, *f = fall + (size_t)p * m;
    double *g = gall + (size_t)p * n;
    const double *col = J + (size_t)j * m;
    double s = 0.0;
    int i = 0;
    for (; i + 8 <= m; i += 8) {
        double c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = col[i + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) s = s + c[u] * f[i + u];
    }
    for (; i < m; ++i) s = s + col[i] * f[i];
    g[j] = s;
}
