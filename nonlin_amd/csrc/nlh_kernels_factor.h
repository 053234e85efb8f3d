// nlh_kernels_factor.h -- the two factorisations behind lmfactor's interface and the
// outer-loop head of lss_solve that consumes them.
//
//   k_chol_factor : pivoted Cholesky of G = J^T J (normal-equations path).  With
//                   MINPACK's pivot rule ("largest remaining column norm" = largest
//                   remaining Schur diagonal) P^T G P = R^T R gives lmfactor's R up to
//                   row signs; qtf = R^-T P^T g is carried as an extra column.
//   k_qr_factor   : lmfactor itself (pivoted Householder QR, src/nonlin_least_squares.f90:
//                   569-667) plus the Q^T f sweep (:241-253), one workgroup per problem.
//   lm_head       : :229-238 (first-iteration scaling), :256-267 (scaled gradient norm),
//                   :270-278 (gradient convergence, rescale).
#pragma once
#include "nlh_common.h"
#include "nlh_lm_head.h"

// ---------------------------------------------------------------------------
// Pivoted Cholesky, one workgroup per problem, G (n-by-n, column-major, symmetric, both
// triangles valid on entry) in global memory/L2; the upper triangle is overwritten by R.
//
// Left-looking: step j forms row j of R from the untouched entries G(j,k) and the rows above,
//   R(j,k) = (G(j,k) - sum_{i<j} R(i,j) R(i,k)) / R(j,j),   one wave per column k,
// so a step has no trailing-matrix update and only two dependent L2 round trips (the column
// interchange and the dot products).  Everything a step decides on lives in LDS: the Schur
// diagonal d (pivot search = "largest remaining column norm", lmfactor :622-625), column j
// of R (broadcast operand of the dots) and the permuted gradient (qtf is carried as an extra
// column: qtf(j) = (g_j - sum_i R(i,j) qtf(i)) / R(j,j)).
// Dynamic LDS: (3n + 64) doubles.
// standalone != 0: only factor (stage-level entry point nlh_chol_factor).
// ---------------------------------------------------------------------------
static __global__ void __launch_bounds__(1024)
k_chol_factor(int n, double *__restrict__ Gall, const double *__restrict__ Gsrc_all,
              const double *__restrict__ gall,
              LmVecs v, const double *__restrict__ xall, LmState *__restrict__ st,
              int32_t *__restrict__ info, double factor, double gtol, double pivot_tol,
              int standalone, int want_stage)
{
    extern __shared__ double smem[];
    const int p = blockIdx.x;
    LmState *s = st ? st + p : nullptr;
    if (s && s->stage != want_stage) return;
    const int tid = threadIdx.x, BS = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = BS >> 6;
    double *colj = smem;            // n   : R(0:j, j)
    double *gp = smem + n;          // n   : permuted gradient -> qtf
    double *d = smem + 2 * n;       // n   : Schur diagonal
    double *red = smem + 3 * n;     // 64
    int *redi = reinterpret_cast<int *>(red + 32);
    double *G = Gall + (size_t)p * n * n;
    int32_t *ipvt = v.ipvt + (size_t)p * n;
    double *acnorm = v.acnorm + (size_t)p * n;
    double *qtf = v.qtf + (size_t)p * n;

    if (Gsrc_all) {                             // factor a copy: the Gram matrix itself is preserved
        const double *Gs = Gsrc_all + (size_t)p * n * n;
        for (int e = tid; e < n * n; e += BS) G[e] = Gs[e];
        __syncthreads();
    }
    for (int k = tid; k < n; k += BS) {
        const double dk = G[(size_t)k * n + k];
        d[k] = dk;
        acnorm[k] = sqrt(fmax(dk, 0.0));        // ||J(:,k)||, lmfactor :611-616
        ipvt[k] = k;
        gp[k] = gall[(size_t)p * n + k];
    }
    __syncthreads();

    int bad = 0;
    for (int j = 0; j < n; ++j) {
        // pivot: first maximum of the remaining Schur diagonal (:622-625)
        double bv = 0.0;
        int bk = 0x7fffffff;
        for (int k = j + tid; k < n; k += BS) {
            const double dk = d[k];
            if (bk == 0x7fffffff || dk > bv) { bv = dk; bk = k; }
        }
        const int q = block_argmax_first(bv, bk, red, redi);
        if (q != j) {                           // symmetric interchange j <-> q (upper storage)
            for (int r = tid; r < j; r += BS) {                 // rows of R above
                double t = G[(size_t)j * n + r];
                G[(size_t)j * n + r] = G[(size_t)q * n + r];
                G[(size_t)q * n + r] = t;
            }
            for (int r = j + 1 + tid; r < q; r += BS) {         // untouched entries between j and q
                double t = G[(size_t)r * n + j];
                G[(size_t)r * n + j] = G[(size_t)q * n + r];
                G[(size_t)q * n + r] = t;
            }
            for (int c = q + 1 + tid; c < n; c += BS) {         // untouched entries right of q
                double t = G[(size_t)c * n + j];
                G[(size_t)c * n + j] = G[(size_t)c * n + q];
                G[(size_t)c * n + q] = t;
            }
            if (tid == 0) {
                double t = d[j]; d[j] = d[q]; d[q] = t;
                int32_t ti = ipvt[j]; ipvt[j] = ipvt[q]; ipvt[q] = ti;
                double tg = gp[j]; gp[j] = gp[q]; gp[q] = tg;
            }
            __syncthreads();
        }
        const double dj = d[j];
        const double an = acnorm[ipvt[j]];
        if (!(dj > pivot_tol * an * an) || !(dj > 0.0)) { bad = j + 1; break; }   // uniform
        const double rjj = sqrt(dj);
        for (int i = tid; i < j; i += BS) colj[i] = G[(size_t)j * n + i];         // R(0:j, j)
        __syncthreads();
        // row j of R: a wave takes CW columns k > j at a time (independent loads in flight before
        // the shuffle reductions); column index n is the gradient, i.e. qtf
        constexpr int CW = 8;
        for (int k0 = j + 1 + wid * CW; k0 <= n; k0 += nw * CW) {
            double sm[CW], g0[CW];
#pragma unroll
            for (int u = 0; u < CW; ++u) {
                sm[u] = 0.0;
                g0[u] = (k0 + u < n) ? G[(size_t)(k0 + u) * n + j] : 0.0;   // untouched G(j,k), loaded up front
            }
            for (int i = lane; i < j; i += 64) {
                const double cj = colj[i];
                double v[CW];
#pragma unroll
                for (int u = 0; u < CW; ++u) {
                    const int k = k0 + u;
                    v[u] = (k < n) ? G[(size_t)k * n + i] : ((k == n) ? gp[i] : 0.0);
                }
#pragma unroll
                for (int u = 0; u < CW; ++u) sm[u] = sm[u] + cj * v[u];
            }
#pragma unroll
            for (int u = 0; u < CW; ++u) sm[u] = wave_reduce_sum(sm[u]);
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < CW; ++u) {
                    const int k = k0 + u;
                    if (k < n) {
                        const double r = (g0[u] - sm[u]) / rjj;
                        G[(size_t)k * n + j] = r;
                        d[k] = d[k] - r * r;
                    } else if (k == n) {
                        gp[j] = (gp[j] - sm[u]) / rjj;          // qtf(j)
                    }
                }
            }
        }
        if (tid == 0) G[(size_t)j * n + j] = rjj;
        __syncthreads();
    }
    if (bad) {
        if (tid == 0) {
            if (info) info[p] = bad;
            if (s) { s->stage = ST_NEED_QR; }
        }
        return;
    }
    for (int k = tid; k < n; k += BS) qtf[k] = gp[k];
    if (tid == 0 && info) info[p] = 0;
    __syncthreads();
    if (standalone || !s) return;
    if (tid == 0) { s->factor_kind = 0; s->pivoted = 1; }
    if (s->head_done) {                         // re-factorisation inside an outer iteration: head already ran
        __syncthreads();
        if (tid == 0) s->stage = ST_NE_READY;
        return;
    }
    lm_head<false>(n, G, n, ipvt, acnorm, qtf, xall + (size_t)p * n, v.diag + (size_t)p * n,
                   v.diag_prev + (size_t)p * n, s, factor, gtol, ST_NE_READY, red, nullptr);
}

// ---------------------------------------------------------------------------
// Blocked Cholesky in natural column order (no pivoting): the fast path.
// When the Gauss-Newton step is accepted (the common case) the step, the scaled gradient norm
// and the predicted reduction do not depend on lmfactor's pivot order, only on J^T J itself; the
// pivoted kernel above is run afterwards only for problems that need lmfactor's R (lmpar
// iteration, weak pivots).  Right-looking with NB-wide panels: the diagonal block is factored
// in LDS, the block row is a per-column triangular solve (thread per column), and the trailing
// update is one pass with the panel broadcast from LDS -- n/NB dependent global round trips
// instead of n.  The gradient rides along as column n (qtf = R^-T g).
// Dynamic LDS: NB*n + NB*NB + n + NB + 64 doubles.
// ---------------------------------------------------------------------------
template <int NB>
__global__ void __launch_bounds__(1024)
k_chol_nopiv(int n, const double *__restrict__ Gall, const double *__restrict__ gall,
             double *__restrict__ Rall, LmVecs v, const double *__restrict__ xall,
             LmState *__restrict__ st, double factor, double gtol, double pivot_tol)
{
    extern __shared__ double smem[];
    const int p = blockIdx.x;
    LmState *s = st + p;
    if (s->stage != ST_HAVE_JAC) return;
    const int tid = threadIdx.x, BS = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = BS >> 6;
    double *panel = smem;                   // NB x n  : rows of R of the current block, panel[i*n + k]
    double *r11 = panel + (size_t)NB * n;   // NB x NB : diagonal block, r11[i + c*NB] (upper)
    double *gp = r11 + NB * NB;             // n       : gradient -> qtf
    double *yb = gp + n;                    // NB      : qtf entries of the current block
    double *red = yb + NB;                  // 64
    __shared__ int bad_sh;
    __shared__ double dinv[NB];             // 1 / diag of the current diagonal block
    const double *G = Gall + (size_t)p * n * n;
    double *R = Rall + (size_t)p * n * n;
    int32_t *ipvt = v.ipvt + (size_t)p * n;
    double *acnorm = v.acnorm + (size_t)p * n;
    double *qtf = v.qtf + (size_t)p * n;

    for (int e = tid; e < n * n; e += BS) R[e] = G[e];
    for (int k = tid; k < n; k += BS) {
        acnorm[k] = sqrt(fmax(G[(size_t)k * n + k], 0.0));
        ipvt[k] = k;
        gp[k] = gall[(size_t)p * n + k];
    }
    if (tid == 0) bad_sh = 0;
    __syncthreads();

#ifdef NLH_DEBUG_TIMING
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tq = wall_clock64();
#define TPH(i) { __syncthreads(); unsigned long long t_ = wall_clock64(); tph[i] += t_ - tq; tq = t_; }
#else
#define TPH(i)
#endif
    for (int jb = 0; jb < n; jb += NB) {
        const int nbk = min(NB, n - jb);
        TPH(4)
        // diagonal block -> LDS
        for (int e = tid; e < NB * NB; e += BS) {
            const int i = e % NB, c = e / NB;
            r11[e] = (i <= c && c < nbk) ? R[(size_t)(jb + c) * n + jb + i] : 0.0;
        }
        __syncthreads();
        TPH(5)
        // factor it inside wave 0: lane c keeps column c of the block in registers and row j is
        // broadcast with v_readlane (compile-time lane numbers), so the 16 steps need neither LDS
        // round trips nor barriers.
        if (wid == 0) {
            double col[NB];                                     // lane c: col[i] = A(i, c), i <= c
#pragma unroll
            for (int i = 0; i < NB; ++i) col[i] = (lane < nbk && i <= lane) ? r11[i + lane * NB] : 0.0;
            const double an_l = (lane < nbk) ? acnorm[jb + lane] : 0.0;
            const double thr_l = pivot_tol * an_l * an_l;       // weak-pivot threshold of column `lane`
            double myinv = 0.0;
            int bad = 0;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j < nbk) {                                  // uniform
                    const double dj = readlane_f64(col[j], j);
                    const double tj = readlane_f64(thr_l, j);
                    if ((!(dj > tj) || !(dj > 0.0)) && bad == 0) bad = jb + j + 1;
                    double rjj, rinv;
                    sqrt_rsqrt(fmax(dj, 1e-300), rjj, rinv);
                    const double rjc = (lane > j) ? col[j] * rinv : 0.0;    // row j, entry `lane`
                    if (lane == j) { col[j] = rjj; myinv = rinv; }
                    else if (lane > j) col[j] = rjc;
                    // entries i > lane of col are never read, so the update needs no predicate
#pragma unroll
                    for (int i = j + 1; i < NB; ++i) col[i] = col[i] - readlane_f64(rjc, i) * rjc;
                }
            }
            if (bad && lane == 0) bad_sh = bad;
#pragma unroll
            for (int i = 0; i < NB; ++i)
                if (lane < nbk && i <= lane) r11[i + lane * NB] = col[i];
            if (lane < NB) dinv[lane] = myinv;                  // reciprocal diagonal for the block row
        }
        TPH(6)
        __syncthreads();
        if (bad_sh) break;                                      // uniform
        TPH(0)
        // write the factored diagonal block back
        for (int e = tid; e < NB * NB; e += BS) {
            const int i = e % NB, c = e / NB;
            if (i <= c && c < nbk) R[(size_t)(jb + c) * n + jb + i] = r11[e];
        }
        // block row: R12 = R11^-T A12, one thread per column k >= jb + nbk; k == n is the gradient
        for (int k = jb + nbk + tid; k <= n; k += BS) {
            double a[NB];
            if (k < n) {
#pragma unroll
                for (int i = 0; i < NB; ++i) a[i] = (i < nbk) ? R[(size_t)k * n + jb + i] : 0.0;
            } else {
#pragma unroll
                for (int i = 0; i < NB; ++i) a[i] = (i < nbk) ? gp[jb + i] : 0.0;
            }
#pragma unroll
            for (int l = 0; l < NB; ++l) {                      // forward substitution, axpy form
                if (l < nbk) {
                    const double al = a[l] * dinv[l];
                    a[l] = al;
#pragma unroll
                    for (int i = l + 1; i < NB; ++i) a[i] = a[i] - r11[l + i * NB] * al;
                }
            }
            if (k < n) {
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    if (i < nbk) { R[(size_t)k * n + jb + i] = a[i]; panel[(size_t)i * n + k] = a[i]; }
            } else {
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    if (i < nbk) { gp[jb + i] = a[i]; yb[i] = a[i]; }
            }
        }
        __syncthreads();
        TPH(1)
        // trailing update: A22(r,c) -= sum_i R12(i,r) R12(i,c), jb+nbk <= r <= c; gradient likewise.
        // A wave walks its columns in 64-row pieces and handles TU pieces at a time: all global loads
        // of a group are issued before any store, so the read-modify-write round trips overlap.
        const int t0 = jb + nbk;
        {
            // 16x16 tiles of the upper triangle with v_mfma_f64_16x16x4_f64: T = C - P_c^T P_r computed as
            // D[c][r] so that lanes & 15 run along r (contiguous in the column-major R).  Operand maps (f64):
            // A[row = l & 15][k = l >> 4], B[k = l >> 4][col = l & 15], D row = (l >> 4) + 4*reg, col = l & 15.
            typedef double v4d_t __attribute__((ext_vector_type(4)));
            const int nt = (n - t0 + 15) / 16;                     // tiles per side
            const int ntile = nt * (nt + 1) / 2;
            for (int tix = wid; tix < ntile; tix += nw) {
                int tc = (int)((sqrtf(8.0f * (float)tix + 1.0f) - 1.0f) * 0.5f);   // tile column block, row block tr <= tc
                while ((tc + 1) * (tc + 2) / 2 <= tix) ++tc;
                while (tc * (tc + 1) / 2 > tix) --tc;
                const int tr = tix - tc * (tc + 1) / 2;
                const int cb = t0 + tc * 16, rb = t0 + tr * 16;
                const int crow = cb + (lane >> 4), rcol = rb + (lane & 15);
                v4d_t acc;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = crow + 4 * q;
                    acc[q] = (c < n && rcol <= c) ? R[(size_t)c * n + rcol] : 0.0;
                }
                const int ca = cb + (lane & 15);
#pragma unroll
                for (int kk = 0; kk < NB / 4; ++kk) {
                    const int k = kk * 4 + (lane >> 4);
                    const double av = (k < nbk && ca < n) ? -panel[(size_t)k * n + ca] : 0.0;      // A[c][k] = -P(k, c)
                    const double bv = (k < nbk && rcol < n) ? panel[(size_t)k * n + rcol] : 0.0;   // B[k][r] =  P(k, r)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = crow + 4 * q;
                    if (c < n && rcol <= c) R[(size_t)c * n + rcol] = acc[q];
                }
            }
            if (wid == nw - 1) {                            // gradient "column"
                for (int r = t0 + lane; r < n; r += 64) {
                    double acc = gp[r];
#pragma unroll
                    for (int i = 0; i < NB; ++i)
                        if (i < nbk) acc = acc - panel[(size_t)i * n + r] * yb[i];
                    gp[r] = acc;
                }
            }
        }
        __syncthreads();
        TPH(2)
    }
#ifdef NLH_DEBUG_TIMING
    if (tid == 0 && p == 0) printf("[chol_nopiv] diag %.1f us, blockrow %.1f us, trailing %.1f us, other %.1f us\n",
                                   tph[0] * 1e-2, tph[1] * 1e-2, tph[2] * 1e-2, tph[4] * 1e-2);
    if (tid == 0 && p == 0) printf("[chol_nopiv]   diag: load %.1f us, factor %.1f us\n", tph[5] * 1e-2, tph[6] * 1e-2);
#endif
    if (bad_sh) {
        if (tid == 0) s->stage = ST_NEED_PCHOL;                 // let the pivoted kernel decide (it may ask for QR)
        return;
    }
    for (int k = tid; k < n; k += BS) qtf[k] = gp[k];
    __syncthreads();
    if (tid == 0) { s->factor_kind = 0; s->pivoted = 0; }
    lm_head<false>(n, R, n, ipvt, acnorm, qtf, xall + (size_t)p * n, v.diag + (size_t)p * n,
                   v.diag_prev + (size_t)p * n, s, factor, gtol, ST_NE_READY, red, nullptr);
}

// ---------------------------------------------------------------------------
// The same blocked Cholesky for a HANDFUL of problems (BASELINE config 5: one 65536 x 512 problem): one launch per
// panel step over CHOLMC_NWG workgroups per problem instead of one workgroup for the whole factorisation -- on one CU
// the trailing update alone took 608 of the 985 us of an n = 512 factorisation (every 16 x 16 tile a read-modify-write
// through one CU's path to L2).  No workgroup waits for another inside a launch: every workgroup factors the diagonal
// block and solves the block row REDUNDANTLY (the same instructions on the same operands: the same bits), keeps the
// panel in LDS and then takes its share of the trailing tiles; the solved panel goes to a side buffer and is copied into
// R by the next launch (the other workgroups of this launch still read the unsolved entries).  The gradient column has
// a workgroup of its own.  Same operations in the same order as k_chol_nopiv: bitwise the same R, qtf and flags.
// Side buffer: 2 x NB x n doubles per problem (alternating between steps), bad: one int per problem.
// Dynamic LDS of the step kernel: NB*n + NB*NB + 2*NB doubles.
// ---------------------------------------------------------------------------
#define CHOLMC_NWG 32
// The diagonal block inside one wave: lane c keeps column c in registers, row j is broadcast with v_readlane (k_chol_nopiv's
// code, shared by the begin kernel -- first block -- and the look-ahead of the step kernel).  r11: NB x NB in LDS, upper,
// r11[i + c*NB]; returns the 1-based index of a bad pivot or 0, leaves the factored block in r11 and 1/diag in dinv.
template <int NB>
__device__ __forceinline__ int chol_diag_wave(double *r11, double *dinv, double an_l /* acnorm[jb + lane], 0 for lane >= nbk */, int jb, int nbk,
                                              double pivot_tol, int lane)
{
    double col[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) col[i] = (lane < nbk && i <= lane) ? r11[i + lane * NB] : 0.0;
    const double thr_l = pivot_tol * an_l * an_l;
    double myinv = 0.0;
    int bd = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        if (j < nbk) {
            const double dj = readlane_f64(col[j], j);
            const double tj = readlane_f64(thr_l, j);
            if ((!(dj > tj) || !(dj > 0.0)) && bd == 0) bd = jb + j + 1;
            double rjj, rinv;
            sqrt_rsqrt(fmax(dj, 1e-300), rjj, rinv);
            const double rjc = (lane > j) ? col[j] * rinv : 0.0;
            if (lane == j) { col[j] = rjj; myinv = rinv; }
            else if (lane > j) col[j] = rjc;
#pragma unroll
            for (int i = j + 1; i < NB; ++i) col[i] = col[i] - readlane_f64(rjc, i) * rjc;
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
        if (lane < nbk && i <= lane) r11[i + lane * NB] = col[i];
    if (lane < NB) dinv[lane] = myinv;
    return bd;
}

// The substitution step l of the transposed block row (see the look-ahead workgroup of k_chol_mc_step); the DPP control
// word must be a literal, hence the recursion over L.
template <int NB, int L>
__device__ __forceinline__ void chol_tsub(double &a, const double (&rr)[NB], double dvi, int ci, int nbk)
{
    if constexpr (L < NB) {
        if (L < nbk) {
            const double t = a * dvi;                                // lane L of the row: the solved entry a(L) * (1 / r11(L, L))
            const int tlo = __double2loint(t), thi = __double2hiint(t);
            const double al = __hiloint2double(__builtin_amdgcn_update_dpp(thi, thi, 0x150 + L, 0xf, 0xf, false),
                                               __builtin_amdgcn_update_dpp(tlo, tlo, 0x150 + L, 0xf, 0xf, false));   // row_newbcast:L
            if (ci == L) a = al;
            else if (ci > L) a = a - rr[L] * al;
        }
        chol_tsub<NB, L + 1>(a, rr, dvi, ci, nbk);
    }
}

// fact: per problem NB*NB + NB doubles -- the factored diagonal block of the NEXT step and its reciprocal diagonal
template <int NB>
__global__ void __launch_bounds__(256)
k_chol_mc_begin(int n, const double *__restrict__ Gall, const double *__restrict__ gall, double *__restrict__ Rall, LmVecs v,
                double *__restrict__ fact, int32_t *__restrict__ bad, const LmState *__restrict__ st, double pivot_tol)
{
    __shared__ double r11[NB * NB];
    __shared__ double dinv[NB];
    const int p = blockIdx.y;
    if (st[p].stage != ST_HAVE_JAC) return;
    const double *G = Gall + (size_t)p * n * n;
    double *R = Rall + (size_t)p * n * n;
    const size_t nn = (size_t)n * n;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nn; e += (size_t)gridDim.x * blockDim.x) R[e] = G[e];
    if (blockIdx.x != 0) return;
    const int tid = threadIdx.x;
    for (int k = tid; k < n; k += blockDim.x) {
        v.acnorm[(size_t)p * n + k] = sqrt(fmax(G[(size_t)k * n + k], 0.0));
        v.ipvt[(size_t)p * n + k] = k;
        v.qtf[(size_t)p * n + k] = gall[(size_t)p * n + k];           // the gradient, transformed in place into qtf
    }
    const int nbk = min(NB, n);
    for (int e = tid; e < NB * NB; e += blockDim.x) {
        const int i = e % NB, c = e / NB;
        r11[e] = (i <= c && c < nbk) ? G[(size_t)c * n + i] : 0.0;
    }
    __syncthreads();                                                // (also: acnorm of this problem is written)
    if (tid < 64) {
        const int bd = chol_diag_wave<NB>(r11, dinv, tid < nbk ? v.acnorm[(size_t)p * n + tid] : 0.0, 0, nbk, pivot_tol, tid);
        if (tid == 0) bad[p] = bd;
    }
    __syncthreads();
    double *fp = fact + (size_t)p * 2 * (NB * NB + NB);              // step 0 reads copy 0
    for (int e = tid; e < NB * NB + NB; e += blockDim.x) fp[e] = e < NB * NB ? r11[e] : dinv[e - NB * NB];
}

template <int NB>
__global__ void __launch_bounds__(512)
k_chol_mc_step(int n, int jb, double *__restrict__ Rall, LmVecs v, double *__restrict__ side, double *__restrict__ fact,
               int32_t *__restrict__ bad, const LmState *__restrict__ st, double pivot_tol)
{
    extern __shared__ double smem[];
    const int p = blockIdx.y;
    if (st[p].stage != ST_HAVE_JAC || bad[p]) return;
    // workgroups 0 .. NWG-1: the trailing tiles; NWG: the gradient column; NWG + 1: the LOOK-AHEAD -- tile (0, 0) of the
    // trailing matrix is the next step's diagonal block: a workgroup of its own solves the sixteen columns of the block row
    // that tile needs, updates the tile and factors it (one wave, ~4 us of dependent steps) while the others do the rest,
    // and leaves the factored block in `fact` for the next launch
    const int wg = blockIdx.x, NWG = gridDim.x - 2;
    const bool gradwg = (wg == NWG), la = (wg == NWG + 1);
    const int tid = threadIdx.x, BS = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = BS >> 6;
    double *panel = smem;                   // NB x n : rows of R of the current block, panel[i*n + k]
    double *r11 = panel + (size_t)NB * n;   // NB x NB: the factored diagonal block, r11[i + c*NB] (upper)
    double *dinv = r11 + NB * NB;           // NB
    double *yb = dinv + NB;                 // NB
    double *r11n = yb + NB;                 // NB x NB + NB: the next diagonal block (look-ahead, workgroup 0)
    double *R = Rall + (size_t)p * n * n;
    double *gp = v.qtf + (size_t)p * n;
    const int step = jb / NB;
    // (two copies, by step parity: the look-ahead of this launch writes the next step's while late workgroups of this launch
    // may not have read this step's yet)
    const double *fp = fact + ((size_t)p * 2 + (step & 1)) * (NB * NB + NB);
    double *fpn = fact + ((size_t)p * 2 + ((step & 1) ^ 1)) * (NB * NB + NB);
    double *cur = side + ((size_t)p * 2 + (step & 1)) * NB * n;
    const double *prev = side + ((size_t)p * 2 + ((step & 1) ^ 1)) * NB * n;
    const int nbk = min(NB, n - jb), t0 = jb + nbk;
    typedef double v4d_t __attribute__((ext_vector_type(4)));
#ifdef NLH_DEBUG_TIMING
    long long ck[6]; ck[0] = wall_clock64();
#define CKM(i) ck[i] = wall_clock64();
#else
#define CKM(i)
#endif
    // this wave's trailing tiles (at most CHOLMC_TPW of them): their entries are requested first -- they do not depend on
    // the panel -- and wait in registers
    constexpr int TPW = 2;
    const int nt = (n - t0 + 15) / 16, ntile = nt * (nt + 1) / 2;
    v4d_t acc[TPW];
    int tcs[TPW], trs[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const int tix = gradwg ? ntile : la ? ((wid == 0 && u == 0) ? 0 : ntile) : 1 + wg * nw + wid + u * NWG * nw;
        int tc = 0, tr = 0;
        if (tix < ntile) {
            tc = (int)((sqrtf(8.0f * (float)tix + 1.0f) - 1.0f) * 0.5f);
            while ((tc + 1) * (tc + 2) / 2 <= tix) ++tc;
            while (tc * (tc + 1) / 2 > tix) --tc;
            tr = tix - tc * (tc + 1) / 2;
        } else tc = -1;
        tcs[u] = tc; trs[u] = tr;
        const int crow = t0 + tc * 16 + (lane >> 4), rcol = t0 + tr * 16 + (lane & 15);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = crow + 4 * q;
            acc[u][q] = (tc >= 0 && c < n && rcol <= c) ? R[(size_t)c * n + rcol] : 0.0;
        }
    }
    // the factored diagonal block (from the launch before: its look-ahead, or the begin kernel)
    for (int e = tid; e < NB * NB + NB; e += BS) r11[e] = fp[e];    // (dinv follows r11 in both)
    const double an_next = (la && t0 + lane < n && lane < NB) ? v.acnorm[(size_t)p * n + t0 + lane] : 0.0;   // for the look-ahead's pivot test
    // the previous step's solved panel (diagonal block and block row) goes to R in this launch, every workgroup its share
    // of the columns: requested here, stored at the very end (nothing in this launch reads those rows of R)
    const int pj = jb - NB;
    const int kcb = pj + tid;                                       // (n - pj <= BS + ...: one column per thread and trip, see the tail)
    const bool cbk = step > 0 && !gradwg && !la && kcb < n && (((kcb - pj) / 16) % NWG == wg);
    double cbv[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) cbv[i] = cbk ? prev[(size_t)i * n + kcb] : 0.0;
    // ... and this thread's column of the block row (first trip), which does not depend on the factored block either: what
    // the previous launch wrote on other XCDs comes from memory, 2-3 us away -- one such wait per step, not two
    const int kend = la ? min(n, t0 + 16) : n + (gradwg ? 1 : 0);
    const int k1 = t0 + tid;
    double a1[NB];
    {
        const bool vec = (nbk == NB) && ((n & 1) == 0);
        if (!la && k1 < kend && k1 < n && vec) {
            const double2 *src = reinterpret_cast<const double2 *>(R + (size_t)k1 * n + jb);   // (n and jb even: 16-byte aligned)
#pragma unroll
            for (int i = 0; i < NB / 2; ++i) { const double2 t2 = src[i]; a1[2 * i] = t2.x; a1[2 * i + 1] = t2.y; }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i)
                a1[i] = (!la && k1 < kend && i < nbk) ? (k1 < n ? R[(size_t)k1 * n + jb + i] : gp[jb + i]) : 0.0;
        }
    }
    const double a_la = (la && tid < 256 && t0 + (tid >> 4) < n && (tid & 15) < nbk) ? R[(size_t)(t0 + (tid >> 4)) * n + jb + (tid & 15)] : 0.0;
    nlh_lds_barrier();                                              // (LDS traffic only: the loads above stay in flight)
    CKM(1)
    if (wg == 0)                                                    // the factored diagonal block -> side buffer
        for (int e = tid; e < NB * NB; e += BS) {
            const int i = e % NB, c = e / NB;
            if (i <= c && c < nbk) cur[(size_t)i * n + jb + c] = r11[e];
        }
    // block row R12 = R11^-T A12, a thread per column; k == n: the gradient (its own workgroup)
    if (la) {
        // The look-ahead's sixteen columns, TRANSPOSED: sixteen lanes per column (a DPP row), lane (k, i) holds entry i of
        // column k and its own column i of the factored block; step l: the row's lane l has the solved entry, a DPP row
        // broadcast hands it to the lanes below, one multiply and subtract each -- the thread-per-column substitution's
        // operations on the same operands in the same order, 16 short steps instead of 136 dependent round trips (the
        // look-ahead is the critical path of the launch).
        const int ci = tid & 15, kk = t0 + (tid >> 4);
        if (tid < 256) {
            double a = a_la;
            double rr[NB];
#pragma unroll
            for (int l = 0; l < NB; ++l) rr[l] = r11[l + ci * NB];   // r11(l, ci), used for l < ci
            const double dvi = dinv[ci];
            chol_tsub<NB, 0>(a, rr, dvi, ci, nbk);
            if (kk < n && ci < nbk) panel[(size_t)ci * n + kk] = a;
        }
    }
    for (int k = la ? kend : k1; k < kend; k += BS) {
        double a[NB];
        if (k == k1) {
#pragma unroll
            for (int i = 0; i < NB; ++i) a[i] = a1[i];
        } else if (k < n) {
#pragma unroll
            for (int i = 0; i < NB; ++i) a[i] = (i < nbk) ? R[(size_t)k * n + jb + i] : 0.0;
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) a[i] = (i < nbk) ? gp[jb + i] : 0.0;
        }
#pragma unroll
        for (int l = 0; l < NB; ++l) {                      // forward substitution, axpy form
            if (l < nbk) {
                const double al = a[l] * dinv[l];
                a[l] = al;
#pragma unroll
                for (int i = l + 1; i < NB; ++i) a[i] = a[i] - r11[l + i * NB] * al;
            }
        }
        if (k < n) {
            const bool mine = !gradwg && !la && (((k - jb) / 16) % NWG == wg);
#pragma unroll
            for (int i = 0; i < NB; ++i)
                if (i < nbk) { panel[(size_t)i * n + k] = a[i]; if (mine) cur[(size_t)i * n + k] = a[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i)
                if (i < nbk) yb[i] = a[i];
        }
    }
    nlh_lds_barrier();
    CKM(2)
    if (gradwg) {
        for (int i = tid; i < nbk; i += BS) gp[jb + i] = yb[i];
        for (int r = t0 + tid; r < n; r += BS) {
            double g = gp[r];
#pragma unroll
            for (int i = 0; i < NB; ++i)
                if (i < nbk) g = g - panel[(size_t)i * n + r] * yb[i];
            gp[r] = g;
        }
        return;
    }
    // trailing tiles (k_chol_nopiv's tile arithmetic): T = C - P_c^T P_r as D[c][r]
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        if (tcs[u] < 0) continue;                                   // (wave-uniform)
        const int cb = t0 + tcs[u] * 16, rb = t0 + trs[u] * 16;
        const int crow = cb + (lane >> 4), rcol = rb + (lane & 15), ca = cb + (lane & 15);
#pragma unroll
        for (int kk = 0; kk < NB / 4; ++kk) {
            const int k = kk * 4 + (lane >> 4);
            const double av = (k < nbk && ca < n) ? -panel[(size_t)k * n + ca] : 0.0;
            const double bv = (k < nbk && rcol < n) ? panel[(size_t)k * n + rcol] : 0.0;
            acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[u], 0, 0, 0);
        }
        if (!la) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = crow + 4 * q;
                if (c < n && rcol <= c) R[(size_t)c * n + rcol] = acc[u][q];
            }
        }
        if (u == 0) { CKM(3) }
        if (la && wid == 0 && u == 0 && nt > 0) {
            // LOOK-AHEAD: tile (0, 0) is the next step's diagonal block.  This wave has it in registers: through LDS into
            // the lane-per-column form, factored here while the other waves and workgroups finish their tiles, and left in
            // `fact` for the next launch.
            const int njb = t0, nnbk = min(NB, n - njb);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = (lane >> 4) + 4 * q, r = lane & 15;   // D row = column c of the block, D col = row r
                r11n[r + c * NB] = (r <= c && c < nnbk) ? acc[u][q] : 0.0;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // one wave: program order is enough
            const int bd = chol_diag_wave<NB>(r11n, r11n + NB * NB, an_next, njb, nnbk, pivot_tol, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (bd && lane == 0) bad[p] = bd;
            for (int e = lane; e < NB * NB + NB; e += 64) fpn[e] = r11n[e];
            CKM(4)
#ifdef NLH_DEBUG_TIMING
            if (lane == 0 && (jb == 64 || jb == 256)) printf("[chol_mc jb=%d look-ahead wg] loads+sync %lld blockrow %lld tile0 %lld lookahead %lld (x10 ns)\n", jb, ck[1]-ck[0], ck[2]-ck[1], ck[3]-ck[2], ck[4]-ck[3]);
#endif
        }
    }
#ifdef NLH_DEBUG_TIMING
    CKM(5)
    if (tid == 0 && (wg == NWG - 1 || wg == 0) && (jb == 64 || jb == 256)) printf("[chol_mc jb=%d wg%d] loads+sync %lld blockrow %lld tiles %lld (x10 ns)\n", jb, wg, ck[1]-ck[0], ck[2]-ck[1], ck[5]-ck[2]);
#endif
    if (cbk) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (kcb >= pj + i) R[(size_t)kcb * n + pj + i] = cbv[i];
    }
    if (step > 0 && !gradwg && !la)                                 // (n - pj > 512: the columns beyond the first trip)
        for (int k = kcb + BS; k < n; k += BS) {
            if (((k - pj) / 16) % NWG != wg) continue;
#pragma unroll
            for (int i = 0; i < NB; ++i) R[(size_t)k * n + pj + i] = prev[(size_t)i * n + k];
        }
    // (n > 528: more tiles than the prefetched two per wave)
    for (int tix = la ? ntile : 1 + wg * nw + wid + TPW * NWG * nw; tix < ntile; tix += NWG * nw) {
        int tc = (int)((sqrtf(8.0f * (float)tix + 1.0f) - 1.0f) * 0.5f);
        while ((tc + 1) * (tc + 2) / 2 <= tix) ++tc;
        while (tc * (tc + 1) / 2 > tix) --tc;
        const int tr = tix - tc * (tc + 1) / 2;
        const int cb = t0 + tc * 16, rb = t0 + tr * 16;
        const int crow = cb + (lane >> 4), rcol = rb + (lane & 15), ca = cb + (lane & 15);
        v4d_t a2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = crow + 4 * q;
            a2[q] = (c < n && rcol <= c) ? R[(size_t)c * n + rcol] : 0.0;
        }
#pragma unroll
        for (int kk = 0; kk < NB / 4; ++kk) {
            const int k = kk * 4 + (lane >> 4);
            const double av = (k < nbk && ca < n) ? -panel[(size_t)k * n + ca] : 0.0;
            const double bv = (k < nbk && rcol < n) ? panel[(size_t)k * n + rcol] : 0.0;
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, a2, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = crow + 4 * q;
            if (c < n && rcol <= c) R[(size_t)c * n + rcol] = a2[q];
        }
    }
}

template <int NB>
__global__ void __launch_bounds__(1024)
k_chol_mc_end(int n, double *__restrict__ Rall, LmVecs v, const double *__restrict__ side, const int32_t *__restrict__ bad,
              const double *__restrict__ xall, LmState *__restrict__ st, double factor, double gtol)
{
    __shared__ double red[64];
    extern __shared__ double qs_end[];                          // n doubles
    const int p = blockIdx.x;
    LmState *s = st + p;
    if (s->stage != ST_HAVE_JAC) return;
    const int tid = threadIdx.x, BS = blockDim.x;
    if (bad[p]) {
        if (tid == 0) s->stage = ST_NEED_PCHOL;                 // let the pivoted kernel decide (it may ask for QR)
        return;
    }
    double *R = Rall + (size_t)p * n * n;
    const int nsteps = (n + NB - 1) / NB, pj = (nsteps - 1) * NB;
    const double *prev = side + ((size_t)p * 2 + ((nsteps - 1) & 1)) * NB * n;
    for (int k = pj + tid; k < n; k += BS)                       // the last step's diagonal block
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (k >= pj + i) R[(size_t)k * n + pj + i] = prev[(size_t)i * n + k];
    __syncthreads();
    if (tid == 0) { s->factor_kind = 0; s->pivoted = 0; }
    lm_head<false>(n, R, n, v.ipvt + (size_t)p * n, v.acnorm + (size_t)p * n, v.qtf + (size_t)p * n, xall + (size_t)p * n,
                   v.diag + (size_t)p * n, v.diag_prev + (size_t)p * n, s, factor, gtol, ST_NE_READY, red, nullptr, qs_end);
}

// ---------------------------------------------------------------------------
// lmfactor (MINPACK qrfac) + Q^T f, faithful: same pivot rule, same norm down-date
// with the 0.05 (rdiag/wa)^2 <= eps recompute test, same reflector scaling.  One
// workgroup per problem, the m-by-n Jacobian in global memory; a wave owns a trailing
// column per step (dot, axpy, norm down-date), so a step costs a handful of barriers.
// Only reduction order differs from the CPU path.
// Dynamic LDS: (3n + 64) doubles + 64 ints.
// Rout: n-by-n (ld n) receives R (upper, diagonal = rdiag) for lmpar.
// ---------------------------------------------------------------------------
static __global__ void __launch_bounds__(1024)
k_qr_factor(int m, int n, double *__restrict__ Jall, const double *__restrict__ fall,
            double *__restrict__ Rall, LmVecs v, double *__restrict__ wa4all,
            double *__restrict__ scratch_all /* [nprob][m], used when inner_pass > 0 */,
            const double *__restrict__ xall, LmState *__restrict__ st, double factor,
            double gtol, int standalone)
{
    extern __shared__ double smem[];
    const int p = blockIdx.x;
    LmState *s = st ? st + p : nullptr;
    if (s && s->stage != ST_NEED_QR) return;
    const int tid = threadIdx.x, BS = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = BS >> 6;
    double *rdiag = smem;           // n
    double *wa = smem + n;          // n
    double *red = smem + 2 * n;     // 64
    int *redi = reinterpret_cast<int *>(red + 32);
    double *a = Jall + (size_t)p * m * n;
    int32_t *ipvt = v.ipvt + (size_t)p * n;
    double *acnorm = v.acnorm + (size_t)p * n;
    double *qtf = v.qtf + (size_t)p * n;
    const int minmn = m < n ? m : n;
    const double p05 = 5.0e-2;

    // initial column norms (:611-616), wave per column
    for (int j = wid; j < n; j += nw) {
        const double *col = a + (size_t)j * m;
        double sq = 0.0;
        for (int i = lane; i < m; i += 64) sq = sq + col[i] * col[i];
        sq = wave_reduce_sum(sq);
        if (lane == 0) {
            const double nr = sqrt(sq);
            acnorm[j] = nr; rdiag[j] = nr; wa[j] = nr; ipvt[j] = j;
        }
    }
    __syncthreads();

    for (int j = 0; j < minmn; ++j) {
        // pivot (:622-637)
        double bv = 0.0;
        int bk = 0x7fffffff;
        for (int k = j + tid; k < n; k += BS) {
            const double d = rdiag[k];
            if (bk == 0x7fffffff || d > bv) { bv = d; bk = k; }
        }
        const int kmax = block_argmax_first(bv, bk, red, redi);
        if (kmax != j) {
            double *cj = a + (size_t)j * m, *ck = a + (size_t)kmax * m;
            for (int i = tid; i < m; i += BS) { double t = cj[i]; cj[i] = ck[i]; ck[i] = t; }
            if (tid == 0) {
                rdiag[kmax] = rdiag[j];
                wa[kmax] = wa[j];
                int32_t t = ipvt[j]; ipvt[j] = ipvt[kmax]; ipvt[kmax] = t;
            }
            __syncthreads();
        }
        // reflector (:642-646)
        double *cj = a + (size_t)j * m;
        double sq = 0.0;
        for (int i = j + tid; i < m; i += BS) sq = sq + cj[i] * cj[i];
        double ajnorm = sqrt(block_reduce_sum(sq, red));
        if (ajnorm != 0.0) {
            if (cj[j] < 0.0) ajnorm = -ajnorm;
            __syncthreads();                       // everyone has read cj[j]
            for (int i = j + tid; i < m; i += BS) {
                double t = cj[i] / ajnorm;
                if (i == j) t = t + 1.0;
                cj[i] = t;
            }
            __syncthreads();
            const double ajj = cj[j];
            // trailing columns (:652-662), wave per column
            for (int k = j + 1 + wid; k < n; k += nw) {
                double *ck = a + (size_t)k * m;
                double sm = 0.0;
                for (int i = j + lane; i < m; i += 64) sm = sm + cj[i] * ck[i];
                sm = wave_reduce_sum(sm);
                sm = __shfl(sm, 0, 64);
                const double temp = sm / ajj;
                for (int i = j + lane; i < m; i += 64) ck[i] = ck[i] - temp * cj[i];
                double rk = rdiag[k];
                if (rk != 0.0) {                   // wave-uniform
                    const double ajk = __shfl((lane == 0) ? ck[j] : 0.0, 0, 64);
                    const double t2 = ajk / rk;
                    rk = rk * sqrt(fmax(0.0, 1.0 - t2 * t2));
                    const double q = rk / wa[k];
                    if (!(p05 * (q * q) > NLH_EPS)) {
                        double s2 = 0.0;
                        for (int i = j + 1 + lane; i < m; i += 64) s2 = s2 + ck[i] * ck[i];
                        s2 = wave_reduce_sum(s2);
                        rk = sqrt(__shfl(s2, 0, 64));
                        if (lane == 0) wa[k] = rk;
                    }
                    if (lane == 0) rdiag[k] = rk;
                }
            }
        }
        __syncthreads();
        if (tid == 0) rdiag[j] = -ajnorm;          // :665
        __syncthreads();
    }

    // Q^T f (:241-253).  On a fallback after a rejected trial the caller's wa4 still
    // holds the rejected residual (lmpar deviation A reads its tail), so sweep a scratch.
    const bool first = (!s) || (s->inner_pass == 0);
    double *w4 = first ? (wa4all + (size_t)p * m) : (scratch_all + (size_t)p * m);
    const double *f = fall + (size_t)p * m;
    for (int i = tid; i < m; i += BS) w4[i] = f[i];
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        double *cj = a + (size_t)j * m;
        const double ajj = (j < m) ? cj[j] : 0.0;
        if (j < m && ajj != 0.0) {
            double sm = 0.0;
            for (int i = j + tid; i < m; i += BS) sm = sm + cj[i] * w4[i];
            sm = block_reduce_sum(sm, red);
            const double temp = -sm / ajj;
            for (int i = j + tid; i < m; i += BS) w4[i] = w4[i] + cj[i] * temp;
        }
        __syncthreads();
        if (tid == 0 && j < m) { cj[j] = rdiag[j]; qtf[j] = w4[j]; }
    }
    __syncthreads();
    // R for lmpar: strict upper from the factored Jacobian, diagonal = rdiag
    double *R = Rall + (size_t)p * n * n;
    for (int e = tid; e < n * n; e += BS) {
        const int i = e % n, c = e / n;
        if (i <= c && i < m) R[e] = a[(size_t)c * m + i];
    }
    for (int k = tid; k < n; k += BS) v.rdiag[(size_t)p * n + k] = rdiag[k];
    if (first && s) {
        double tq = 0.0;
        for (int i = n + tid; i < m; i += BS) tq = tq + w4[i] * w4[i];
        tq = block_reduce_sum(tq, red);
        if (tid == 0) s->tailsq = tq;
    }
    __syncthreads();
    if (standalone || !s) return;
    if (tid == 0) { s->factor_kind = 1; s->qr_count += 1; }
    if (first) {
        // redo the head from the pre-head scaling (the normal-equations head may have run)
        if (s->iter > 1 && s->head_done) {
            for (int j = tid; j < n; j += BS) v.diag[(size_t)p * n + j] = v.diag_prev[(size_t)p * n + j];
            __syncthreads();
        }
        lm_head<false>(n, R, n, ipvt, acnorm, qtf, xall + (size_t)p * n, v.diag + (size_t)p * n,
                       v.diag_prev + (size_t)p * n, s, factor, gtol, ST_QR_READY, red, nullptr);
    } else {
        if (tid == 0) s->stage = ST_QR_READY;
    }
}
