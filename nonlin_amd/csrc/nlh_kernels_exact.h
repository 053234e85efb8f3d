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

// Column-major m-by-n  ->  row-major m-by-n, 32x32 tiles through LDS.
__global__ void __launch_bounds__(256)
k_transpose(int m, int n, const double *__restrict__ J, double *__restrict__ Jt,
            const LmState *__restrict__ st, int want_stage)
{
    __shared__ double tile[32][33];
    const int p = blockIdx.z;
    if (st && st[p].stage != want_stage) return;
    const double *Jp = J + (size_t)p * m * n;
    double *Tp = Jt + (size_t)p * m * n;
    const int i0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int r = ty; r < 32; r += 8) {                            // r = column offset, tx = row offset
        const int i = i0 + tx, k = k0 + r;
        tile[r][tx] = (i < m && k < n) ? Jp[(size_t)k * m + i] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {                            // r = row offset, tx = column offset
        const int i = i0 + r, k = k0 + tx;
        if (i < m && k < n) Tp[(size_t)i * n + k] = tile[tx][r];
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
        const double nr = norm2_flang_serial([&](int i) { return a[(size_t)i * n + k]; }, m);
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
