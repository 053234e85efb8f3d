// nlh_kernels_bfgs.h -- dense kernels of bfgs%solve (src/nonlin_optimize.f90:557-770) and the
// finite-difference gradient of fcnnvar_helper (src/nonlin_multi_var.f90:182-246) for the device model.
//
// The reference takes B = R^T R, DSYMV, the rank-one Cholesky update / downdate, the Cholesky
// factorisation and the triangular solves from the third-party linalg library (BLAS / LAPACK / qrupdate,
// unpinned).  The CPU restatement defines them with ascending-index sums; the kernels perform the same
// operations on every element, so R and every iterate are bit-identical to it.
// R is kept ROW-major (Rt[j*n + c] = R(j,c), upper triangular): the update / downdate sweeps and the
// forward solve touch one row of R across columns per step.
#pragma once
#include "nlh_common.h"
#include "nlh_kernels_broyden.h"

// B <- R^T R (tri_mtx_mult(.true., 1, r, 0, b), :709): thread per entry (i <= j), sum over k ascending;
// both triangles of the column-major B are written.
static __global__ void __launch_bounds__(256)
k_bf_rtr(int n, const double *__restrict__ Rt, double *__restrict__ B, const LmState *__restrict__ gst, int gwant)
{
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y, p = blockIdx.z;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    if (i > j || i >= n) return;
    Rt += (size_t)p * n * n; B += (size_t)p * n * n;
    double t = 0.0;
    for (int k = 0; k <= i; ++k) t = t + Rt[(size_t)k * n + i] * Rt[(size_t)k * n + j];
    B[(size_t)j * n + i] = t;
    B[(size_t)i * n + j] = t;
}

// x <- R^-T x (DTRSV 'U','T','N'), one workgroup: x_j loses R(i,j) x_i for i ascending, which is the dot form's
// order of subtractions.  Dynamic LDS: n doubles.
static __global__ void __launch_bounds__(1024)
k_bf_solve_upper_t(int n, const double *__restrict__ Rt, double *__restrict__ x, const LmState *__restrict__ gst, int gwant)
{
    extern __shared__ double xs[];
    const int tid = threadIdx.x, BS = blockDim.x, p = blockIdx.x;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    Rt += (size_t)p * n * n; x += (size_t)p * n;
    for (int i = tid; i < n; i += BS) xs[i] = x[i];
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        const double xi = xs[i] / Rt[(size_t)i * n + i];
        __syncthreads();
        for (int j = i + 1 + tid; j < n; j += BS) xs[j] = xs[j] - Rt[(size_t)i * n + j] * xi;
        if (tid == 0) xs[i] = xi;
        __syncthreads();
    }
    for (int i = tid; i < n; i += BS) x[i] = xs[i];
}

// R1^T R1 = R^T R + u u^T (cholesky_rank1_update, qrupdate DCH1UP, :721).  One workgroup, thread per
// column (NC columns per thread when n > blockDim): column c carries its u value through the rotations
// 0 .. c-1 and then generates rotation c.  u is consumed.  Dynamic LDS: 2n doubles.
template <int NC>
__global__ void __launch_bounds__(1024)
k_bf_chol_update(int n, double *__restrict__ Rt, const double *__restrict__ u, const LmState *__restrict__ gst, int gwant)
{
    extern __shared__ double cs[];                 // c[n], s[n]
    const int tid = threadIdx.x, BS = blockDim.x, p = blockIdx.x;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    Rt += (size_t)p * n * n; u += (size_t)p * n;
    double ui[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) { const int c = tid + q * BS; ui[q] = (c < n) ? u[c] : 0.0; }
    for (int j = 0; j < n; ++j) {
        const int oq = j / BS, ot = j - oq * BS;
        if (tid == ot) {
#pragma unroll
            for (int q = 0; q < NC; ++q)
                if (q == oq) {
                    double cj, sj, rr;
                    givens_dev(Rt[(size_t)j * n + j], ui[q], cj, sj, rr);
                    cs[j] = cj; cs[n + j] = sj;
                    Rt[(size_t)j * n + j] = rr;
                }
        }
        __syncthreads();
        const double cj = cs[j], sj = cs[n + j];
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int c = tid + q * BS;
            if (c > j && c < n) {
                const double rjc = Rt[(size_t)j * n + c];
                const double t = cj * rjc + sj * ui[q];
                ui[q] = cj * ui[q] - sj * rjc;
                Rt[(size_t)j * n + c] = t;
            }
        }
    }
}

// Downdate, first half (qrupdate DCH1DN): given v = R^-T u (k_bf_solve_upper_t), rho = sqrt(1 - ||v||^2) with
// NORM2 as the flang runtime evaluates it, then the rotations from the bottom.  info = 1: not positive definite.
static __global__ void k_bf_downdate_rot(int n, double *__restrict__ v, double *__restrict__ c, int *__restrict__ info, const LmState *__restrict__ gst, int gwant)
{
    const int p = blockIdx.x;
    if (threadIdx.x != 0) return;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    v += (size_t)p * n; c += (size_t)p * n; info += p;
    double mx = 0.0, s = 0.0;
    for (int i = 0; i < n; ++i) {
        const double a = fabs(v[i]);
        if (mx == 0.0) mx = a;
        else if (a > mx) { const double t = mx / a, tsq = t * t; s = s * tsq; s = s + tsq; mx = a; }
        else if (a != 0.0) { const double t = a / mx; s = s + t * t; }
    }
    double rho = mx * sqrt(1.0 + s);
    rho = 1.0 - rho * rho;
    if (rho <= 0.0) { *info = 1; return; }
    *info = 0;
    rho = sqrt(rho);
    for (int i = n - 1; i >= 0; --i) {
        double ci, si, rr;
        givens_dev(rho, v[i], ci, si, rr);
        c[i] = ci; v[i] = si;
        rho = rr;
    }
}

// Downdate, second half: thread per column i, rows i .. 0.
static __global__ void __launch_bounds__(256)
k_bf_downdate_apply(int n, double *__restrict__ Rt, const double *__restrict__ c, const double *__restrict__ s,
                    const int *__restrict__ info, const LmState *__restrict__ gst, int gwant)
{
    extern __shared__ double cs[];
    const int p = blockIdx.y;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    Rt += (size_t)p * n * n; c += (size_t)p * n; s += (size_t)p * n; info += p;
    if (*info) return;
    for (int k = threadIdx.x; k < n; k += 256) { cs[k] = c[k]; cs[n + k] = s[k]; }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double ui = 0.0;
    for (int j = i; j >= 0; --j) {
        const double rji = Rt[(size_t)j * n + i];
        const double t = cs[j] * ui + cs[n + j] * rji;
        Rt[(size_t)j * n + i] = cs[j] * rji - cs[n + j] * ui;
        ui = t;
    }
}

// R <- chol(B), upper (cholesky_factor(b, .true.), DPOTF2 'U', :724), row-major result with zeros below the
// diagonal.  One workgroup, thread per column (NC per thread).  info = 1-based index of a non-positive pivot.
template <int NC>
__global__ void __launch_bounds__(1024)
k_bf_chol_factor(int n, const double *__restrict__ B, double *__restrict__ Rt, int *__restrict__ info, const LmState *__restrict__ gst, int gwant)
{
    extern __shared__ double colj[];               // n: column j of R above the diagonal
    __shared__ double ajj_sh;
    __shared__ int bad;
    const int tid = threadIdx.x, BS = blockDim.x, p = blockIdx.x;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    B += (size_t)p * n * n; Rt += (size_t)p * n * n; info += p;
    if (tid == 0) bad = 0;
    for (size_t e = tid; e < (size_t)n * n; e += BS) {          // row-major copy of the symmetric B
        const int r = (int)(e / n), c = (int)(e % n);
        Rt[e] = (r <= c) ? B[(size_t)c * n + r] : 0.0;
    }
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        for (int k = tid; k < j; k += BS) colj[k] = Rt[(size_t)k * n + j];      // strided, all threads
        __syncthreads();
        if (tid == 0) {                             // ordered sum from LDS
            double a = Rt[(size_t)j * n + j];
            int k = 0;
            for (; k + 16 <= j; k += 16) {
                double t[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) t[u] = colj[k + u];
#pragma unroll
                for (int u = 0; u < 16; ++u) a = a - t[u] * t[u];
            }
            for (; k < j; ++k) a = a - colj[k] * colj[k];
            if (!(a > 0.0)) bad = j + 1;
            else { a = sqrt(a); Rt[(size_t)j * n + j] = a; }
            ajj_sh = a;
        }
        __syncthreads();
        if (bad) break;
        const double ajj = ajj_sh;
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int c = tid + q * BS;
            if (c > j && c < n) {
                double t = Rt[(size_t)j * n + c];
                int k = 0;
                for (; k + 16 <= j; k += 16) {      // 16 coalesced loads in flight per thread
                    double v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = Rt[(size_t)(k + u) * n + c];
#pragma unroll
                    for (int u = 0; u < 16; ++u) t = t - colj[k + u] * v[u];
                }
                for (; k < j; ++k) t = t - colj[k] * Rt[(size_t)k * n + c];
                Rt[(size_t)j * n + c] = t / ajj;
            }
        }
        __syncthreads();
    }
    if (tid == 0) *info = bad;
}

// The same factorisation BLOCKED (round 4; n <= 1024, a thread per column): the column-at-a-time form above pays
// global-memory round trips in every one of its n steps (n = 256: 1.55 ms a call, and bfgs calls it whenever the
// curvature test fails).  Rows are taken BFC_W at a time.  For the panel's rows every thread first subtracts the
// products of the rows above the panel (its own column's entries stream from global memory eight ahead, the panel
// columns' entries of those rows sit in LDS) -- no barrier in that loop --, then the wave that holds the panel's own
// columns finishes the BFC_W x BFC_W diagonal block with v_readlane broadcasts (no barrier either), and everybody else
// solves its BFC_W rows against that block from LDS.  Every element still receives a(j,c) - sum_k r(k,j) r(k,c) with k
// ascending, a separate multiply and subtract per term, then the division by r(j,j): the bits of the loop above.
// A non-positive pivot stops at the same row with the same rows written.  Dynamic LDS: bf_chol_lds(n).
#define BFC_W 16
static inline size_t bf_chol_lds(int n) { return sizeof(double) * ((size_t)n * BFC_W + BFC_W * BFC_W + (size_t)(((n + 63) / 64) * 64) * BFC_W); }

// G thread groups share a column in the prefix phase, the only phase with real arithmetic (one wave per SIMD reads two LDS
// entries, waits, uses them: 0.3 us per row above the panel, 770 us a call at n = 256): group g forms the products for
// rows g * (W / G) ... of the panel only, the partial columns meet in LDS and group 0 goes on alone.
// blockDim = G * CT, CT = n rounded up to 64.
template <int G>
static __global__ void __launch_bounds__(1024)
k_bf_chol_blocked(int n, const double *__restrict__ B, double *__restrict__ Rt, int *__restrict__ info, const LmState *__restrict__ gst, int gwant)
{
    constexpr int W = BFC_W, RW = W / G;
    extern __shared__ double bfc_sm[];
    double *pre = bfc_sm;                            // pre[k * W + jj] = r(k, jb + jj), k < jb
    double *P = bfc_sm + (size_t)n * W;              // P[kk * W + jj] = r(jb + kk, jb + jj)
    double *colbuf = P + W * W;                      // colbuf[c * W + jj]: the panel rows of column c after the prefix phase
    __shared__ int bad;
    const int tid = threadIdx.x, BS = blockDim.x, p = blockIdx.x, CT = BS / G, c = tid % CT, g = tid / CT, lane = tid & 63;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    B += (size_t)p * n * n; Rt += (size_t)p * n * n; info += p;
    if (tid == 0) bad = 0;
    // row-major copy of the symmetric B, zeros below the diagonal: a wave per row, 64 columns at a time, four loads in flight
    for (int r = tid >> 6; r < n; r += BS >> 6) {
        for (int c0 = 0; c0 < n; c0 += 256) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int cc = c0 + u * 64 + lane; v[u] = B[(size_t)(cc < n ? cc : n - 1) * n + r]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cc = c0 + u * 64 + lane;
                asm volatile("" : "+v"(v[u]));
                if (cc < n) Rt[(size_t)r * n + cc] = (r <= cc) ? v[u] : 0.0;
            }
        }
    }
    const int cc_ = c < n ? c : n - 1;
    for (int jb = 0; jb < n; jb += W) {
        const int w = n - jb < W ? n - jb : W;
        double part[RW];                             // this group's rows of the panel
#pragma unroll
        for (int q = 0; q < RW; ++q) { const int jj = g * RW + q; part[q] = B[(size_t)cc_ * n + jb + (jj < w ? jj : w - 1)]; }
        __syncthreads();                            // the rows of the earlier panels are in memory
        for (int e0 = tid; e0 < jb * W; e0 += 4 * BS) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * BS, ec = e < jb * W ? e : jb * W - 1, jj = ec % W;
                v[u] = Rt[(size_t)(ec / W) * n + jb + (jj < w ? jj : w - 1)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * BS;
                asm volatile("" : "+v"(v[u]));
                if (e < jb * W) pre[e] = v[u];
            }
        }
#pragma unroll
        for (int q = 0; q < RW; ++q) asm volatile("" : "+v"(part[q]));
        __syncthreads();
        // rows above the panel: k ascending, this column's entries eight ahead
        if (jb > 0 && c >= jb) {                  // (finished columns have nothing to do)
            double rk[8], rn[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) rk[u] = Rt[(size_t)u * n + cc_];
            for (int k = 0; k < jb; k += 8) {
                const int kn = k + 8 < jb ? k + 8 : k;
#pragma unroll
                for (int u = 0; u < 8; ++u) rn[u] = Rt[(size_t)(kn + u) * n + cc_];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    asm volatile("" : "+v"(rk[u]));
                    const double *pk = pre + (size_t)(k + u) * W + g * RW;
#pragma unroll
                    for (int q = 0; q < RW; ++q) part[q] = part[q] - pk[q] * rk[u];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) rk[u] = rn[u];
            }
        }
        // the groups' rows meet in LDS; group 0 goes on with the whole column
#pragma unroll
        for (int q = 0; q < RW; ++q) colbuf[(size_t)c * W + g * RW + q] = part[q];
        __syncthreads();
        double acc[W];
        if (g == 0) {
#pragma unroll
            for (int jj = 0; jj < W; ++jj) acc[jj] = colbuf[(size_t)c * W + jj];
        }
        // the diagonal block: the wave that holds columns jb .. jb + W - 1 (all its lanes run the same steps, so its
        // other columns are solved on the way); r(jb + kk, jb + jj) comes from lane L0 + jj's registers
        int stop = w;
        if (g == 0 && (tid >> 6) == (jb >> 6)) {
            const int L0 = jb & 63;
#pragma unroll
            for (int jj = 0; jj < W; ++jj) {
                if (jj < stop) {
                    double t = acc[jj];
#pragma unroll
                    for (int kk = 0; kk < jj; ++kk) {
                        const double pk = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(acc[kk]), L0 + jj),
                                                           __builtin_amdgcn_readlane(__double2loint(acc[kk]), L0 + jj));
                        t = t - pk * acc[kk];
                    }
                    const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t), L0 + jj),
                                                      __builtin_amdgcn_readlane(__double2loint(t), L0 + jj));
                    if (!(d > 0.0)) {                // (uniform) not positive definite at row jb + jj
                        stop = jj;
                        if (lane == L0 + jj) bad = jb + jj + 1;
                    } else {
                        const double rjj = sqrt(d);
                        acc[jj] = (lane == L0 + jj) ? rjj : t / rjj;
                    }
                }
            }
            if (lane >= L0 && lane < L0 + W) {
#pragma unroll
                for (int jj = 0; jj < W; ++jj) P[jj * W + (lane - L0)] = acc[jj];
            }
        }
        __syncthreads();
        const int badrow = bad;                      // (uniform after the barrier)
        const int lim = badrow ? badrow - 1 - jb : w;
        if (g == 0 && (tid >> 6) != (jb >> 6)) {
#pragma unroll
            for (int jj = 0; jj < W; ++jj) {
                if (jj < lim) {
                    double t = acc[jj];
#pragma unroll
                    for (int kk = 0; kk < jj; ++kk) t = t - P[kk * W + jj] * acc[kk];
                    acc[jj] = t / P[jj * W + jj];
                }
            }
        }
        if (g == 0 && c < n) {
#pragma unroll
            for (int jj = 0; jj < W; ++jj)
                if (jj < lim && c >= jb + jj) Rt[(size_t)(jb + jj) * n + c] = acc[jj];
        }
        if (badrow) break;
    }
    __syncthreads();
    if (tid == 0) *info = bad;
}

// R <- temp * I (DLASET, :705)
static __global__ void __launch_bounds__(256)
k_bf_scaled_identity(int n, double temp, double *__restrict__ Rt, const double *__restrict__ tempall, size_t tstride,
                     const int32_t *__restrict__ iterall, size_t istride, const LmState *__restrict__ gst, int gwant)
{
    // tempall: the problem's own factor at tempall[p * tstride]; iterall: only for problems in their first iteration
    const int p = blockIdx.y;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    if (iterall && iterall[(size_t)p * istride] != 1) return;
    if (tempall) temp = tempall[(size_t)p * tstride];
    Rt += (size_t)p * n * n;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e < (size_t)n * n) Rt[e] = (e / n == e % n) ? temp : 0.0;
}

// f_j = 0.5 * sum_i P(i,j)^2 for every column of the residual panel (objective of the device model at the
// n perturbed points), sum over i ascending; then g_j = (f_j - f0) / h_j (:240).
// Round 4: a WAVE per column, the ordered sum down its lanes (ordered_sum_wave: the squares are formed by every lane
// for its own run of the column, read straight from global memory; 2.2 ns per term instead of the 10 ns of a thread
// that reads its terms one LDS round trip at a time) -- n workgroups instead of n / 64, every load unconditional.
static __global__ void __launch_bounds__(64)
k_bf_fd_gradient(int m, int n, const double *__restrict__ P, const double *__restrict__ x, double f0,
                 double *__restrict__ g, const double *__restrict__ f0all, size_t fstride, const LmState *__restrict__ gst, int gwant)
{
    const int k = blockIdx.x, p = blockIdx.y;
    if (gst && gst[p].stage != gwant) return;                    // lock-step batches: only problems in this stage
    if (f0all) f0 = f0all[(size_t)p * fstride];    // the problem's own objective value
    const double *col = P + (size_t)p * m * n + (size_t)k * m;
    x += (size_t)p * n; g += (size_t)p * n;
    const double acc = ordered_sum_wave<32>([&](int i) { const double r = col[i]; return r * r; }, m, 0.0);
    if (threadIdx.x == 0) g[k] = (0.5 * acc - f0) / fd_step(x[k]);
}
