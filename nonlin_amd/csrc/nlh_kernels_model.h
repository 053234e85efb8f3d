// nlh_kernels_model.h -- residual ("vecfcn") evaluation of the dense-quadratic device
// model, the n perturbed evaluations of the forward-difference Jacobian, and the
// forward-difference column write itself.
//
// Reference: vfh_jac_fcn, src/nonlin_multi_eqn_mult_var.f90:198-277.
// Bit-parity rule: every residual value must equal the CPU path's bit for bit, so a
// row's sum runs over j ascending with a separate multiply and add (the translation
// unit is built with -ffp-contract=off) and r = (u + (gamma*u)*u) - b.
#pragma once
#include "nlh_common.h"

// f = F(x) for one x per problem.  One thread per residual row: lanes read a column
// of A contiguously (column-coalesced), x is broadcast from LDS.
// part (optional): per-block partial sums {sum f^2, sum_{i>=n} f^2} in fixed order.
template <int BS>
__global__ void __launch_bounds__(BS)
k_dq_residual(int m, int n, const double *__restrict__ A, const double *__restrict__ b,
              double gamma, const double *__restrict__ xsrc, double *__restrict__ fout,
              double *__restrict__ part, const LmState *__restrict__ st, int want_stage)
{
    extern __shared__ double smem[];
    const int p = blockIdx.y;
    if (st && st[p].stage != want_stage) return;
    double *xs = smem;            // n
    double *red = smem + n;       // BS/64 + 1
    const double *Ap = A + (size_t)p * m * n;
    const double *xp = xsrc + (size_t)p * n;
    for (int k = threadIdx.x; k < n; k += BS) xs[k] = xp[k];
    __syncthreads();

    const int i = blockIdx.x * BS + threadIdx.x;
    double r = 0.0;
    if (i < m) {
        double u = 0.0;
        const double *a = Ap + i;
        int k = 0;
        for (; k + 16 <= n; k += 16) {
            double av[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) av[q] = a[(size_t)(k + q) * m];
#pragma unroll
            for (int q = 0; q < 16; ++q) u = u + av[q] * xs[k + q];
        }
        for (; k < n; ++k) u = u + a[(size_t)k * m] * xs[k];
        r = (u + (gamma * u) * u) - b[(size_t)p * m + i];
        fout[(size_t)p * m + i] = r;
    }
    if (part) {
        double sq = (i < m) ? r * r : 0.0;
        double tq = (i < m && i >= n) ? sq : 0.0;
        double s = block_reduce_sum(sq, red);
        double t = block_reduce_sum(tq, red);
        if (threadIdx.x == 0) {
            part[((size_t)p * gridDim.x + blockIdx.x) * 2 + 0] = s;
            part[((size_t)p * gridDim.x + blockIdx.x) * 2 + 1] = t;
        }
    }
}

// Two rows per thread (m even, 16-byte aligned operands): the same row sums, but A is read with 16-byte accesses,
// which this part serves at a markedly higher rate than 8-byte ones.  A block still covers 2*BS rows, so the
// per-block partial sums keep their layout.
template <int BS>
__global__ void __launch_bounds__(BS)
k_dq_residual2(int m, int n, const double *__restrict__ A, const double *__restrict__ b,
               double gamma, const double *__restrict__ xsrc, double *__restrict__ fout,
               double *__restrict__ part, const LmState *__restrict__ st, int want_stage)
{
    extern __shared__ double smem[];
    const int p = blockIdx.y;
    if (st && st[p].stage != want_stage) return;
    double *xs = smem;            // n
    double *red = smem + n;       // BS/64 + 1
    const double *Ap = A + (size_t)p * m * n;
    const double *xp = xsrc + (size_t)p * n;
    for (int k = threadIdx.x; k < n; k += BS) xs[k] = xp[k];
    __syncthreads();

    const int i = (blockIdx.x * BS + threadIdx.x) * 2;
    double r0 = 0.0, r1 = 0.0;
    if (i < m) {                                    // m even: i + 1 < m as well
        double u0 = 0.0, u1 = 0.0;
        const double *a = Ap + i;
        int k = 0;
        for (; k + 16 <= n; k += 16) {
            double2 av[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) av[q] = *reinterpret_cast<const double2 *>(a + (size_t)(k + q) * m);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const double xk = xs[k + q];
                u0 = u0 + av[q].x * xk;
                u1 = u1 + av[q].y * xk;
            }
        }
        for (; k < n; ++k) {
            const double2 av = *reinterpret_cast<const double2 *>(a + (size_t)k * m);
            u0 = u0 + av.x * xs[k];
            u1 = u1 + av.y * xs[k];
        }
        const double2 bv = *reinterpret_cast<const double2 *>(b + (size_t)p * m + i);
        r0 = (u0 + (gamma * u0) * u0) - bv.x;
        r1 = (u1 + (gamma * u1) * u1) - bv.y;
        *reinterpret_cast<double2 *>(fout + (size_t)p * m + i) = make_double2(r0, r1);
    }
    if (part) {
        const double q0 = (i < m) ? r0 * r0 : 0.0, q1 = (i < m) ? r1 * r1 : 0.0;
        const double sq = q0 + q1;
        const double tq = ((i < m && i >= n) ? q0 : 0.0) + ((i < m && i + 1 >= n) ? q1 : 0.0);
        double s = block_reduce_sum(sq, red);
        double t = block_reduce_sum(tq, red);
        if (threadIdx.x == 0) {
            part[((size_t)p * gridDim.x + blockIdx.x) * 2 + 0] = s;
            part[((size_t)p * gridDim.x + blockIdx.x) * 2 + 1] = t;
        }
    }
}

// The n perturbed evaluations of vfh_jac_fcn (:267-273), all at once:
// P(i,j) = F_i(x + h_j e_j).  One thread per row, JT columns of the panel per thread.
// The sequential row sum of column j equals the unperturbed prefix for k < j, so a tile
// [j0, j0+JT) shares one prefix accumulator, forks inside the tile and then adds the same
// product to every accumulator: ~n^2/2 adds per row instead of n^2, still bit-identical
// to n independent evaluations.
// FUSE: the epilogue forms the forward-difference column (:274) itself, J(i,j) = (F_i(x + h_j e_j) - f0_i) / h_j,
// and writes it to P (which is then the Jacobian); the residual panel never reaches memory.
template <int BS, int JT, bool FUSE>
__global__ void __launch_bounds__(BS)
k_dq_panel(int m, int n, const double *__restrict__ A, const double *__restrict__ b,
           double gamma, const double *__restrict__ x, double *__restrict__ P,
           const double *__restrict__ f0, const LmState *__restrict__ st, int want_stage,
           int tld, int tcoff, size_t tst)
{
    extern __shared__ double smem[];
    const int p = blockIdx.z;
    if (st && st[p].stage != want_stage) return;
    double *xs = smem;                 // n
    const double *Ap = A + (size_t)p * m * n;
    const double *xp = x + (size_t)p * n;
    for (int k = threadIdx.x; k < n; k += BS) xs[k] = xp[k];
    __syncthreads();

    const int j0 = blockIdx.y * JT;
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= m) return;
    const int jt = min(JT, n - j0);    // valid columns in this tile
    const double *a = Ap + i;

    // Loads are issued PF rows ahead of their use: the row sums are serial recurrences, so without
    // this every iteration would wait out a full memory round trip.
    constexpr int PF = 8;
    double base = 0.0;
    {
        int k = 0;
        for (; k + PF <= j0; k += PF) {
            double av[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) av[u] = a[(size_t)(k + u) * m];
#pragma unroll
            for (int u = 0; u < PF; ++u) base = base + av[u] * xs[k + u];
        }
        for (; k < j0; ++k) base = base + a[(size_t)k * m] * xs[k];
    }

    double acc[JT];
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) acc[jj] = base;

#pragma unroll
    for (int kk = 0; kk < JT; ++kk) {
        if (kk < jt) {
            const int k = j0 + kk;
            const double av = a[(size_t)k * m];
            const double xk = xs[k];
            const double pr = av * xk;
            const double pp = av * (xk + fd_step(xk));   // x(j) = temp + h, :271
#pragma unroll
            for (int jj = 0; jj < JT; ++jj) acc[jj] = acc[jj] + (jj == kk ? pp : pr);
        }
    }
    {
        // the loads of the next PF columns are in flight while the JT * PF adds of the current ones run
        int k = j0 + JT;
        double av[PF], aw[PF];
        if (k + PF <= n) {
#pragma unroll
            for (int u = 0; u < PF; ++u) av[u] = a[(size_t)(k + u) * m];
        }
        for (; k + PF <= n; k += PF) {
            const bool more = k + 2 * PF <= n;
            if (more) {
#pragma unroll
                for (int u = 0; u < PF; ++u) aw[u] = a[(size_t)(k + PF + u) * m];
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const double pr = av[u] * xs[k + u];
#pragma unroll
                for (int jj = 0; jj < JT; ++jj) acc[jj] = acc[jj] + pr;
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < PF; ++u) av[u] = aw[u];
            }
        }
        for (; k < n; ++k) {
            const double pr = a[(size_t)k * m] * xs[k];
#pragma unroll
            for (int jj = 0; jj < JT; ++jj) acc[jj] = acc[jj] + pr;
        }
    }
    const double bi = b[(size_t)p * m + i];
    const double f0i = FUSE ? f0[(size_t)p * m + i] : 0.0;
    // tld > 0 (fused form only): the Jacobian goes straight into the exact factorisation's row-blocked working matrix
    // (nlh_qrx.hip, qrx_at: element (i, c) at ((i / 8) * tld + c) * 8 + i % 8, column j at c = tcoff + j) -- the eight
    // lanes of a row block complete a 64-byte sector per column -- instead of a column-major J that would then be
    // re-laid out.
    double *Pp = tld ? P + (size_t)p * tst + ((size_t)(i >> 3) * tld + tcoff) * 8 + (i & 7) : P + (size_t)p * m * n + i;
    const size_t cstride = tld ? (size_t)8 : (size_t)m;
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) {
        if (jj < jt) {
            const double u = acc[jj];
            const double r = (u + (gamma * u) * u) - bi;
            // written once, read once by the next kernel: non-temporal, so that it does not displace A (re-read n/JT times)
            if (FUSE) __builtin_nontemporal_store((r - f0i) / fd_step(xs[j0 + jj]), Pp + (size_t)(j0 + jj) * cstride);
            else __builtin_nontemporal_store(r, Pp + (size_t)(j0 + jj) * cstride);
        }
    }
}

// The forward-difference column write (:274): J(:,j) = (P(:,j) - f0) / h_j with a true
// division.  Pure streaming: reads the panel once, writes J once; f0 is held in
// registers across CJ columns.  Two rows per lane (16-byte accesses) when m is even.
template <int BS, int CJ, bool VEC2>
__global__ void __launch_bounds__(BS)
k_fd_jacobian(int m, int n, const double *__restrict__ P, const double *__restrict__ f0,
              const double *__restrict__ x, double *__restrict__ J,
              const LmState *__restrict__ st, int want_stage, const int32_t *__restrict__ list = nullptr, int nfull = 0)
{
    // list (the open device-residual path, nlh_devfcn.hip): the panel of compact slot blockIdx.z belongs to problem list[slot];
    // nfull > n: the panel holds a group of n columns of ONE problem of nfull columns (J and x arrive shifted to the group)
    const int p = list ? list[blockIdx.z] : blockIdx.z;
    if (st && st[p].stage != want_stage) return;
    const int ns = nfull ? nfull : n;
    const double *Pp = P + (size_t)blockIdx.z * m * n;
    double *Jp = J + (size_t)p * m * ns;
    const double *fp = f0 + (size_t)p * m;
    const double *xp = x + (size_t)p * ns;
    const int jbeg = blockIdx.y * CJ;
    const int jend = min(n, jbeg + CJ);
    if (VEC2) {
        const int i = (blockIdx.x * BS + threadIdx.x) * 2;
        if (i >= m) return;              // m even => i+1 < m
        const double2 f = *reinterpret_cast<const double2 *>(fp + i);
#pragma unroll 4
        for (int j = jbeg; j < jend; ++j) {
            const double h = fd_step(xp[j]);
            // streamed once: non-temporal so the panel and J do not displace A in L2 / Infinity Cache
            typedef double v2d __attribute__((ext_vector_type(2)));
            const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(Pp + (size_t)j * m + i));
            v2d o;
            o.x = (v.x - f.x) / h;
            o.y = (v.y - f.y) / h;
            __builtin_nontemporal_store(o, reinterpret_cast<v2d *>(Jp + (size_t)j * m + i));
        }
    } else {
        const int i = blockIdx.x * BS + threadIdx.x;
        if (i >= m) return;
        const double f = fp[i];
        for (int j = jbeg; j < jend; ++j) {
            const double h = fd_step(xp[j]);
            Jp[(size_t)j * m + i] = (Pp[(size_t)j * m + i] - f) / h;
        }
    }
}

// Analytic jacobianfcn of the model: J(i,j) = (1 + 2 gamma u_i) A(i,j).
template <int BS>
__global__ void __launch_bounds__(BS)
k_dq_jacobian(int m, int n, const double *__restrict__ A, double gamma,
              const double *__restrict__ x, double *__restrict__ J, const LmState *__restrict__ st, int want)
{
    extern __shared__ double smem[];
    const int p = blockIdx.y;
    if (st && st[p].stage != want) return;                       // lock-step batches: only problems in this stage
    double *xs = smem;
    const double *Ap = A + (size_t)p * m * n;
    const double *xp = x + (size_t)p * n;
    for (int k = threadIdx.x; k < n; k += BS) xs[k] = xp[k];
    __syncthreads();
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= m) return;
    const double *a = Ap + i;
    double u = 0.0;
    for (int k = 0; k < n; ++k) u = u + a[(size_t)k * m] * xs[k];
    const double s = 1.0 + 2.0 * gamma * u;
    double *Jp = J + (size_t)p * m * n + i;
    for (int k = 0; k < n; ++k) Jp[(size_t)k * m] = s * a[(size_t)k * m];
}
