// user_models.hip -- residual families written OUTSIDE libnonlin_hip.so, the way a user of the open device-residual
// path writes them (include/nonlin_hip.h: nlh_device_vecfcn / nlh_device_jacfcn): a HIP kernel
// that evaluates F at many points, a launcher with the library's callback signature that enqueues it on the stream it is
// handed, and -- for the parity tests -- the SAME arithmetic as a plain host function with nonlin's vecfcn shape
// (src/nonlin_multi_eqn_mult_var.f90:14-25), which the CPU oracle drives.  This file links nothing of the library.
//
// Only +, -, *, / occur (IEEE, correctly rounded on both sides; built with -ffp-contract=off), sums run in ascending index
// order, so device and host produce the same bits.
//
//   lorentz : least squares, a spectrum of K Lorentzian peaks, n = 3 K unknowns (a_k, c_k, w_k):
//             r_i = sum_k a_k / (1 + ((t_i - c_k) / w_k)^2)  -  y_i ,  i = 1 .. m
//   btri    : Broyden's tridiagonal system (More, Garbow, Hillstrom no. 30) with a per-problem constant:
//             F_i = (3 - 2 x_i) x_i - x_{i-1} - 2 x_{i+1} + c ,  x_0 = x_{n+1} = 0 ;  analytic tridiagonal Jacobian
//
// Build: hipcc -O2 -ffp-contract=off --offload-arch=gfx950 -fPIC -shared -o libuser_models.so user_models.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

extern "C" {

// ---------------------------------------------------------------- lorentz ----
typedef struct {            // device side: every problem of the batch
    int32_t nprob, m;
    double *dt, *dy;        // [nprob][m]
} lorentz_dev;

typedef struct {            // host side: ONE problem (what a vecfcn's `args` would carry)
    int32_t m;
    const double *t, *y;    // [m]
    int64_t ncalls;
} lorentz_host;

__host__ __device__ static inline double lorentz_row(int n, const double *x, double t, double y)
{
    double s = 0.0;
    for (int k = 0; k + 2 < n; k += 3) {
        const double d = (t - x[k + 1]) / x[k + 2];
        const double q = 1.0 + d * d;
        s = s + x[k] / q;
    }
    return s - y;
}

__global__ void __launch_bounds__(256)
k_lorentz(int m, int n, int nblk, const double *__restrict__ t, const double *__restrict__ y, const int32_t *__restrict__ dprob,
          const double *__restrict__ X, double *__restrict__ F)
{
    extern __shared__ double xs[];
    const int q = blockIdx.x / nblk, rb = blockIdx.x - q * nblk;
    const int p = dprob[q];
    for (int c = threadIdx.x; c < n; c += 256) xs[c] = X[(size_t)q * n + c];
    __syncthreads();
    const int i = rb * 256 + threadIdx.x;
    if (i < m) F[(size_t)q * m + i] = lorentz_row(n, xs, t[(size_t)p * m + i], y[(size_t)p * m + i]);
}

void *lorentz_create(int32_t nprob, int32_t m, const double *t, const double *y)
{
    lorentz_dev *c = (lorentz_dev *)calloc(1, sizeof *c);
    c->nprob = nprob; c->m = m;
    const size_t bytes = sizeof(double) * (size_t)nprob * m;
    if (hipMalloc(&c->dt, bytes) != hipSuccess || hipMalloc(&c->dy, bytes) != hipSuccess) { free(c); return NULL; }
    hipMemcpy(c->dt, t, bytes, hipMemcpyHostToDevice);
    hipMemcpy(c->dy, y, bytes, hipMemcpyHostToDevice);
    return c;
}

void lorentz_destroy(void *ctx)
{
    lorentz_dev *c = (lorentz_dev *)ctx;
    if (!c) return;
    hipFree(c->dt); hipFree(c->dy);
    free(c);
}

// nlh_device_vecfcn
int lorentz_launch(void *ctx, void *hip_stream, int32_t npoints, const int32_t *dprob, int32_t n, const double *dX, int32_t m, double *dF)
{
    const lorentz_dev *c = (const lorentz_dev *)ctx;
    if (!c || m != c->m || n % 3) return 1;
    const int nblk = (m + 255) / 256;
    hipLaunchKernelGGL(k_lorentz, dim3((unsigned)((size_t)npoints * nblk)), dim3(256), sizeof(double) * (size_t)n, (hipStream_t)hip_stream, m, n, nblk,
                       (const double *)c->dt, (const double *)c->dy, dprob, dX, dF);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// the same function as a host vecfcn (nlh_vecfcn / the oracle's callback type)
void lorentz_host_fcn(void *ctx, int32_t n, const double *x, int32_t m, double *f)
{
    lorentz_host *c = (lorentz_host *)ctx;
    c->ncalls += 1;
    for (int32_t i = 0; i < m; ++i) f[i] = lorentz_row(n, x, c->t[i], c->y[i]);
}

// ------------------------------------------------------------------- btri ----
typedef struct {
    int32_t nprob;
    double *dc;             // [nprob]
} btri_dev;

typedef struct {
    double c;
    int64_t ncalls, njcalls;
} btri_host;

__host__ __device__ static inline double btri_row(int n, const double *x, int i, double c)
{
    const double xm = i > 0 ? x[i - 1] : 0.0, xp = i + 1 < n ? x[i + 1] : 0.0;
    return (((3.0 - 2.0 * x[i]) * x[i] - xm) - 2.0 * xp) + c;
}

__host__ __device__ static inline double btri_jac(int n, const double *x, int i, int j)
{
    (void)n;
    if (j == i) return 3.0 - 4.0 * x[i];
    if (j == i - 1) return -1.0;
    if (j == i + 1) return -2.0;
    return 0.0;
}

__global__ void __launch_bounds__(256)
k_btri(int n, int nblk, const double *__restrict__ cs, const int32_t *__restrict__ dprob, const double *__restrict__ X, double *__restrict__ F)
{
    const int q = blockIdx.x / nblk, rb = blockIdx.x - q * nblk;
    const int i = rb * 256 + threadIdx.x;
    if (i < n) F[(size_t)q * n + i] = btri_row(n, X + (size_t)q * n, i, cs[dprob[q]]);
}

__global__ void __launch_bounds__(256)
k_btri_jac(int n, int nblk, const double *__restrict__ X, double *__restrict__ J)
{
    const int q = blockIdx.y, j = blockIdx.x / nblk, rb = blockIdx.x - j * nblk;
    const int i = rb * 256 + threadIdx.x;
    if (i < n) J[((size_t)q * n + j) * n + i] = btri_jac(n, X + (size_t)q * n, i, j);
}

void *btri_create(int32_t nprob, const double *c)
{
    btri_dev *b = (btri_dev *)calloc(1, sizeof *b);
    b->nprob = nprob;
    if (hipMalloc(&b->dc, sizeof(double) * (size_t)nprob) != hipSuccess) { free(b); return NULL; }
    hipMemcpy(b->dc, c, sizeof(double) * (size_t)nprob, hipMemcpyHostToDevice);
    return b;
}

void btri_destroy(void *ctx)
{
    btri_dev *b = (btri_dev *)ctx;
    if (!b) return;
    hipFree(b->dc);
    free(b);
}

int btri_launch(void *ctx, void *hip_stream, int32_t npoints, const int32_t *dprob, int32_t n, const double *dX, int32_t m, double *dF)
{
    const btri_dev *b = (const btri_dev *)ctx;
    if (!b || m != n) return 1;
    const int nblk = (n + 255) / 256;
    hipLaunchKernelGGL(k_btri, dim3((unsigned)((size_t)npoints * nblk)), dim3(256), 0, (hipStream_t)hip_stream, n, nblk, (const double *)b->dc, dprob, dX, dF);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// nlh_device_jacfcn: point q's n x n Jacobian, column-major, at dJ + q n n
int btri_launch_jac(void *ctx, void *hip_stream, int32_t npoints, const int32_t *dprob, int32_t n, const double *dX, int32_t m, double *dJ)
{
    (void)dprob;
    if (!ctx || m != n) return 1;
    const int nblk = (n + 255) / 256;
    for (int32_t q0 = 0; q0 < npoints; q0 += 65535) {
        const int32_t cnt = npoints - q0 < 65535 ? npoints - q0 : 65535;
        hipLaunchKernelGGL(k_btri_jac, dim3((unsigned)(n * nblk), (unsigned)cnt), dim3(256), 0, (hipStream_t)hip_stream, n, nblk,
                           dX + (size_t)q0 * n, dJ + (size_t)q0 * n * n);
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

void btri_host_fcn(void *ctx, int32_t n, const double *x, int32_t m, double *f)
{
    btri_host *b = (btri_host *)ctx;
    b->ncalls += 1;
    for (int32_t i = 0; i < m; ++i) f[i] = btri_row(n, x, i, b->c);
}

void btri_host_jac(void *ctx, int32_t n, const double *x, int32_t m, double *jac)
{
    btri_host *b = (btri_host *)ctx;
    b->njcalls += 1;
    for (int32_t j = 0; j < n; ++j)
        for (int32_t i = 0; i < m; ++i) jac[(size_t)j * m + i] = btri_jac(n, x, i, j);
}

// ---------------------------------------------------------------------------------------------------------------------
// Family 3: a scalar objective (the reference's fcnnvar, src/nonlin_multi_var.f90:17-44), for bfgs -- a chained
// Rosenbrock function with a per-problem target c:
//     f(x) = sum_{i = 0}^{n - 2} [ 10 (x_{i+1} - x_i^2)^2 + (c - x_i)^2 ],   terms added in ascending i by ONE thread.
// Launchers of the nlh_device_vecfcn / nlh_device_jacfcn types called with m = 1 (nlh_bfgs_solve_batch_device); the context
// is the btri family's (one c per problem).
__host__ __device__ static inline double crosen_f(int n, const double *x, double c)
{
    double s = 0.0;
    for (int i = 0; i + 1 < n; ++i) {
        const double d = x[i + 1] - x[i] * x[i], e = c - x[i];
        const double t = 10.0 * (d * d) + e * e;
        s = s + t;
    }
    return s;
}

__host__ __device__ static inline double crosen_g(int n, const double *x, int i, double c)
{
    double g = 0.0;
    if (i + 1 < n) {
        const double d = x[i + 1] - x[i] * x[i], e = c - x[i];
        g = (-40.0 * d) * x[i] - 2.0 * e;
    }
    if (i > 0) {
        const double dm = x[i] - x[i - 1] * x[i - 1];
        g = g + 20.0 * dm;
    }
    return g;
}

__global__ void __launch_bounds__(64)
k_crosen(int npoints, int n, const double *__restrict__ cs, const int32_t *__restrict__ dprob, const double *__restrict__ X, double *__restrict__ F)
{
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q < npoints) F[q] = crosen_f(n, X + (size_t)q * n, cs[dprob[q]]);
}

__global__ void __launch_bounds__(256)
k_crosen_grad(int n, int nblk, const double *__restrict__ cs, const int32_t *__restrict__ dprob, const double *__restrict__ X, double *__restrict__ G)
{
    const int q = blockIdx.x / nblk, rb = blockIdx.x - q * nblk;
    const int i = rb * 256 + threadIdx.x;
    if (i < n) G[(size_t)q * n + i] = crosen_g(n, X + (size_t)q * n, i, cs[dprob[q]]);
}

int crosen_launch(void *ctx, void *hip_stream, int32_t npoints, const int32_t *dprob, int32_t n, const double *dX, int32_t m, double *dF)
{
    const btri_dev *b = (const btri_dev *)ctx;
    if (!b || m != 1) return 1;
    hipLaunchKernelGGL(k_crosen, dim3((unsigned)((npoints + 63) / 64)), dim3(64), 0, (hipStream_t)hip_stream, (int)npoints, n, (const double *)b->dc, dprob, dX, dF);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int crosen_launch_grad(void *ctx, void *hip_stream, int32_t npoints, const int32_t *dprob, int32_t n, const double *dX, int32_t m, double *dG)
{
    const btri_dev *b = (const btri_dev *)ctx;
    if (!b || m != 1) return 1;
    const int nblk = (n + 255) / 256;
    hipLaunchKernelGGL(k_crosen_grad, dim3((unsigned)((size_t)npoints * nblk)), dim3(256), 0, (hipStream_t)hip_stream, n, nblk, (const double *)b->dc, dprob, dX, dG);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

double crosen_host_f(double c, int32_t n, const double *x) { return crosen_f(n, x, c); }

void crosen_host_grad(double c, int32_t n, const double *x, double *g)
{
    for (int32_t i = 0; i < n; ++i) g[i] = crosen_g(n, x, i, c);
}

}   // extern "C"
