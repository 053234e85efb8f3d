// dp_issue.hip -- how fast does ONE wave issue fp64 VALU operations on gfx950?  K independent dependent-add chains
// per lane, N steps each; 1 / 2 / 4 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 dp_issue.hip -o dp_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K, bool MUL>
__global__ void __launch_bounds__(1024) k_chain(double *out, int n, double x)
{
    double a[K];
#pragma unroll
    for (int k = 0; k < K; ++k) a[k] = threadIdx.x * 1e-9 + k;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < K; ++k) a[k] = MUL ? a[k] * x : a[k] + x;
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one chain, unrolled 64x: the latency of a dependent fp64 add without loop overhead
__global__ void __launch_bounds__(64) k_dep(double *out, int n, double x)
{
    double a = threadIdx.x * 1e-9;
    for (int i = 0; i < n; i += 64) {
#pragma unroll
        for (int u = 0; u < 64; ++u) a = a + x;
    }
    out[threadIdx.x] = a;
}
// the same with an LDS operand per add (the adder of k_qrx_pass_rp / the NORM2 serial phase)
__global__ void __launch_bounds__(64) k_dep_lds(double *out, int n, double x)
{
    __shared__ double buf[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) buf[i] = x;
    __syncthreads();
    double a = threadIdx.x * 1e-9;
    for (int i = 0; i < n; i += 64) {
        double w[64];
#pragma unroll
        for (int u = 0; u < 64; ++u) w[u] = buf[u * 64 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < 64; ++u) a = a + w[u];
    }
    out[threadIdx.x] = a;
}

template <int K, bool MUL>
static void run(int threads, int blocks, double *d)
{
    const int n = 1 << 20;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_chain<K, MUL>), dim3(blocks), dim3(threads), 0, 0, d, 1024, 1.0000001);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_chain<K, MUL>), dim3(blocks), dim3(threads), 0, 0, d, n, 1.0000001);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s chains/lane %2d  waves/WG %2d  WGs %4d : %7.3f ns per chain step, %7.3f ns per op per wave\n", MUL ? "mul" : "add", K,
           threads / 64, blocks, ms * 1e6 / n, ms * 1e6 / n / K);
}

int main()
{
    double *d; hipMalloc(&d, sizeof(double) * 1024 * 1024);
    for (int threads : {64, 256, 512, 1024}) {
        run<1, false>(threads, 1, d); run<2, false>(threads, 1, d); run<4, false>(threads, 1, d); run<8, false>(threads, 1, d);
        run<16, false>(threads, 1, d);
    }
    {
        const int n = 1 << 22;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int which = 0; which < 2; ++which) {
            if (which == 0) hipLaunchKernelGGL(k_dep, dim3(1), dim3(64), 0, 0, d, 64, 1.0000001); else hipLaunchKernelGGL(k_dep_lds, dim3(1), dim3(64), 0, 0, d, 64, 1.0000001);
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_dep, dim3(1), dim3(64), 0, 0, d, n, 1.0000001); else hipLaunchKernelGGL(k_dep_lds, dim3(1), dim3(64), 0, 0, d, n, 1.0000001);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("dependent add, unrolled 64x%s: %7.3f ns per add\n", which ? ", LDS operand" : "", ms * 1e6 / n);
        }
    }
    run<1, true>(64, 1, d); run<4, true>(64, 1, d); run<8, true>(64, 1, d);
    run<4, false>(256, 256, d); run<4, false>(512, 256, d); run<8, false>(1024, 256, d);
    return 0;
}
