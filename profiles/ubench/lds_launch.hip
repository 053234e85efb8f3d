// lds_launch.hip -- does the start-up time of a small, latency-bound launch depend on its LDS allocation?  A kernel of
// 256 workgroups x 256 threads whose threads do one dependent global load and store and touch one LDS word, launched
// back to back on one stream with 0 .. 128 KB of dynamic LDS per workgroup; the time per launch is the quantity the
// per-step kernels of a lone factorisation pay 2 n times.
// hipcc --offload-arch=gfx950 -O3 -o lds_launch lds_launch.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256) k_touch(const int *__restrict__ idx, double *__restrict__ out, int nwords)
{
    extern __shared__ double lds[];
    const int t = threadIdx.x;
    if (nwords > 0) lds[t % nwords] = t;
    __syncthreads();
    const int k = idx[blockIdx.x];                     // a dependent load, as the step kernels start with
    out[(size_t)k * 256 + t] = (nwords > 0 ? lds[(t + 1) % nwords] : 0.0) + k;
}

int main()
{
    const int nwg = 256, reps = 200;
    int *idx; double *out;
    (void)hipMalloc(&idx, sizeof(int) * nwg); (void)hipMalloc(&out, sizeof(double) * nwg * 256);
    int h[nwg]; for (int i = 0; i < nwg; ++i) h[i] = (i * 7) % nwg;
    (void)hipMemcpy(idx, h, sizeof h, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute((const void *)k_touch, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int kb : {0, 1, 8, 16, 32, 48, 64, 96, 128, 159}) {
        for (int pass = 0; pass < 2; ++pass) {
            (void)hipEventRecord(e0);
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_touch, dim3(nwg), dim3(256), (size_t)kb * 1024, 0, idx, out, kb * 128);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (pass) printf("%3d KB of LDS per workgroup: %.2f us per launch\n", kb, 1e3 * ms / reps);
        }
    }
    return 0;
}
