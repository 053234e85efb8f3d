// Microbenchmark: sustained v_add_f64 / v_fma_f64 / v_mul_f64 issue rate on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, double c, int iters)
{
    double acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = threadIdx.x + j;
    double one = 1.0;
    asm volatile("" : "+s"(one));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (MODE == 0) acc[j] = acc[j] + c;
                else if (MODE == 1) acc[j] = __builtin_fma(c, one, acc[j]);
                else acc[j] = acc[j] * c;
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, int wgs_per_cu)
{
    const int nwg = 256 * wgs_per_cu, iters = 20000;
    double *d; hipMalloc(&d, sizeof(double) * nwg * 256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<nwg, 256>>>(d, 1.0000001, 100);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<nwg, 256>>>(d, 1.0000001, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = (double)nwg * 256 * iters * 128.0;
    printf("%s wgs/cu=%d: %.2f ms  %.2f Tops/s  (cycles per wave-instr per SIMD at 2.4GHz: %.2f)\n", name, wgs_per_cu, ms,
           ops / ms * 1e-9, 1024.0 * 2.4e9 * 64 / (ops / (ms * 1e-3)));
    hipFree(d);
}
int main()
{
    for (int w : {1, 2, 4, 8}) { run<0>("add", w); run<1>("fma", w); run<2>("mul", w); }
    return 0;
}
