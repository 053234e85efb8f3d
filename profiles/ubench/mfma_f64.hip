// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate on gfx950 (registers only).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k(double *out, double a, double b, int iters)
{
    v4d acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    double av = a + threadIdx.x, bv = b - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    for (int w : {1, 2, 4}) {
        const int nwg = 256 * w, iters = 4000;
        double *d; hipMalloc(&d, sizeof(double) * nwg * 256);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<nwg, 256>>>(d, 1.0, 2.0, 10);
        hipDeviceSynchronize();
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            k<<<nwg, 256>>>(d, 1.0, 2.0, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double nm = (double)nwg * 4 * iters * 32.0;            // wave-level MFMAs
            printf("wgs/cu=%d: %.2f ms  %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", w, ms,
                   nm * 2048.0 / (ms * 1e-3) * 1e-12, 1024.0 * 2.4e9 / (nm / (ms * 1e-3)));
        }
    }
    return 0;
}
