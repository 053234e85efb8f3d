// How fast does a wave run straight-line code it has never executed?  (gfx950)
// One wave executes a block of K independent 8-byte VALU instructions twice; pass 1 fetches every line from L2 / HBM
// (the instruction cache is invalidated at kernel start), pass 2 finds it in the instruction cache when the block fits.
// Prints cycles per instruction and ns per KiB of code for both passes.
//   hipcc --offload-arch=gfx950 -O2 -o icache_cold icache_cold.hip && ./icache_cold
#include <hip/hip_runtime.h>
#include <cstdio>

template <int K>
__global__ void k_block(long long *out, double seed)
{
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    long long t[3];
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        t[pass] = clock64();
        asm volatile(".rept %8\n"
                     "v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n"
                     "v_fma_f64 %4, %4, %4, %4\n v_fma_f64 %5, %5, %5, %5\n v_fma_f64 %6, %6, %6, %6\n v_fma_f64 %7, %7, %7, %7\n"
                     ".endr\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "n"(K / 8));
    }
    t[2] = clock64();
    if (threadIdx.x == 0) {
        out[0] = t[1] - t[0];
        out[1] = t[2] - t[1];
        out[2] = (long long)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
    }
}

template <int K>
void run(long long *d)
{
    long long h[3];
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_block<K>, dim3(1), dim3(64), 0, 0, d, 0.0);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    }
    // clock64() counts at 100 MHz on this part: report ns
    const double ns1 = h[0] * 10.0, ns2 = h[1] * 10.0, kib = K * 8 / 1024.0;
    printf("K = %6d instr (%6.1f KiB): pass 1 %9.0f ns = %6.2f ns/instr, %7.1f ns/KiB;  pass 2 %9.0f ns = %6.2f ns/instr\n", K, kib,
           ns1, ns1 / K, ns1 / kib, ns2, ns2 / K);
}

int main()
{
    long long *d;
    hipMalloc(&d, 64);
    run<256>(d); run<1024>(d); run<4096>(d); run<6144>(d); run<8192>(d); run<12288>(d);
    return 0;
}
