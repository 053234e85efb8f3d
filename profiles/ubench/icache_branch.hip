// What does a TAKEN BRANCH INTO CODE THAT HAS NEVER RUN cost?  (gfx950)
// One wave runs N chunks of C dependent-free 8-byte VALU instructions; every chunk ends in an s_branch over a dead block of
// D instructions, so each chunk starts at an address the sequential prefetch has not reached.  Pass 1 is cold (the
// instruction cache is invalidated at kernel start), pass 2 warm.  Cycles are shader clocks (clock64()).
//   hipcc --offload-arch=gfx950 -O2 -o icache_branch icache_branch.hip && ./icache_branch
#include <hip/hip_runtime.h>
#include <cstdio>

template <int N, int C, int D>
__global__ void k_chunks(long long *out, double seed)
{
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
    long long t[3];
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        t[pass] = clock64();
        asm volatile(".rept %4\n"
                     "  .rept %5\n  v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n  .endr\n"
                     "  s_branch 1f\n"
                     "  .rept %6\n  v_fma_f64 %0, %0, %0, %0\n  .endr\n"
                     "1:\n"
                     ".endr\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                     : "n"(N), "n"(C / 4), "n"(D));
    }
    t[2] = clock64();
    if (threadIdx.x == 0) {
        out[0] = t[1] - t[0];
        out[1] = t[2] - t[1];
        out[2] = (long long)(a0 + a1 + a2 + a3);
    }
}

template <int N, int C, int D>
void run(long long *d)
{
    long long h[3];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k_chunks<N, C, D>), dim3(1), dim3(64), 0, 0, d, 0.0);
        (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    }
    const double base = 4.0 * N * C;                            // the FMAs alone: 4 cycles each
    printf("%4d chunks of %3d instr, %4d dead instr between (%5.1f KiB spanned): cold %8lld cycles = %6.0f per branch above the FMAs;  warm %8lld = %6.0f\n",
           N, C, D, N * (C + D + 1) * 8 / 1024.0, h[0], (h[0] - base) / N, h[1], (h[1] - base) / N);
}

int main()
{
    long long *d;
    (void)hipMalloc(&d, 64);
    run<64, 32, 0>(d);       // branch to the next instruction: no skip
    run<64, 32, 8>(d);       // one line skipped
    run<64, 32, 32>(d);
    run<64, 32, 96>(d);
    run<32, 32, 224>(d);
    run<16, 32, 480>(d);
    run<128, 8, 24>(d);
    return 0;
}
