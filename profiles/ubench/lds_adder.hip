// lds_adder.hip -- what the ordered-sum wave of the row-parallel pass pays per row: a dependent fp64 add per row (64 columns
// in the lanes) fed from LDS, as (A) one ds_read_b128 per row PAIR from a [pair][lane][2] layout (the shipped form),
// (B) one ds_read2_b64 per row pair from a [row][lane] layout (row stride 528 bytes), (C) two ds_read_b64, (D) no LDS at
// all (the chain alone).  One wave per workgroup, one workgroup; reads issued 12 pairs ahead as in the kernel.
// hipcc --offload-arch=gfx950 -O3 -o lds_adder lds_adder.hip ; ./lds_adder
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(64) k_add(int rounds, double *out, long long *clk)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 8192; i += 64) lds[i] = 1.0 + i * 1e-9;
    __syncthreads();
    double s = 0.0;
    const long long t0 = wall_clock64();
    for (int r = 0; r < rounds; ++r) {
        // 48 row pairs per round
        v2d w[12];
        const unsigned a128 = (unsigned)(size_t)(lds + 2 * lane), a64 = (unsigned)(size_t)(lds + lane);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int h = 0; h < 12; ++h) {
                const int pr = c * 12 + h;
                if (MODE == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[h]) : "v"(a128), "n"(pr * 1056) : "memory");
                if (MODE == 1) asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(w[h]) : "v"(a64 + (unsigned)(pr / 2) * 2112u), "n"((pr % 2) * 132), "n"((pr % 2) * 132 + 66) : "memory");
                if (MODE == 2) {
                    double x, y;
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x) : "v"(a64), "n"(pr * 1056) : "memory");
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(y) : "v"(a64), "n"(pr * 1056 + 528) : "memory");
                    w[h].x = x; w[h].y = y;
                }
                if (MODE == 3) { w[h].x = 1.0; w[h].y = 2.0; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int h = 0; h < 12; ++h) { asm volatile("" : "+v"(w[h])); s = s + w[h].x; s = s + w[h].y; }
        }
    }
    const long long t1 = wall_clock64();
    if (lane == 0) { clk[0] = t1 - t0; }
    out[lane] = s;
}

// the same with the reads of the next 12 pairs in flight under the adds of the current 12 (software pipelined)
template <int MODE>
__global__ void __launch_bounds__(64) k_add_pipe(int rounds, double *out, long long *clk)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 8192; i += 64) lds[i] = 1.0 + i * 1e-9;
    __syncthreads();
    double s = 0.0;
    const unsigned a128 = (unsigned)(size_t)(lds + 2 * lane), a64 = (unsigned)(size_t)(lds + lane);
    v2d w[2][12];
    auto rd = [&](v2d (&ww)[12], int c) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 12; ++h) {
            if (MODE == 0) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(ww[h]) : "v"(a128 + (unsigned)(c * 12 + h) * 1056u) : "memory");
            if (MODE == 1) asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:66" : "=v"(ww[h]) : "v"(a64 + (unsigned)(c * 12 + h) * 1056u) : "memory");
        }
    };
    const long long t0 = wall_clock64();
    rd(w[0], 0);
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            rd(w[(c + 1) & 1], (c + 1) & 3);
            asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
#pragma unroll
            for (int h = 0; h < 12; ++h) { asm volatile("" : "+v"(w[c & 1][h])); s = s + w[c & 1][h].x; s = s + w[c & 1][h].y; }
        }
    }
    const long long t1 = wall_clock64();
    if (lane == 0) { clk[0] = t1 - t0; }
    out[lane] = s;
}

int main()
{
    double *out; long long *clk;
    hipMalloc(&out, 64 * 8); hipMalloc(&clk, 64);
    const int rounds = 200;
    auto run = [&](const char *name, auto kern) {
        long long best = 1LL << 60;
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(kern, dim3(1), dim3(64), 65536, 0, rounds, out, clk);
            hipDeviceSynchronize();
            long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
            if (c < best) best = c;
        }
        printf("%-64s %7.2f ns per row (%5.1f clocks at 2.1 GHz)\n", name, best * 10.0 / (rounds * 96.0), best * 10.0 / (rounds * 96.0) * 2.1);
    };
    hipFuncSetAttribute((const void *)k_add<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run("(A) ds_read_b128 per row pair, wait for all, 24 adds", k_add<0>);
    run("(B) ds_read2_b64 per row pair (rows 528 B apart)", k_add<1>);
    run("(C) two ds_read_b64 per row pair", k_add<2>);
    run("(D) no LDS reads: the chain alone", k_add<3>);
    run("(A) pipelined: next 12 pairs in flight under the adds", k_add_pipe<0>);
    run("(B) pipelined", k_add_pipe<1>);
    return 0;
}
