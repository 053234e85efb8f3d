// rpw_lds_skel.hip -- the LDS side of the wide row-parallel pass without its memory side: sixteen waves per workgroup,
// wave 0 the adder (per round 48 row pairs x 64 columns read from LDS, 96 dependent adds), twelve producers (per round
// four product writes of 1 KB each and NV 16-byte reads of staged reflector entries), one LDS-only barrier per round.
// What does a round cost, and which part of the LDS traffic is it?
//   MODE 0: everything      1: no reflector reads     2: no product writes     3: adder adds without reading
//   4: products as ds_write2_b64 / ds_read2_b64 in a [row][column] layout (528-byte rows) instead of 16 bytes per lane
//   5: adder alone (producers only join the barrier)
// hipcc --offload-arch=gfx950 -O3 -o rpw_lds_skel rpw_lds_skel.hip ; ./rpw_lds_skel
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int MODE, int NV>
__global__ void __launch_bounds__(1024) k_skel(int rounds, double *out, long long *clk)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];          // [2][48][132] products, then 4 KB of entries
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double *pb0 = lds, *pb1 = lds + 48 * 132, *vs = lds + 2 * 48 * 132;
    for (int i = threadIdx.x; i < 2 * 48 * 132 + 512; i += 1024) lds[i] = 1.0 + i * 1e-9;
    __syncthreads();
    if (wv > 12) return;
    const long long t0 = wall_clock64(), c0 = clock64();
    if (wv == 0) {
        __builtin_amdgcn_s_setprio(3);
        double s = 0.0;
        lds_barrier();
        for (int t = 0; t <= rounds; ++t) {
            if (t >= 1) {
                const double *half = ((t - 1) & 1) ? pb1 : pb0;
                if (MODE == 3) {
#pragma unroll
                    for (int i = 0; i < 96; ++i) { asm volatile("" : "+v"(s)); s = s + 1.25; }
                } else if (MODE == 4) {
                    const unsigned addr = (unsigned)(size_t)(half + lane);
                    v2d w[4][4];
#define RD4(c) asm volatile("ds_read2_b64 %0, %4 offset0:0 offset1:66\n\tds_read2_b64 %1, %5 offset0:0 offset1:66\n\tds_read2_b64 %2, %6 offset0:0 offset1:66\n\tds_read2_b64 %3, %7 offset0:0 offset1:66" \
                        : "=&v"(w[(c) & 3][0]), "=&v"(w[(c) & 3][1]), "=&v"(w[(c) & 3][2]), "=&v"(w[(c) & 3][3])                   \
                        : "v"(addr + (c) * 4224u), "v"(addr + (c) * 4224u + 1056u), "v"(addr + (c) * 4224u + 2112u), "v"(addr + (c) * 4224u + 3168u) : "memory")
#define WT4(c, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(w[(c) & 3][0]), "+v"(w[(c) & 3][1]), "+v"(w[(c) & 3][2]), "+v"(w[(c) & 3][3]))
                    RD4(0); RD4(1); RD4(2);
#pragma unroll
                    for (int c = 0; c < 12; ++c) {
                        if (c + 2 < 12) { WT4(c, 8); } else if (c + 1 < 12) { WT4(c, 4); } else { WT4(c, 0); }
                        if (c + 3 < 12) { RD4(c + 3); }
#pragma unroll
                        for (int h = 0; h < 4; ++h) { s = s + w[c & 3][h].x; s = s + w[c & 3][h].y; }
                    }
#undef RD4
#undef WT4
                } else {
                    const unsigned addr = (unsigned)(size_t)(half + 2 * lane);
                    v2d w[4][4];
#define RD4(c) asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7" \
                        : "=&v"(w[(c) & 3][0]), "=&v"(w[(c) & 3][1]), "=&v"(w[(c) & 3][2]), "=&v"(w[(c) & 3][3])                   \
                        : "v"(addr + (c) * 4224u), "v"(addr + (c) * 4224u + 1056u), "v"(addr + (c) * 4224u + 2112u), "v"(addr + (c) * 4224u + 3168u) : "memory")
#define WT4(c, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(w[(c) & 3][0]), "+v"(w[(c) & 3][1]), "+v"(w[(c) & 3][2]), "+v"(w[(c) & 3][3]))
                    RD4(0); RD4(1); RD4(2);
#pragma unroll
                    for (int c = 0; c < 12; ++c) {
                        if (c + 2 < 12) { WT4(c, 8); } else if (c + 1 < 12) { WT4(c, 4); } else { WT4(c, 0); }
                        if (c + 3 < 12) { RD4(c + 3); }
#pragma unroll
                        for (int h = 0; h < 4; ++h) { s = s + w[c & 3][h].x; s = s + w[c & 3][h].y; }
                    }
#undef RD4
#undef WT4
                }
            }
            lds_barrier();
        }
        if (lane == 0 && blockIdx.x == 0) { clk[0] = wall_clock64() - t0; clk[1] = clock64() - c0; }
        out[blockIdx.x * 64 + lane] = s;
        return;
    }
    const int pw = wv - 1, c4 = lane >> 2, rq = lane & 3;
    double acc = 1.0 + lane;
    lds_barrier();
    for (int t = 0; t <= rounds; ++t) {
        if (t < rounds && MODE != 5) {
            v2d vq[NV > 0 ? NV : 1];
            if (MODE != 1) {
#pragma unroll
                for (int q = 0; q < NV; ++q) vq[q] = *reinterpret_cast<const v2d *>(vs + q * 96 + pw * 8 + 2 * rq);
            } else {
#pragma unroll
                for (int q = 0; q < NV; ++q) { vq[q].x = 1.0 + q; vq[q].y = 0.5; }
            }
            double *buf = (t & 1) ? pb1 : pb0;
#pragma unroll
            for (int q2 = 0; q2 < 4; ++q2) {
                v2d ww; ww.x = acc; ww.y = acc * 0.5;
#pragma unroll
                for (int q = 0; q < NV; ++q) { ww.x = ww.x - vq[q].x * 1.0000001; ww.y = ww.y - vq[q].y * 0.9999999; }
                acc = acc + ww.x;
                if (MODE == 2) { asm volatile("" :: "v"(ww)); }
                else if (MODE == 4) {
                    // [row][column]: rows 2rq, 2rq + 1 of the producer's group, 528-byte rows
                    const unsigned a = (unsigned)(size_t)(buf + (size_t)(pw * 8 + 2 * rq) * 66 + 16 * q2 + c4);
                    asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:66" :: "v"(a), "v"(ww.x), "v"(ww.y) : "memory");
                } else
                    *reinterpret_cast<v2d *>(buf + (size_t)(pw * 4 + rq) * 132 + 2 * (16 * q2 + c4)) = ww;
            }
        }
        lds_barrier();
    }
    out[4096 + blockIdx.x * 1024 + threadIdx.x] = acc;
}

int main()
{
    double *out; long long *clk;
    hipMalloc(&out, sizeof(double) * (4096 + 1024 * 256)); hipMalloc(&clk, 64);
    const int rounds = 42, lds = (2 * 48 * 132 + 512) * 8;
    auto run = [&](const char *name, auto kern) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        long long best = 1LL << 60, cyc = 0;
        for (int rep = 0; rep < 4; ++rep) {
            hipLaunchKernelGGL(kern, dim3(128), dim3(1024), lds, 0, rounds, out, clk);
            hipDeviceSynchronize();
            long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
            if (rep > 0 && c[0] < best) { best = c[0]; cyc = c[1]; }
        }
        printf("%-78s %6.2f us per 4032-row pass = %5.0f ns per round; s_memtime ticks / 10 ns: %.2f\n", name, best / 100.0, best * 10.0 / rounds, (double)cyc / best);
    };
    run("0: everything, 2 entry reads per group", k_skel<0, 2>);
    run("0: everything, 5 entry reads per group", k_skel<0, 5>);
    run("1: no entry reads (2 slots of arithmetic)", k_skel<1, 2>);
    run("2: no product writes", k_skel<2, 2>);
    run("3: adder adds without reading", k_skel<3, 2>);
    run("4: products through ds_write2_b64 / ds_read2_b64, 2 entry reads", k_skel<4, 2>);
    run("5: adder alone", k_skel<5, 2>);
    return 0;
}
