// chain_waves.hip -- how fast the serial fp64 add chain of the exact kernels runs (the "down the lanes" form: 64 terms per
// lane, the running sum handed on with a DPP wave shift) alone on a CU, with two workgroups per CU whose chain waves are
// the SAME wave index, and with DIFFERENT wave indices; plus where the waves of a workgroup sit (HW_ID register).
// hipcc --offload-arch=gfx950 -O3 -o chain_waves chain_waves.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ double shr1(double t)
{
    int lo = __double2loint(t), hi = __double2hiint(t);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <int EL>
__global__ void __launch_bounds__(256) k_chain(int reps, int mode, double *out, unsigned *hwid)
{
    __shared__ double pad[9000];                       // 72 KB: at most two workgroups per CU
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned b = blockIdx.x;
    const int cw = mode == 0 ? 0 : (int)((b + (b >> 3) + (b >> 8)) & 3u);
    if (lane == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        hwid[b * 4 + wid] = id;
    }
    pad[tid] = tid;
    __syncthreads();
    if (wid != cw) return;
    double d[EL];
#pragma unroll
    for (int u = 0; u < EL; ++u) d[u] = 1e-3 * (lane + u) + pad[(lane + u) & 255] * 1e-9;
    double t = 0.0;
    for (int r = 0; r < reps; ++r) {
#pragma unroll 1
        for (int l = 0; l < 64; ++l) {
            if (l > 0) t = shr1(t);
#pragma unroll
            for (int u = 0; u < EL; ++u) t = t + d[u];
        }
    }
    if (t == 123.0) out[b] = t;
}

// The same dependent-add chain with only `nact` lanes of the wave enabled (EXEC): does the SIMD skip the 16-lane passes
// of a wave64 instruction whose lanes are all off?
__global__ void __launch_bounds__(64) k_chain_exec(int n, int nact, double *out)
{
    const int lane = threadIdx.x;
    double t = lane * 1e-3;
    const double d = 1e-3 + lane * 1e-9;
    if (lane < nact) {
#pragma unroll 1
        for (int i = 0; i < n; i += 64) {
#pragma unroll
            for (int u = 0; u < 64; ++u) t = t + d;
        }
    }
    if (t == 123.0) out[lane] = t;
}

int main()
{
    double *out; unsigned *hwid;
    (void)hipMalloc(&out, 8 * 4096); hipMalloc(&hwid, 4 * 4 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 50;
    auto run = [&](const char *name, auto kern, int nwg, int mode, int el) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, 0, reps, mode, out, hwid);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-52s %4d wgs: %7.2f us per %d-add chain = %.2f ns per add\n", name, nwg, 1e3 * ms / reps, 64 * el, 1e6 * ms / reps / (64 * el));
        }
    };
    run("EL 64, one workgroup", k_chain<64>, 1, 0, 64);
    run("EL 64, one per CU", k_chain<64>, 256, 0, 64);
    run("EL 64, two per CU, chain wave 0 in both", k_chain<64>, 512, 0, 64);
    run("EL 64, two per CU, chain waves spread", k_chain<64>, 512, 1, 64);
    run("EL 32, one per CU", k_chain<32>, 256, 0, 32);
    run("EL 32, two per CU, chain wave 0 in both", k_chain<32>, 512, 0, 32);
    run("EL 32, two per CU, chain waves spread", k_chain<32>, 512, 1, 32);
    for (int nact : {64, 32, 16, 1}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_chain_exec, dim3(1), dim3(64), 0, 0, 409600, nact, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("dependent adds, %2d lanes enabled: %.2f ns per add\n", nact, 1e6 * ms / 409600);
        }
    }
    // where the waves of the first workgroups ran: HW_ID bits: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx9 layout)
    unsigned h[4 * 520];
    hipMemcpy(h, hwid, sizeof h, hipMemcpyDeviceToHost);
    for (int b : {0, 1, 8, 256, 257, 264}) {
        printf("wg %3d:", b);
        for (int w = 0; w < 4; ++w) printf("  wave %d -> simd %u cu %u se %u (raw %08x)", w, (h[b * 4 + w] >> 4) & 3, (h[b * 4 + w] >> 8) & 15, (h[b * 4 + w] >> 13) & 7, h[b * 4 + w]);
        printf("\n");
    }
    return 0;
}
