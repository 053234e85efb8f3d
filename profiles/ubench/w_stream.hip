// w_stream.hip -- write-only streaming: which store shape / cache policy the memory system takes fastest.
// Every wave writes its 64-column window of a row-blocked matrix (4 KB contiguous per 8-row block), block after block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// SHAPE 0: b128, lane stride 64 B (a sector per lane: the flush as it is)   1: b128, lane stride 16 B (1 KB contiguous)
//       2: b64, lane stride 8 B (512 B contiguous)                         3: b32, lane stride 4 B (256 B contiguous)
template <int SHAPE, int AUX>
__global__ void __launch_bounds__(64) k_w(double *__restrict__ A, int m, int ld, int nwin, size_t tst)
{
    const int b = blockIdx.x, p = b / nwin, win = b % nwin, lane = threadIdx.x;
    double *Ap = A + (size_t)p * tst;
    const int nblk = m / 8;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(Ap, 0, (int)((size_t)nblk * ld * 64), 0x00020000);
    const unsigned base = (unsigned)(win * 64) * 64u, ldb = (unsigned)ld * 64u;
    u32x4 c; c.x = lane; c.y = 0x3ff00000u; c.z = b; c.w = 0x3ff00000u;
    u32x2 c2; c2.x = lane; c2.y = 0x3ff00000u;
    for (int blk = 0; blk < nblk; ++blk) {
        const unsigned so = (unsigned)blk * ldb;
        if (SHAPE == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b128(c, ra, base + lane * 64u + 16u * q, so, AUX);
        } else if (SHAPE == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b128(c, ra, base + q * 1024u + lane * 16u, so, AUX);
        } else if (SHAPE == 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) __builtin_amdgcn_raw_buffer_store_b64(c2, ra, base + q * 512u + lane * 8u, so, AUX);
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) __builtin_amdgcn_raw_buffer_store_b32(c.x, ra, base + q * 256u + lane * 4u, so, AUX);
        }
    }
}

int main(int argc, char **argv)
{
    const int nprob = argc > 1 ? atoi(argv[1]) : 1024, m = 4096, ld = 320, nwin = 4;
    const size_t tst = (size_t)m * ld;
    double *A;
    hipMalloc(&A, sizeof(double) * tst * nprob + (1 << 20));
    hipMemset(A, 0, sizeof(double) * tst * nprob);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double gb = 8.0 * m * 256.0 * nprob / 1e9;
    auto run = [&](const char *name, auto kern) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(nprob * nwin), dim3(64), 0, 0, A, m, ld, nwin, tst);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, gb / (ms * 1e-3));
        }
    };
    printf("write only, nprob %d (%.1f GB)\n", nprob, gb);
    run("b128 sector per lane, default", k_w<0, 0>);
    run("b128 sector per lane, sc0", k_w<0, 1>);
    run("b128 sector per lane, nt", k_w<0, 2>);
    run("b128 contiguous, default", k_w<1, 0>);
    run("b128 contiguous, sc0", k_w<1, 1>);
    run("b128 contiguous, nt", k_w<1, 2>);
    run("b128 contiguous, sc1", k_w<1, 16>);
    run("b128 contiguous, sc0 sc1", k_w<1, 17>);
    run("b64 contiguous, default", k_w<2, 0>);
    run("b64 contiguous, nt", k_w<2, 2>);
    run("b32 contiguous, default", k_w<3, 0>);
    run("b32 contiguous, sc0", k_w<3, 1>);
    run("b32 contiguous, nt", k_w<3, 2>);
    // torch-style flat memset for reference
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipMemsetAsync(A, 1, sizeof(double) * tst * nprob, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("%-44s %8.3f ms  %7.1f GB/s\n", "hipMemsetAsync (whole buffer incl. padding)", ms, sizeof(double) * tst * nprob / 1e9 / (ms * 1e-3));
    }
    return 0;
}
