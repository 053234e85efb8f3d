// rw_stream.hip -- what the memory system gives a streaming kernel with the exact pass's access pattern (one wave per
// 64-column window of a row-blocked matrix, 16-byte accesses at a 64-byte lane stride, 4 KB contiguous per wave and
// 8-row block) when it (0) only reads, (1) only writes, (2) reads and writes in place, (3) reads one buffer and writes
// another, (4) reads and writes in place one tile LATER (the stores trail the loads by a 64-row tile).
// hipcc --offload-arch=gfx950 -O3 -o rw_stream rw_stream.hip ; ./rw_stream [nprob] [waves_per_simd_limit via LDS pad KB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int G, int XS = 0>     // G = 8-row blocks per load group; XS: stores line-contiguous through LDS
__global__ void __launch_bounds__(64) k_rw(double *__restrict__ A, double *__restrict__ B, int m, int ld, int nwin, size_t tst,
                                           double *__restrict__ out, int ldspad)
{
    extern __shared__ double pad[];
    __shared__ __attribute__((aligned(16))) double xst[64 * 8];
    const int b = blockIdx.x, p = b / nwin, win = b % nwin, lane = threadIdx.x;
    double *Ap = A + (size_t)p * tst, *Bp = (MODE == 3 ? B : A) + (size_t)p * tst;
    const int nblk = m / 8;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(Ap, 0, (int)((size_t)nblk * ld * 64), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(Bp, 0, (int)((size_t)nblk * ld * 64), 0x00020000);
    const unsigned so = (unsigned)(win * 64 + lane) * 64u, ldb = (unsigned)ld * 64u;
    double acc = 0.0;
    u32x4 v0[4 * G], v1[4 * G];
    auto load = [&](u32x4 (&v)[4 * G], int blk) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[g * 4 + q] = __builtin_amdgcn_raw_buffer_load_b128(ra, so + 16u * q, (unsigned)(blk + g) * ldb, 0);
    };
    auto work = [&](u32x4 (&v)[4 * G], int blk) {
#pragma unroll
        for (int i = 0; i < 4 * G; ++i) {
            double x = __hiloint2double((int)v[i].y, (int)v[i].x), y = __hiloint2double((int)v[i].w, (int)v[i].z);
            acc = acc + x; acc = acc + y;
            x = x * 1.0000001; y = y * 0.9999999;
            v[i].x = (unsigned)__double2loint(x); v[i].y = (unsigned)__double2hiint(x);
            v[i].z = (unsigned)__double2loint(y); v[i].w = (unsigned)__double2hiint(y);
        }
        if (MODE >= 1 && !XS) {
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b128(v[g * 4 + q], rb, so + 16u * q, (unsigned)(blk + g) * ldb, 1);
        }
        if (MODE >= 1 && XS) {
            u32x4 *xs = reinterpret_cast<u32x4 *>(xst);
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int q = 0; q < 4; ++q) xs[lane * 4 + q] = v[g * 4 + q];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x4 w = xs[q * 64 + lane];
                    __builtin_amdgcn_raw_buffer_store_b128(w, rb, (unsigned)(win * 64) * 64u + q * 1024u + lane * 16u, (unsigned)(blk + g) * ldb, 1);
                }
            }
        }
    };
    if (MODE == 1) {
        u32x4 c; c.x = lane; c.y = 0x3ff00000u; c.z = b; c.w = 0x3ff00000u;
        for (int blk = 0; blk < nblk; blk += G)
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b128(c, rb, so + 16u * q, (unsigned)(blk + g) * ldb, 1);
    } else {
        load(v0, 0);
        for (int blk = 0; blk < nblk; blk += 2 * G) {
            load(v1, blk + G);
            work(v0, blk);
            load(v0, blk + 2 * G);          // past the end: out of range, returns zero
            work(v1, blk + G);
        }
    }
    if (acc == 123.456) out[b] = acc + pad[ldspad ? 1 : 0];
}

int main(int argc, char **argv)
{
    const int nprob = argc > 1 ? atoi(argv[1]) : 1024, m = 4096, ld = 320, nwin = 4;
    const int ldskb = argc > 2 ? atoi(argv[2]) : 0;           // dynamic LDS per workgroup: limits workgroups per CU
    const size_t tst = (size_t)m * ld;
    double *A, *B, *out;
    hipMalloc(&A, sizeof(double) * tst * nprob + (1 << 20));
    hipMalloc(&B, sizeof(double) * tst * nprob + (1 << 20));
    hipMalloc(&out, sizeof(double) * nprob * nwin);
    hipMemset(A, 0, sizeof(double) * tst * nprob);
    hipMemset(B, 0, sizeof(double) * tst * nprob);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double gb = 8.0 * m * 256.0 * nprob / 1e9;          // bytes one direction
    auto run = [&](const char *name, auto kern, double dirs) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(nprob * nwin), dim3(64), ldskb * 1024, 0, A, B, m, ld, nwin, tst, out, ldskb);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%-34s %8.3f ms  %7.1f GB/s (%.0f GB moved)\n", name, ms, dirs * gb / (ms * 1e-3), dirs * gb);
        }
    };
    printf("nprob %d, %d KB LDS per workgroup\n", nprob, ldskb);
    run("read only, 16 rows/group", k_rw<0, 2>, 1.0);
    run("write only", k_rw<1, 2>, 1.0);
    run("read+write in place, 16 rows/group", k_rw<2, 2>, 2.0);
    run("read+write in place, 8 rows/group", k_rw<2, 1>, 2.0);
    run("read+write in place, 32 rows/group", k_rw<2, 4>, 2.0);
    run("read A write B, 16 rows/group", k_rw<3, 2>, 2.0);
    run("in place, contiguous stores, 16 rows", (k_rw<2, 2, 1>), 2.0);
    run("in place, contiguous stores, 32 rows", (k_rw<2, 4, 1>), 2.0);
    run("A->B, contiguous stores, 16 rows", (k_rw<3, 2, 1>), 2.0);
    return 0;
}
