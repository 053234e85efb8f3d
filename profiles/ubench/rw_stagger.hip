// rw_stagger.hip -- does it pay to SPREAD the flush of the exact pass over the steps?  Same access pattern as rw_stream.hip
// (one wave per 64-column window of a row-blocked matrix, 16-byte accesses at a 64-byte lane stride).  A cycle of four
// passes over the same matrices is timed in two shapes:
//   (a) three read-only passes + one pass that rewrites everything in place   (today: one flushing pass per period)
//   (b) four passes that each rewrite ONE of the four windows of every problem (a quarter of the stores in every pass)
// Both move the same bytes.  If reads and writes simply add up, (a) == (b).
// hipcc --offload-arch=gfx950 -O3 -o rw_stagger rw_stagger.hip ; ./rw_stagger [nprob]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int G>
__global__ void __launch_bounds__(64) k_rw(double *__restrict__ A, int m, int ld, int nwin, size_t tst, double *__restrict__ out, int wmask, int rowsplit)
{
    const int b = blockIdx.x, p = b / nwin, win = b % nwin, lane = threadIdx.x;
    double *Ap = A + (size_t)p * tst;
    const int nblk = m / 8;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(Ap, 0, (int)((size_t)nblk * ld * 64), 0x00020000);
    const unsigned so = (unsigned)(win * 64 + lane) * 64u, ldb = (unsigned)ld * 64u;
    const bool wr = (wmask >> win) & 1;                      // wave-uniform
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
        // rowsplit > 0: only the row blocks of one quarter of the rows are rewritten (the stagger over the ROWS instead)
        const bool w2 = rowsplit > 0 ? (((blk * 4) / nblk) == rowsplit - 1) : wr;
        if (w2) {
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b128(v[g * 4 + q], ra, so + 16u * q, (unsigned)(blk + g) * ldb, 1);
        }
    };
    load(v0, 0);
    for (int blk = 0; blk < nblk; blk += 2 * G) {
        load(v1, blk + G);
        work(v0, blk);
        load(v0, blk + 2 * G);
        work(v1, blk + G);
    }
    if (acc == 123.456) out[b] = acc;
}

int main(int argc, char **argv)
{
    const int nprob = argc > 1 ? atoi(argv[1]) : 1024, m = 4096, ld = 320, nwin = 4;
    const size_t tst = (size_t)m * ld;
    double *A, *out;
    hipMalloc(&A, sizeof(double) * tst * nprob + (1 << 20));
    hipMalloc(&out, sizeof(double) * nprob * nwin);
    hipMemset(A, 0, sizeof(double) * tst * nprob);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto cycle = [&](const char *name, const int (&masks)[4], const int (&rs)[4]) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            for (int k = 0; k < 4; ++k)
                hipLaunchKernelGGL(k_rw<2>, dim3(nprob * nwin), dim3(64), 0, 0, A, m, ld, nwin, tst, out, masks[k], rs[k]);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        printf("%-66s %8.3f ms per cycle of four passes\n", name, best);
    };
    printf("nprob %d: each pass reads %.1f GB; a cycle rewrites the matrices once\n", nprob, 8.0 * m * 256.0 * nprob / 1e9);
    const int z[4] = {0, 0, 0, 0};
    { const int a[4] = {0, 0, 0, 0}; cycle("four read-only passes", a, z); }
    { const int a[4] = {0, 0, 0, 15}; cycle("(a) three read-only passes + one that rewrites everything", a, z); }
    { const int a[4] = {1, 2, 4, 8}; cycle("(b) each pass rewrites one window of four", a, z); }
    { const int a[4] = {0, 0, 0, 0}; const int r[4] = {1, 2, 3, 4}; cycle("(c) each pass rewrites one quarter of the rows", a, r); }
    { const int a[4] = {15, 15, 15, 15}; cycle("four passes that each rewrite everything", a, z); }
    return 0;
}
