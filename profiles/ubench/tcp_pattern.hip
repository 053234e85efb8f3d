// tcp_pattern.hip -- what ONE CU can pull through its vector memory path with the two ways a wave can read a 4 KB block
// (8 rows x 64 columns of the row-blocked working matrix: each column's 8 rows are one 64-byte sector):
//   (A) a lane per column: 16 bytes per lane at a 64-byte lane stride, four instructions per block -- every instruction
//       touches all 64 sectors (32 lines) of the block for a quarter of each  (the lane-per-column passes)
//   (B) a lane quad per column: lanes 4c .. 4c+3 read column c's whole sector, 16 columns per instruction -- every
//       instruction reads 1 KB contiguous (8 lines)
// One workgroup per CU (96 KB of LDS keeps a second one away), 12 waves each walking its own share of the rows, 4 groups of
// loads in flight per wave -- the shape of the row-parallel pass's producers.  Also the same with stores back (the flush).
// hipcc --offload-arch=gfx950 -O3 -o tcp_pattern tcp_pattern.hip ; ./tcp_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PAT, int STORE>
__global__ void __launch_bounds__(768) k_pat(double *A, int nblk, int ld, size_t tst, double *out)
{
    extern __shared__ double pad[];
    const int wg = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double *Ap = A + (size_t)wg * tst;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(Ap, 0, (int)((size_t)nblk * ld * 64), 0x00020000);
    const unsigned ldb = (unsigned)ld * 64u;
    unsigned off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) off[q] = PAT == 0 ? (unsigned)lane * 64u + 16u * q : (unsigned)q * 1024u + (unsigned)lane * 16u;
    double acc = 0.0;
    u32x4 v[4][4];
    auto load = [&](int g, int blk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[g][q] = __builtin_amdgcn_raw_buffer_load_b128(ra, off[q], (unsigned)blk * ldb, 0);
    };
    auto use = [&](int g, int blk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double x = __hiloint2double((int)v[g][q].y, (int)v[g][q].x), y = __hiloint2double((int)v[g][q].w, (int)v[g][q].z);
            acc = acc + x * 1.0000001; acc = acc + y;
            if (STORE) {
                u32x4 w = v[g][q]; w.x ^= 1u;
                __builtin_amdgcn_raw_buffer_store_b128(w, ra, off[q], (unsigned)blk * ldb, 1);
            }
        }
    };
    // wave wv takes blocks wv, wv + 12, ...
    int b = wv;
    load(0, b); load(1, b + 12); load(2, b + 24);
    for (; b < nblk; b += 48) {
        load(3, b + 36); use(0, b);
        load(0, b + 48); use(1, b + 12);
        load(1, b + 60); use(2, b + 24);
        load(2, b + 72); use(3, b + 36);
    }
    if (acc == 123.456) out[wg] = acc + pad[0];
}

int main()
{
    const int nwg = 256, m = 4096, ld = 64, nblk = m / 8;
    const size_t tst = (size_t)m * ld + 4096;
    double *A, *out;
    hipMalloc(&A, sizeof(double) * tst * nwg + (1 << 20));
    hipMalloc(&out, sizeof(double) * nwg);
    hipMemset(A, 0, sizeof(double) * tst * nwg);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto kern, int nw) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(nw), dim3(768), 96 * 1024, 0, A, nblk, ld, tst, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        const double bytes = 8.0 * m * 64;
        printf("%-58s %3d wgs: %7.2f us per 4096 x 64 window = %5.1f bytes/clk/CU at 2.1 GHz\n", name, nw, best * 1e3, bytes / (best * 1e-3 * 2.1e9));
    };
    for (int nw : {256, 32}) {
        run("(A) lane per column, 16 B at a 64 B stride, read", k_pat<0, 0>, nw);
        run("(B) 1 KB contiguous per instruction, read", k_pat<1, 0>, nw);
        run("(A) read + store back", k_pat<0, 1>, nw);
        run("(B) read + store back", k_pat<1, 1>, nw);
    }
    return 0;
}
