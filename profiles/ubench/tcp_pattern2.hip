// tcp_pattern2.hip -- the design questions behind a sector-per-lane-quad producer for the row-parallel pass:
// how many waves and how many 8-row groups in flight does one CU need to reach its rate with shape (B) (tcp_pattern.hip),
// and what do NV extra 16-byte loads per group (the reflector entries, four distinct addresses per wave, in the same
// in-order vector-memory queue) cost?  Matrix rows as in the solver: ld = 320 sectors (n = 256), a window = 64 of them.
// hipcc --offload-arch=gfx950 -O3 -o tcp_pattern2 tcp_pattern2.hip ; ./tcp_pattern2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NW, int AH, int PAT, int STORE, int NV>
__global__ void __launch_bounds__(NW * 64) k_pat(double *A, const double *V, int nblk, int ld, size_t tst, size_t vst, int nwin, double *out)
{
    extern __shared__ double pad[];
    const int wg = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int prob = wg / nwin, win = wg % nwin;
    double *Ap = A + (size_t)prob * tst;
    const double *Vp = V + (size_t)prob * vst * 8;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(Ap, 0, (int)((size_t)nblk * ld * 64), 0x00020000);
    const unsigned ldb = (unsigned)ld * 64u, wbase = (unsigned)win * 4096u;
    unsigned off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) off[q] = wbase + (PAT == 0 ? (unsigned)lane * 64u + 16u * q : (unsigned)q * 1024u + (unsigned)lane * 16u);
    double acc = 0.0;
    u32x4 v[AH][4];
    double2 rv[AH][NV > 0 ? NV : 1];
    auto load = [&](int g, int blk) {
#pragma unroll
        for (int q = 0; q < NV; ++q) rv[g][q] = *reinterpret_cast<const double2 *>(Vp + (size_t)q * vst + (size_t)blk * 8 + 2 * (lane & 3));
#pragma unroll
        for (int q = 0; q < 4; ++q) v[g][q] = __builtin_amdgcn_raw_buffer_load_b128(ra, off[q], (unsigned)blk * ldb, 0);
    };
    auto use = [&](int g, int blk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double x = __hiloint2double((int)v[g][q].y, (int)v[g][q].x), y = __hiloint2double((int)v[g][q].w, (int)v[g][q].z);
#pragma unroll
            for (int s = 0; s < NV; ++s) { x = x - rv[g][s].x * 1.0000001; y = y - rv[g][s].y * 0.9999999; }
            acc = acc + x; acc = acc + y;
            if (STORE) {
                u32x4 w; w.x = (unsigned)__double2loint(x); w.y = (unsigned)__double2hiint(x); w.z = (unsigned)__double2loint(y); w.w = (unsigned)__double2hiint(y);
                __builtin_amdgcn_raw_buffer_store_b128(w, ra, off[q], (unsigned)blk * ldb, 1);
            }
        }
    };
    // wave wv takes blocks wv, wv + NW, ...; AH - 1 groups in flight ahead of the one in use
    int b = wv;
#pragma unroll
    for (int i = 0; i < AH - 1; ++i) load(i, b + i * NW);
    for (; b < nblk; b += AH * NW) {
#pragma unroll
        for (int i = 0; i < AH; ++i) {
            load((i + AH - 1) % AH, b + (i + AH - 1) * NW);          // (blocks past the end: dropped by the range check)
            __builtin_amdgcn_sched_barrier(0);
            use(i, b + i * NW);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (acc == 123.456) out[wg] = acc + pad[0];
}

int main()
{
    const int m = 4096, ld = 320, nblk = m / 8, nwin = 4, maxprob = 64;
    const size_t tst = (size_t)(m + 160) * ld, vst = m + 128;
    double *A, *V, *out;
    hipMalloc(&A, sizeof(double) * tst * maxprob + (1 << 20));
    hipMalloc(&V, sizeof(double) * vst * 8 * maxprob + (1 << 20));
    hipMalloc(&out, sizeof(double) * 1024);
    hipMemset(A, 0, sizeof(double) * tst * maxprob);
    hipMemset(V, 0, sizeof(double) * vst * 8 * maxprob);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto kern, int threads, int nw) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(nw), dim3(threads), 96 * 1024, 0, A, V, nblk, ld, tst, vst, nwin, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        printf("%-40s %3d wgs: %7.2f us per pass (%5.2f TB/s)\n", name, nw, best * 1e3, 8.0 * m * 64 * nw / (best * 1e-3) / 1e12);
    };
#define RUN(NW, AH, PAT, ST, NV) for (int nw : {32, 88, 128, 188, 256}) run("waves " #NW " depth " #AH " shape " #PAT " store " #ST " vloads " #NV, k_pat<NW, AH, PAT, ST, NV>, NW * 64, nw)
    RUN(12, 4, 0, 0, 0); RUN(12, 4, 1, 0, 0); RUN(12, 3, 1, 0, 0); RUN(12, 2, 1, 0, 0);
    RUN(6, 4, 1, 0, 0); RUN(6, 6, 1, 0, 0); RUN(6, 8, 1, 0, 0);
    RUN(12, 4, 1, 0, 3); RUN(12, 4, 1, 0, 5); RUN(12, 3, 1, 0, 3); RUN(6, 6, 1, 0, 5);
    RUN(12, 4, 0, 1, 0); RUN(12, 4, 1, 1, 0); RUN(12, 3, 1, 1, 3); RUN(12, 4, 1, 1, 5); RUN(6, 6, 1, 1, 5);
    return 0;
}
