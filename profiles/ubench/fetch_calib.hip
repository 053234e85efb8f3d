// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE for the access pattern of k_qrx_pass:
// one wave reads 512 contiguous, 512-byte-aligned bytes per instruction (8 bytes per lane, buffer_load_dwordx2 through a
// descriptor with a scalar row offset), walking down rows of a row-major matrix far larger than every cache.
// A second kernel reads the same bytes with 16 bytes per lane (the pattern MI355X_MICROARCH.md calibrates:
// FETCH_SIZE = half the bytes).  Known byte counts are printed; FETCH_SIZE of the two dispatches comes from
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- ./fetch_calib
// build: hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(64) k_read8(const double *base, int rows, int ld, double *sink)
{
    // block = one wave = one 64-column window of one "problem" (blockIdx.y), all rows
    const double *T = base + (size_t)blockIdx.y * rows * ld;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)T, 0, (int)((size_t)rows * ld * 8), 0x00020000);
    const unsigned voff = (blockIdx.x * 64 + threadIdx.x) * 8u;
    double s = 0.0;
    for (int i = 0; i < rows; i += 8) {
        double a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(r, voff, (unsigned)(i + u) * ld * 8u, 0);
            a[u] = __hiloint2double((int)w.y, (int)w.x);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += a[u];
    }
    if (s == 123.456) sink[0] = s;
}

__global__ void __launch_bounds__(64) k_read16(const double *base, int rows, int ld, double *sink)
{
    // same bytes, 16 bytes per lane: a wave covers 128 columns per instruction
    const double *T = base + (size_t)blockIdx.y * rows * ld;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)T, 0, (int)((size_t)rows * ld * 8), 0x00020000);
    const unsigned voff = (blockIdx.x * 128 + threadIdx.x * 2) * 8u;
    double s = 0.0;
    for (int i = 0; i < rows; i += 8) {
        u32x4 a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, (unsigned)(i + u) * ld * 8u, 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += __hiloint2double((int)a[u].y, (int)a[u].x) + __hiloint2double((int)a[u].w, (int)a[u].z);
    }
    if (s == 123.456) sink[0] = s;
}

int main()
{
    const int nprob = 256, rows = 4096, ld = 256;                 // 256 x 8 MiB = 2 GiB, read exactly once per kernel
    double *d, *sink;
    const size_t bytes = (size_t)nprob * rows * ld * 8;
    hipMalloc(&d, bytes);
    hipMalloc(&sink, 64);
    hipMemset(d, 0, bytes);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_read8, dim3(ld / 64, nprob), dim3(64), 0, 0, d, rows, ld, sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_read16, dim3(ld / 128, nprob), dim3(64), 0, 0, d, rows, ld, sink);
    hipDeviceSynchronize();
    printf("bytes_read_per_kernel %zu\n", bytes);
    hipFree(d);
    hipFree(sink);
    return 0;
}
