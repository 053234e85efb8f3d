// grid_sync.hip -- what a grid-wide barrier costs on this part: cooperative groups' grid.sync() against a hand-written
// sense-free counter barrier (agent-scope fence + atomic add by one lane per workgroup, sc1-load polling of the counter).
// hipcc --offload-arch=gfx950 -O3 -o grid_sync grid_sync.hip ; ./grid_sync [workgroups] [threads]
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
namespace cg = cooperative_groups;

__global__ void k_cg(int iters, double *out)
{
    cg::grid_group g = cg::this_grid();
    double acc = 0.0;
    for (int i = 0; i < iters; ++i) {
        acc += (double)i;
        g.sync();
    }
    if (acc < 0) out[0] = acc;
}

// counter barrier: *ctr counts arrivals of all barriers so far (never reset): barrier number b completes at b * nwg
__device__ __forceinline__ void ctr_barrier(unsigned *ctr, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                         // release: this workgroup's writes are visible device-wide
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __threadfence();                                         // acquire
    }
    __syncthreads();
}

__global__ void k_ctr(int iters, unsigned *ctr, double *out)
{
    double acc = 0.0;
    const unsigned nwg = gridDim.x;
    for (int i = 0; i < iters; ++i) {
        acc += (double)i;
        ctr_barrier(ctr, (unsigned)(i + 1) * nwg);
    }
    if (acc < 0) out[0] = acc;
}

int main(int argc, char **argv)
{
    int nwg = argc > 1 ? atoi(argv[1]) : 257, nt = argc > 2 ? atoi(argv[2]) : 256, iters = 2000;
    double *out; unsigned *ctr;
    hipMalloc(&out, 8); hipMalloc(&ctr, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        void *args[] = {&iters, &out};
        hipEventRecord(e0);
        hipError_t e = hipLaunchCooperativeKernel((const void *)k_cg, dim3(nwg), dim3(nt), args, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("cooperative grid.sync: %d wgs x %d threads: %s, %.3f us per barrier\n", nwg, nt, hipGetErrorString(e), 1e3 * ms / iters);
    }
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(ctr, 0, 4);
        void *args[] = {&iters, &ctr, &out};
        hipEventRecord(e0);
        hipError_t e = hipLaunchCooperativeKernel((const void *)k_ctr, dim3(nwg), dim3(nt), args, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("counter barrier      : %d wgs x %d threads: %s, %.3f us per barrier\n", nwg, nt, hipGetErrorString(e), 1e3 * ms / iters);
    }
    return 0;
}
