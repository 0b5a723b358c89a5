// grid_sync_bench.hip -- what does a grid-wide barrier cost on this chip?  (Research for the launch-bound sizes: a sort of
// 2^14 .. 2^21 pairs is 8-12 dependent launches of ~4-5 us each; one cooperative launch with grid barriers between the
// phases would replace the launch boundaries if a barrier is much cheaper than that.)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/grid_sync_bench tools/grid_sync_bench.hip
// Always run it under `timeout`: a barrier bug would spin forever.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void coop_kernel(int syncs, unsigned* out)
{
    cg::grid_group grid = cg::this_grid();
    unsigned acc = threadIdx.x;
    for (int i = 0; i < syncs; i++)
    {
        acc = acc * 1664525u + 1013904223u;
        grid.sync();
    }
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}

// hand-made barrier: one counter, every workgroup adds 1 and spins until the count reaches (phase + 1) * blocks.
// Only safe when all workgroups are co-resident (cooperative launch guarantees it).
__global__ void atomic_kernel(int syncs, unsigned* counter, unsigned* out)
{
    unsigned acc = threadIdx.x;
    const unsigned nb = gridDim.x;
    for (int i = 0; i < syncs; i++)
    {
        acc = acc * 1664525u + 1013904223u;
        __syncthreads();
        if (threadIdx.x == 0)
        {
            __threadfence();
            atomicAdd(counter, 1u);
            const unsigned target = (unsigned) (i + 1) * nb;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            __threadfence();
        }
        __syncthreads();
    }
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}

int main(int argc, char** argv)
{
    int blocks = argc > 1 ? atoi(argv[1]) : 256, threads = argc > 2 ? atoi(argv[2]) : 256;
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("%s CUs %d cooperativeLaunch %d, grid %d x %d\n", p.gcnArchName, p.multiProcessorCount, p.cooperativeLaunch, blocks, threads);
    unsigned *out, *counter;
    CK(hipMalloc(&out, 64));
    CK(hipMalloc(&counter, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int syncs : {0, 1, 12, 100})
    {
        float best = 1e9f;
        for (int r = 0; r < 6; r++)
        {
            void* args[] = {&syncs, &out};
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel((const void*) coop_kernel, dim3(blocks), dim3(threads), args, 0, 0));
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        printf("cooperative launch, %3d grid.sync(): %8.2f us\n", syncs, best * 1e3f);
    }
    for (int syncs : {0, 1, 12, 100})
    {
        float best = 1e9f;
        for (int r = 0; r < 6; r++)
        {
            CK(hipMemset(counter, 0, 4));
            void* args[] = {&syncs, &counter, &out};
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel((const void*) atomic_kernel, dim3(blocks), dim3(threads), args, 0, 0));
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        printf("cooperative launch, %3d atomic barriers: %8.2f us\n", syncs, best * 1e3f);
    }
    {   // for scale: the same number of empty ordinary launches
        for (int launches : {1, 12})
        {
            float best = 1e9f;
            int zero = 0;
            for (int r = 0; r < 6; r++)
            {
                CK(hipEventRecord(e0));
                for (int i = 0; i < launches; i++) hipLaunchKernelGGL(atomic_kernel, dim3(blocks), dim3(threads), 0, 0, zero, counter, out);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
            printf("%2d ordinary empty launches: %8.2f us\n", launches, best * 1e3f);
        }
    }
    return 0;
}
