// scatter_bench64.hip -- tuning harness for the 64-bit-key kernels (not part of the product).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include "radix_sort_kernels.hpp"
using namespace glu_hip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void fill64(uint64_t* keys, uint32_t* vals, size_t n)
{
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x, stride = (size_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride)
    {
        uint64_t x = i * 0x9E3779B97F4A7C15ull + 0x1234567;
        x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; x *= 0x94D049BB133111EBull; x ^= x >> 32;
        keys[i] = x;
        vals[i] = (uint32_t) i;
    }
}
__global__ void check64(const uint64_t* keys, const uint32_t* vals, const uint64_t* src, size_t n, uint32_t shift, uint32_t mask,
                        unsigned long long* bad)
{
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x, stride = (size_t) gridDim.x * blockDim.x;
    unsigned long long b = 0;
    for (; i + 1 < n; i += stride)
    {
        uint32_t d0 = (uint32_t) (keys[i] >> shift) & mask, d1 = (uint32_t) (keys[i + 1] >> shift) & mask;
        if (d0 > d1) b++;
        if (d0 == d1 && vals[i] > vals[i + 1]) b++;
        if (src[vals[i]] != keys[i]) b++;
    }
    if (b) atomicAdd(bad, b);
}
struct Ctx { uint64_t *keys, *keys2; uint32_t *vals, *vals2, *table; unsigned long long* bad; size_t n; int cus; hipEvent_t e0, e1; };
template<typename F> float tmin(Ctx& c, int reps, F&& f)
{
    float best = 1e30f;
    for (int r = 0; r < reps; r++)
    {
        CK(hipEventRecord(c.e0)); f(); CK(hipEventRecord(c.e1)); CK(hipEventSynchronize(c.e1));
        float ms; CK(hipEventElapsedTime(&ms, c.e0, c.e1)); best = std::min(best, ms);
    }
    return best;
}
template<int BITS, int THREADS, int KPT, bool CARRY>
void run(Ctx& c, int bpc, uint32_t shift)
{
    constexpr int RADIX = 1 << BITS, TILE = THREADS * KPT;
    using Smem = ScatterSmem<uint64_t, BITS, THREADS, KPT, CARRY>;
    const uint32_t tiles = (uint32_t) ((c.n + TILE - 1) / TILE), nb = std::min<uint32_t>(tiles, c.cus * bpc), mask = RADIX - 1;
    uint32_t* totals = c.table + (size_t) RADIX * nb;
    auto scatter = radix_scatter_kernel<uint64_t, BITS, THREADS, KPT, CARRY, 0, false, 1>;
    CK(hipFuncSetAttribute((const void*) scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem)));
    float tc = tmin(c, 5, [&] { hipLaunchKernelGGL((radix_count_kernel<uint64_t, BITS, THREADS, TILE>), dim3(nb), dim3(THREADS), 0, 0, c.keys, c.table, (uint32_t) c.n, shift, mask, tiles); });
    hipLaunchKernelGGL((radix_row_scan_kernel<256>), dim3(RADIX), dim3(256), 0, 0, c.table, totals, nb);
    float ts = tmin(c, 5, [&] { hipLaunchKernelGGL(scatter, dim3(nb), dim3(THREADS), sizeof(Smem), 0, c.keys, c.vals, c.keys2, c.vals2, c.table, totals, (uint32_t) c.n, shift, mask, tiles, (unsigned long long*) nullptr, 0u, (PassPlan*) nullptr, 0u, (uint32_t*) nullptr); });
    CK(hipGetLastError());
    CK(hipMemset(c.bad, 0, 8));
    hipLaunchKernelGGL(check64, dim3(4096), dim3(256), 0, 0, c.keys2, c.vals2, c.keys, c.n, shift, mask, c.bad);
    unsigned long long bad = 0; CK(hipMemcpy(&bad, c.bad, 8, hipMemcpyDeviceToHost));
    printf("u64 %s bits %d threads %4d kpt %2d tile %5d lds %6zu blk/cu %d | count %.3f ms (%.0f GB/s) | scatter %.3f ms (%.0f GB/s) | pass %.3f %s\n",
           CARRY ? "carry" : "plain", BITS, THREADS, KPT, TILE, sizeof(Smem), bpc, tc, c.n * 8.0 / tc / 1e6, ts, c.n * 24.0 / ts / 1e6, tc + ts, bad ? "WRONG" : "ok");
    fflush(stdout);
}
int main(int argc, char** argv)
{
    Ctx c; c.n = (size_t) 1 << (argc > 1 ? atoi(argv[1]) : 28);
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); c.cus = p.multiProcessorCount;
    CK(hipMalloc(&c.keys, c.n * 8)); CK(hipMalloc(&c.keys2, c.n * 8)); CK(hipMalloc(&c.vals, c.n * 4)); CK(hipMalloc(&c.vals2, c.n * 4));
    CK(hipMalloc(&c.table, (256 * 8192 + 256) * 4)); CK(hipMalloc(&c.bad, 8)); CK(hipEventCreate(&c.e0)); CK(hipEventCreate(&c.e1));
    hipLaunchKernelGGL(fill64, dim3(4096), dim3(256), 0, 0, c.keys, c.vals, c.n); CK(hipDeviceSynchronize());
    const uint32_t shift = 24;
    run<8, 1024, 8, false>(c, 1, shift);
    run<8, 512, 16, true>(c, 1, shift);
    run<8, 512, 16, false>(c, 1, shift);
    run<8, 1024, 10, false>(c, 1, shift);
    run<8, 512, 12, true>(c, 1, shift);
    run<8, 1024, 6, true>(c, 1, shift);
    run<8, 768, 8, true>(c, 1, shift);
    run<8, 768, 10, true>(c, 1, shift);
    run<4, 1024, 8, false>(c, 1, shift);
    run<4, 1024, 8, true>(c, 1, shift);
    run<4, 1024, 12, true>(c, 1, shift);
    run<4, 1024, 10, false>(c, 1, shift);
    run<4, 512, 16, false>(c, 1, shift);
    return 0;
}
