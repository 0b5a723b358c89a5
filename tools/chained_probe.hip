// chained_probe.hip -- where a chained (look-back) scatter pass of a mid-size sort spends its time (tuning harness).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I gl-radix-sort_amd/csrc -o tools/chained_probe tools/chained_probe.hip
//   ./tools/chained_probe [log2n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "radix_sort_kernels.hpp"
using namespace glu_hip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 20;
    const size_t n = (size_t) 1 << log2n;
    constexpr int THREADS = 512, KPT = 8, TILE = THREADS * KPT, RADIX = 256;
    const uint32_t tiles = (uint32_t) ((n + TILE - 1) / TILE);
    std::vector<uint32_t> h(n);
    uint64_t x = 88172645463325252ull;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t) (x >> 16); }
    uint32_t *keys, *vals, *keys2, *vals2, *ghist, *chain;
    unsigned long long* st;
    CK(hipMalloc(&keys, n * 4)); CK(hipMalloc(&vals, n * 4)); CK(hipMalloc(&keys2, n * 4)); CK(hipMalloc(&vals2, n * 4));
    const size_t row_words = ((size_t) tiles + 15) & ~(size_t) 15, grow_words = (row_words / 16 + 15) & ~(size_t) 15;
    const size_t pass_words = (size_t) RADIX * (row_words + grow_words);
    CK(hipMalloc(&ghist, 4 * RADIX * 4)); CK(hipMalloc(&chain, 4 * pass_words * 4)); CK(hipMalloc(&st, 64));
    CK(hipMemcpy(keys, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(vals, h.data(), n * 4, hipMemcpyHostToDevice));
    using Smem = ScatterSmem<uint32_t, 8, THREADS, KPT, false, 1, true>;
    auto scatter = radix_scatter_kernel<uint32_t, 8, THREADS, KPT, false, 0, true, 6, 1, false, false, false, true, false, true>;
    auto plain = radix_scatter_kernel<uint32_t, 8, THREADS, KPT, false, 0, true, 6, 1, false, false, false, true, false, false>;
    CK(hipFuncSetAttribute((const void*) scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem)));
    CK(hipFuncSetAttribute((const void*) plain, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem)));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    for (int rep = 0; rep < 3; rep++)
    {
        CK(hipMemset(ghist, 0, 4 * RADIX * 4));
        CK(hipMemset(st, 0, 64));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((radix_hist_kernel<THREADS, TILE, 4>), dim3(tiles), dim3(THREADS), 0, 0, keys, ghist, chain, (uint32_t) n, 4u, 0x18100800u, 0x08080808u);
        CK(hipEventRecord(e1));
        hipLaunchKernelGGL(scatter, dim3(tiles), dim3(THREADS), sizeof(Smem), 0, keys, vals, keys2, vals2, (const uint32_t*) nullptr, (const uint32_t*) ghist,
                           (uint32_t) n, 0u, 255u, tiles, st, 0u, (PassPlan*) nullptr, 0u, chain);
        CK(hipEventRecord(e2));
        CK(hipEventSynchronize(e2));
        float a, b;
        CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
        unsigned long long hs[8];
        CK(hipMemcpy(hs, st, 64, hipMemcpyDeviceToHost));
        printf("2^%d pairs, %u tiles: hist %.1f us, chained scatter %.1f us | wave 0 cycles per workgroup:", log2n, tiles, a * 1e3, b * 1e3);
        const char* names[8] = {"load-issue", "load-wait", "rank", "bar", "offsets", "lookback+stage", "write-out", "bar"};
        for (int i = 0; i < 8; i++) printf(" %s %.0f", names[i], hs[i] / (double) tiles);
        printf("\n");
    }
    return 0;
}
