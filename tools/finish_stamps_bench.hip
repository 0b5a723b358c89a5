// finish_stamps_bench.hip -- round 5: what bounds the in-LDS pass of a sort that ends in LDS (radix_lds_finish.hpp)?
// Launches the PRODUCT kernel template radix_finish_sort_kernel -- its STAMPS = true instantiation, which the library does
// not build -- on the input the pass sees inside a sort of 2^log2 uniformly drawn pairs: 65536 runs of Poisson-like lengths, keys
// = run << (key bits - 16) | random low bits, values = position, run starts in `starts`.  Prints the kernel time (events, median
// of 9) and the s_memtime phase clock of the first and the last wave of every workgroup, summed over the launch (100 MHz ticks):
//   load | ranking | scan | staging | ties | store | whole workgroup
// and the output is compared with std::stable_sort on sampled runs.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I gl-radix-sort_amd/csrc -I include -o tools/finish_stamps_bench tools/finish_stamps_bench.hip
//   tools/finish_stamps_bench [log2 pairs = 28] [key bytes = 4] [rank bits = 16]
// Records: profiles/r05/finish_stamps_*.txt.  Not part of the product.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "radix_lds_finish.hpp"

using namespace glu_hip;

#define CK(x)                                                                                                          \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e = (x);                                                                                            \
        if (e != hipSuccess)                                                                                           \
        {                                                                                                              \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));                                                       \
            exit(1);                                                                                                   \
        }                                                                                                              \
    } while (0)

template<typename KeyT>
__global__ void fill_kernel(KeyT* keys, uint32_t* vals, const uint32_t* starts, uint32_t nruns, uint32_t n, uint32_t low_bits, uint32_t seed)
{
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
    {
        // the run of position i: binary search in starts
        uint32_t lo = 0, hi = nruns;
        while (hi - lo > 1)
        {
            const uint32_t mid = (lo + hi) / 2;
            if (starts[mid] <= i) lo = mid; else hi = mid;
        }
        uint64_t x = (uint64_t) i * 0x9E3779B97F4A7C15ull + seed;
        x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; x *= 0x94D049BB133111EBull; x ^= x >> 32;
        const uint64_t low = low_bits >= 64 ? x : (x & ((1ull << low_bits) - 1ull));
        keys[i] = (KeyT) (((uint64_t) lo << low_bits) | low);
        vals[i] = (uint32_t) i;
    }
}

template<typename KeyT, int THREADS, int KPT>
void run(uint32_t log2n, uint32_t rank_bits)
{
    const uint32_t n = 1u << log2n, nruns = kFinishRuns, low_bits = 8 * sizeof(KeyT) - 16;
    // run lengths of uniformly drawn top bits
    std::mt19937_64 rng(7);
    std::vector<uint32_t> len(nruns, 0), starts(nruns + 1, 0);
    {
        std::binomial_distribution<uint32_t> b;
        uint32_t left = n;
        for (uint32_t r = 0; r < nruns; r++)
        {
            std::binomial_distribution<uint32_t> d(left, 1.0 / (nruns - r));
            len[r] = r + 1 == nruns ? left : std::min<uint32_t>(d(rng), (uint32_t) (THREADS * KPT));
            left -= len[r];
            starts[r + 1] = starts[r] + len[r];
        }
    }
    KeyT *keys, *keys0;
    uint32_t *vals, *vals0, *d_starts;
    unsigned long long* d_stamps;
    PassPlan* plan;
    CK(hipMalloc(&keys, (size_t) n * sizeof(KeyT)));
    CK(hipMalloc(&keys0, (size_t) n * sizeof(KeyT)));
    CK(hipMalloc(&vals, (size_t) n * 4));
    CK(hipMalloc(&vals0, (size_t) n * 4));
    CK(hipMalloc(&d_starts, (nruns + 1) * 4));
    CK(hipMalloc(&d_stamps, (size_t) nruns * 16 * 8));
    CK(hipMalloc(&plan, sizeof(PassPlan)));
    CK(hipMemset(plan, 0, sizeof(PassPlan)));
    const uint32_t geo = 3; // (any: the kernel is launched with the geometry number the plan holds)
    CK(hipMemcpy(&plan->finish, &geo, 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_starts, starts.data(), (nruns + 1) * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fill_kernel<KeyT>, dim3(4096), dim3(256), 0, 0, keys0, vals0, d_starts, nruns, n, low_bits, 99u);
    CK(hipDeviceSynchronize());
    using Smem = FinishSmem<KeyT, THREADS, KPT, true>;
    auto kern = radix_finish_sort_kernel<KeyT, THREADS, KPT, true, false, false, true>;
    auto kern_plain = radix_finish_sort_kernel<KeyT, THREADS, KPT, true, false, false, false>;
    CK(hipFuncSetAttribute((const void*) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem)));
    CK(hipFuncSetAttribute((const void*) kern_plain, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(Smem)));
    const uint32_t rank_from = low_bits > rank_bits ? ((low_bits - rank_bits) / 8u) * 8u : 0u;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int stamped = 0; stamped < 2; stamped++)
    {
        std::vector<float> ms;
        for (int rep = 0; rep < 10; rep++)
        {
            CK(hipMemcpy(keys, keys0, (size_t) n * sizeof(KeyT), hipMemcpyDeviceToDevice));
            CK(hipMemcpy(vals, vals0, (size_t) n * 4, hipMemcpyDeviceToDevice));
            CK(hipMemset(d_stamps, 0, (size_t) nruns * 16 * 8));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(stamped ? kern : kern_plain, dim3(nruns), dim3(THREADS), sizeof(Smem), 0, keys, vals, keys, vals,
                               (const uint32_t*) d_starts, low_bits, (const PassPlan*) plan, 0u, geo, 0u, nruns, (const uint32_t*) nullptr, 0u,
                               rank_bits, stamped ? d_stamps : (unsigned long long*) nullptr, (const uint32_t*) nullptr, 0u);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (rep) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%s  keys %zu B  tile %d x %d  rank_from %u  2^%u pairs: median %.3f ms  min %.3f  (%.0f GB/s at %zu B/pair)\n",
               stamped ? "with stamps   " : "product kernel", sizeof(KeyT), THREADS, KPT, rank_from, log2n, ms[ms.size() / 2], ms[0],
               (double) n * 2 * (sizeof(KeyT) + 4) / (ms[ms.size() / 2] * 1e-3) / 1e9, 2 * (sizeof(KeyT) + 4));
        if (stamped)
        {
            std::vector<unsigned long long> all((size_t) nruns * 16);
            CK(hipMemcpy(all.data(), d_stamps, all.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long st[16] = {};
            for (size_t w = 0; w < nruns; w++)
                for (int i = 0; i < 16; i++) st[i] += all[w * 16 + i];
            const char* names[8] = {"load", "ranking", "scan", "staging", "ties", "store", "-", "workgroup"};
            for (int w = 0; w < 2; w++)
            {
                printf("  %s wave:", w ? "last " : "first");
                for (int i = 0; i < 8; i++)
                    if (i != 6) printf("  %s %.1f %%", names[i], 100.0 * (double) st[w * 8 + i] / (double) st[w * 8 + 7]);
                printf("   (ticks per workgroup %.0f)\n", (double) st[w * 8 + 7] / nruns);
            }
        }
    }
    // check sampled runs against std::stable_sort
    std::vector<KeyT> hk(n), hk0(n);
    std::vector<uint32_t> hv(n);
    CK(hipMemcpy(hk.data(), keys, (size_t) n * sizeof(KeyT), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hv.data(), vals, (size_t) n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hk0.data(), keys0, (size_t) n * sizeof(KeyT), hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (uint32_t r = 0; r < nruns; r += 997)
    {
        std::vector<std::pair<KeyT, uint32_t>> ref;
        for (uint32_t i = starts[r]; i < starts[r + 1]; i++) ref.push_back({hk0[i], i});
        std::stable_sort(ref.begin(), ref.end(), [](auto& a, auto& b) { return a.first < b.first; });
        for (uint32_t i = starts[r]; i < starts[r + 1]; i++)
            if (hk[i] != ref[i - starts[r]].first || hv[i] != ref[i - starts[r]].second) bad++;
    }
    printf("  output of sampled runs against std::stable_sort: %zu differences\n", bad);
    CK(hipFree(keys)); CK(hipFree(keys0)); CK(hipFree(vals)); CK(hipFree(vals0)); CK(hipFree(d_starts)); CK(hipFree(d_stamps)); CK(hipFree(plan));
}

int main(int argc, char** argv)
{
    const uint32_t log2n = argc > 1 ? (uint32_t) atoi(argv[1]) : 28u;
    const int key_bytes = argc > 2 ? atoi(argv[2]) : 4;
    const uint32_t rank_bits = argc > 3 ? (uint32_t) atoi(argv[3]) : 16u;
#ifndef FSB_T32
#define FSB_T32 256
#define FSB_K32 18
#endif
#ifndef FSB_T64
#define FSB_T64 512
#define FSB_K64 9
#endif
    if (key_bytes == 4)
        run<uint32_t, FSB_T32, FSB_K32>(log2n, rank_bits);
    else
        run<uint64_t, FSB_T64, FSB_K64>(log2n, rank_bits);
    return 0;
}
