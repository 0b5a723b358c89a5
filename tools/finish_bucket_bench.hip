// finish_bucket_bench.hip -- round 6: the in-LDS pass by one bucket round on unique words (radix_lds_bucket.hpp) against round 5's
// ballot-ranked kernel (radix_finish_sort_kernel), both PRODUCT kernel templates, on the input the pass sees inside a sort of
// 2^log2 pairs: 65536 runs of Poisson-like lengths, keys = run << low_bits | low bits, values = position, run starts in `starts`.
// Both kernels sort the same arrays in place; the outputs are compared element by element and sampled runs against
// std::stable_sort.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I gl-radix-sort_amd/csrc -I include -o tools/finish_bucket_bench tools/finish_bucket_bench.hip
//   tools/finish_bucket_bench [log2 pairs = 28] [key bytes = 4] [mode = 0] [low_bits = key bits - 16]
//   mode 0: uniformly drawn low bits   1: four distinct low-bit values per run (crowded: every workgroup takes the ballot rounds)
//        2: one run in 16 crowded       3: ragged (a third of the runs short: 0 .. 300 pairs)
// Records: profiles/r06/finish_bucket_*.txt.  Not part of the product.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "radix_lds_bucket.hpp"

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
__global__ void fill_kernel(KeyT* keys, uint32_t* vals, const uint32_t* starts, uint32_t nruns, uint32_t n, uint32_t low_bits, uint32_t seed,
                            uint32_t mode)
{
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
    {
        uint32_t lo = 0, hi = nruns;
        while (hi - lo > 1)
        {
            const uint32_t mid = (lo + hi) / 2;
            if (starts[mid] <= i) lo = mid; else hi = mid;
        }
        uint64_t x = (uint64_t) i * 0x9E3779B97F4A7C15ull + seed;
        x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; x *= 0x94D049BB133111EBull; x ^= x >> 32;
        uint64_t low = low_bits >= 64 ? x : (x & ((1ull << low_bits) - 1ull));
        const bool crowded = mode == 1 || (mode == 2 && (lo & 15u) == 3u);
        if (crowded) low = (low & 3ull) * 0x0101010101ull & (low_bits >= 64 ? ~0ull : ((1ull << low_bits) - 1ull));
        keys[i] = (KeyT) (((uint64_t) lo << low_bits) | low);
        vals[i] = (uint32_t) i ^ 0x5A5A0000u;
    }
}

template<typename KeyT, int THREADS, int KPT>
void run(uint32_t log2n, uint32_t mode, uint32_t low_bits, uint32_t geo)
{
    const uint32_t n = (1u << log2n) - (mode == 3 ? 3u : 0u), nruns = kFinishRuns;
    std::mt19937_64 rng(7);
    std::vector<uint32_t> len(nruns, 0), starts(nruns + 1, 0);
    {
        uint32_t left = n;
        for (uint32_t r = 0; r < nruns; r++)
        {
            std::binomial_distribution<uint32_t> d(left, 1.0 / (nruns - r));
            uint32_t l = r + 1 == nruns ? left : std::min<uint32_t>(d(rng), (uint32_t) (THREADS * KPT));
            if (mode == 3 && r % 3 == 1 && r + 1 != nruns) l = std::min<uint32_t>(l, (uint32_t) (rng() % 300));
            if (r + 1 == nruns && l > (uint32_t) (THREADS * KPT)) l = left; // (the last run takes what is left: may be long -- skipped by both kernels)
            len[r] = l;
            left -= l;
            starts[r + 1] = starts[r] + len[r];
        }
    }
    KeyT *keys, *keys0, *keys_ref;
    uint32_t *vals, *vals0, *vals_ref, *d_starts;
    PassPlan* plan;
    uint32_t* d_flags;
    CK(hipMalloc(&d_flags, crowded_list_words(nruns) * 4));
    CK(hipMalloc(&keys, (size_t) n * sizeof(KeyT)));
    CK(hipMalloc(&keys0, (size_t) n * sizeof(KeyT)));
    CK(hipMalloc(&keys_ref, (size_t) n * sizeof(KeyT)));
    CK(hipMalloc(&vals, (size_t) n * 4));
    CK(hipMalloc(&vals0, (size_t) n * 4));
    CK(hipMalloc(&vals_ref, (size_t) n * 4));
    CK(hipMalloc(&d_starts, (nruns + 1) * 4));
    CK(hipMalloc(&plan, sizeof(PassPlan)));
    CK(hipMemset(plan, 0, sizeof(PassPlan)));
    CK(hipMemcpy(&plan->finish, &geo, 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_starts, starts.data(), (nruns + 1) * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fill_kernel<KeyT>, dim3(4096), dim3(256), 0, 0, keys0, vals0, d_starts, nruns, n, low_bits, 99u, mode);
    CK(hipDeviceSynchronize());
    using SmemOld = FinishSmem<KeyT, THREADS, KPT, true>;
    using SmemNew = BucketSmem<KeyT, THREADS, KPT, true>;
    auto kern_old = radix_finish_sort_kernel<KeyT, THREADS, KPT, true, false, false, false>;
    auto kern_new = radix_finish_bucket_kernel<KeyT, THREADS, KPT, true, false, false>;
    CK(hipFuncSetAttribute((const void*) kern_old, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(SmemOld)));
    CK(hipFuncSetAttribute((const void*) kern_new, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(SmemNew)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("keys %zu B  tile %d x %d  low_bits %u  mode %u  n %u   LDS old %zu B  new %zu B\n", sizeof(KeyT), THREADS, KPT, low_bits, mode, n,
           sizeof(SmemOld), sizeof(SmemNew));
    auto kern_crowded = radix_finish_sort_kernel<KeyT, THREADS, KPT, true, true, false, false>;
    CK(hipFuncSetAttribute((const void*) kern_crowded, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof(SmemOld)));
    hipEvent_t e2;
    CK(hipEventCreate(&e2));
    for (int which = 0; which < 2; which++)
    {
        std::vector<float> ms, ms2;
        for (int rep = 0; rep < 10; rep++)
        {
            CK(hipMemcpy(keys, keys0, (size_t) n * sizeof(KeyT), hipMemcpyDeviceToDevice));
            CK(hipMemcpy(vals, vals0, (size_t) n * 4, hipMemcpyDeviceToDevice));
            CK(hipMemset(d_flags, 0, kCrowdedLists * kCrowdedCountStride * 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            if (which == 0)
                hipLaunchKernelGGL(kern_old, dim3(nruns), dim3(THREADS), sizeof(SmemOld), 0, keys, vals, keys, vals, (const uint32_t*) d_starts,
                                   low_bits, (const PassPlan*) plan, 0u, geo, 0u, nruns, (const uint32_t*) nullptr, 0u, 16u,
                                   (unsigned long long*) nullptr, (const uint32_t*) nullptr, 0u);
            else
                hipLaunchKernelGGL(kern_new, dim3(nruns), dim3(THREADS), sizeof(SmemNew), 0, keys, vals, keys, vals, (const uint32_t*) d_starts,
                                   low_bits, (const PassPlan*) plan, 0u, geo, 0u, nruns, d_flags);
            CK(hipEventRecord(e1, 0));
            if (which == 1) // the runs the bucket kernel listed: round 5's kernel, 8192 workgroups that loop over the list
                hipLaunchKernelGGL(kern_crowded, dim3(8192), dim3(THREADS), sizeof(SmemOld), 0, keys, vals, keys, vals, (const uint32_t*) d_starts,
                                   low_bits, (const PassPlan*) plan, 0u, 0u, 0u, nruns, (const uint32_t*) nullptr, 0u, 16u,
                                   (unsigned long long*) nullptr, (const uint32_t*) d_flags, 0u);
            CK(hipEventRecord(e2, 0));
            CK(hipEventSynchronize(e2));
            CK(hipGetLastError());
            float t, t2;
            CK(hipEventElapsedTime(&t, e0, e1));
            CK(hipEventElapsedTime(&t2, e1, e2));
            if (rep) ms.push_back(t), ms2.push_back(t2);
        }
        std::sort(ms.begin(), ms.end());
        std::sort(ms2.begin(), ms2.end());
        printf("  %s: median %.3f ms  min %.3f  (%.0f GB/s at %zu B/pair)", which ? "bucket round (round 6)" : "ballot rounds (round 5)",
               ms[ms.size() / 2], ms[0], (double) n * 2 * (sizeof(KeyT) + 4) / (ms[ms.size() / 2] * 1e-3) / 1e9, 2 * (sizeof(KeyT) + 4));
        if (which == 1)
        {
            std::vector<uint32_t> hc(kCrowdedLists * kCrowdedCountStride);
            CK(hipMemcpy(hc.data(), d_flags, hc.size() * 4, hipMemcpyDeviceToHost));
            uint32_t flagged = 0;
            for (uint32_t k = 0; k < kCrowdedLists; k++) flagged += std::min(hc[k * kCrowdedCountStride], crowded_list_capacity(nruns));
            printf("  + flagged runs by the ballot rounds: median %.3f ms (%u of %u runs)", ms2[ms2.size() / 2], flagged, nruns);
        }
        printf("\n");
        if (which == 0)
        {
            CK(hipMemcpy(keys_ref, keys, (size_t) n * sizeof(KeyT), hipMemcpyDeviceToDevice));
            CK(hipMemcpy(vals_ref, vals, (size_t) n * 4, hipMemcpyDeviceToDevice));
        }
    }
    std::vector<KeyT> hk(n), hk0(n), hkr(n);
    std::vector<uint32_t> hv(n), hv0(n), hvr(n);
    CK(hipMemcpy(hk.data(), keys, (size_t) n * sizeof(KeyT), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hv.data(), vals, (size_t) n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hkr.data(), keys_ref, (size_t) n * sizeof(KeyT), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hvr.data(), vals_ref, (size_t) n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hk0.data(), keys0, (size_t) n * sizeof(KeyT), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hv0.data(), vals0, (size_t) n * 4, hipMemcpyDeviceToHost));
    size_t diff = 0, first = (size_t) -1;
    for (size_t i = 0; i < n; i++)
        if (hk[i] != hkr[i] || hv[i] != hvr[i])
        {
            if (first == (size_t) -1) first = i;
            diff++;
        }
    printf("  bucket round against ballot rounds, all %u pairs: %zu differences", n, diff);
    if (diff)
    {
        uint32_t r = (uint32_t) (std::upper_bound(starts.begin(), starts.end(), (uint32_t) first) - starts.begin()) - 1;
        printf("  (first at %zu: run %u [%u, %u), got key %llx val %x, want key %llx val %x)", first, r, starts[r], starts[r + 1],
               (unsigned long long) hk[first], hv[first], (unsigned long long) hkr[first], hvr[first]);
    }
    printf("\n");
    size_t bad = 0;
    for (uint32_t r = 0; r < nruns; r += 499)
    {
        if (starts[r + 1] - starts[r] > (uint32_t) (THREADS * KPT)) continue;
        std::vector<std::pair<KeyT, uint32_t>> ref;
        for (uint32_t i = starts[r]; i < starts[r + 1]; i++) ref.push_back({hk0[i], hv0[i]});
        std::stable_sort(ref.begin(), ref.end(), [](auto& a, auto& b) { return a.first < b.first; });
        for (uint32_t i = starts[r]; i < starts[r + 1]; i++)
            if (hk[i] != ref[i - starts[r]].first || hv[i] != ref[i - starts[r]].second) bad++;
    }
    printf("  bucket round, sampled runs against std::stable_sort: %zu differences\n", bad);
    CK(hipFree(keys)); CK(hipFree(keys0)); CK(hipFree(keys_ref)); CK(hipFree(vals)); CK(hipFree(vals0)); CK(hipFree(vals_ref));
    CK(hipFree(d_starts)); CK(hipFree(plan));
}

int main(int argc, char** argv)
{
    const uint32_t log2n = argc > 1 ? (uint32_t) atoi(argv[1]) : 28u;
    const int key_bytes = argc > 2 ? atoi(argv[2]) : 4;
    const uint32_t mode = argc > 3 ? (uint32_t) atoi(argv[3]) : 0u;
    const uint32_t low_bits = argc > 4 ? (uint32_t) atoi(argv[4]) : (uint32_t) key_bytes * 8u - 16u;
#ifndef FBB_T32
#define FBB_T32 256
#define FBB_K32 18
#endif
#ifndef FBB_T64
#define FBB_T64 512
#define FBB_K64 9
#endif
    if (key_bytes == 4)
        run<uint32_t, FBB_T32, FBB_K32>(log2n, mode, low_bits, 3u);
    else
        run<uint64_t, FBB_T64, FBB_K64>(log2n, mode, low_bits, 3u);
    return 0;
}
