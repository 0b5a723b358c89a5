// cpu_sort_baseline.cpp -- the host-CPU baseline BASELINE.md section 3 asks for next to the GPU number:
// std::sort over an array of struct { uint32_t key; uint32_t val; } ordered by key, single-threaded and on all
// host cores (__gnu_parallel::sort, OpenMP).  TEST / BENCH INFRASTRUCTURE ONLY (see glu_oracle.c header): bench.py's
// cpu_baseline leg times it; the product never links it.  std::sort is not stable, so only the key order of
// its output is comparable with the GPU result; the parity oracle is glu_oracle.c.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <parallel/algorithm>
#include <thread>
#include <vector>

#include <omp.h>

namespace
{
struct Pair
{
    uint32_t key;
    uint32_t val;
};
inline bool by_key(const Pair& a, const Pair& b) { return a.key < b.key; }
} // namespace

extern "C" {

__attribute__((visibility("default"))) unsigned glu_cpu_hardware_threads() { return std::thread::hardware_concurrency(); }

// Sorts `count` pairs (gathered into structs first, untimed) with `threads` threads (1 = std::sort).
// Returns the seconds spent in the sort call itself; writes the sorted keys / vals back.
__attribute__((visibility("default"))) double glu_cpu_sort_pairs(uint32_t* keys, uint32_t* vals, uint64_t count,
                                                                 int threads)
{
    std::vector<Pair> pairs(count);
    for (uint64_t i = 0; i < count; i++) pairs[i] = {keys[i], vals[i]};
    auto t0 = std::chrono::steady_clock::now();
    if (threads <= 1)
    {
        std::sort(pairs.begin(), pairs.end(), by_key);
    }
    else
    {
        omp_set_num_threads(threads);
        __gnu_parallel::sort(pairs.begin(), pairs.end(), by_key);
    }
    auto t1 = std::chrono::steady_clock::now();
    for (uint64_t i = 0; i < count; i++)
    {
        keys[i] = pairs[i].key;
        vals[i] = pairs[i].val;
    }
    return std::chrono::duration<double>(t1 - t0).count();
}
}
