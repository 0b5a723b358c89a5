// The reference's [benchmark] cases as a plain program (reference test/radix_sort_tests.cpp:160-193,
// test/blelloch_scan_tests.cpp:84-108, test/reduce_tests.cpp:185-209): same size ladders, same one-line output
// format, zero-initialised input as in the reference (extra columns appended: GB/s under the operator's algorithmic
// bytes and % of the 8 TB/s HBM peak), plus a uniform-random line for the sort and, with a third argument "cpu", the
// host's single-thread std::sort of the same pairs for sizes up to 2^22.
//   ./benchmark [radix|scan|reduce|all] [max_elements] [cpu]
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "glu/BlellochScan.hpp"
#include "glu/RadixSort.hpp"
#include "glu/Reduce.hpp"
#include "util/timing.hpp"

using namespace glu;

static constexpr double k_hbm_peak_GBps = 8000.0; // MI355X HBM3E

// ", X GB/s (Y % of 8 TB/s)" for `bytes` moved in `ns`
static std::string rate_columns(double bytes, uint64_t ns)
{
    char buf[96];
    const double gbps = bytes / double(ns);
    snprintf(buf, sizeof(buf), ", %.0f GB/s (%.1f %% of 8 TB/s)", gbps, 100.0 * gbps / k_hbm_peak_GBps);
    return buf;
}

static const size_t k_sort_sizes[] = {1024,    16384,   65536,    131072,   524288,   1048576,   2097152,
                                      4194304, 8388608, 16777216, 33554432, 67108864, 134217728, 268435456};
static const size_t k_other_sizes[] = {1024, 16384, 65536, 131072, 524288, 1048576, 16777216, 67108864, 134217728, 268435456};

int main(int argc, char** argv)
{
    const char* which = argc > 1 ? argv[1] : "all";
    const size_t max_elements = argc > 2 ? strtoull(argv[2], nullptr, 10) : 268435456ull;
    const bool all = !strcmp(which, "all");
    const bool with_cpu = argc > 3 && !strcmp(argv[3], "cpu");
    char info[256];
    GLU_CHECK_STATUS(glu_device_info(info, sizeof(info)));
    printf("Device: %s\n", info);

    if (all || !strcmp(which, "reduce"))
        for (size_t n : k_other_sizes)
        {
            if (n > max_elements) break;
            std::vector<GLuint> data(n);
            ShaderStorageBuffer buffer(data);
            Reduce reduce(DataType_Uint, ReduceOperator_Sum);
            reduce(buffer.handle(), n); // warm-up (the reference times a single cold shot)
            uint64_t ns = measure_gl_elapsed_time([&]() { reduce(buffer.handle(), n); });
            printf("Reduce; Num elements: %zu, Elapsed: %s%s\n", n, test_timing::human_time(ns).c_str(),
                   rate_columns(4.0 * double(n), ns).c_str()); // 4 B/element, read once
        }

    if (all || !strcmp(which, "scan"))
        for (size_t n : k_other_sizes)
        {
            if (n > max_elements) break;
            std::vector<GLuint> data(n);
            ShaderStorageBuffer buffer(data);
            BlellochScan blelloch_scan(DataType_Uint);
            blelloch_scan(buffer.handle(), n);
            uint64_t ns = measure_gl_elapsed_time([&]() { blelloch_scan(buffer.handle(), n); });
            printf("BlellochScan; Num elements: %zu, Elapsed: %s%s\n", n, test_timing::human_time(ns).c_str(),
                   rate_columns(8.0 * double(n), ns).c_str()); // 8 B/element, read + write
        }

    if (all || !strcmp(which, "radix"))
        for (size_t n : k_sort_sizes)
        {
            if (n > max_elements) break;
            std::vector<GLuint> keys(n), vals(n);
            ShaderStorageBuffer key_buffer(keys), val_buffer(vals);
            RadixSort radix_sort;
            radix_sort.prepare_internal_buffers(n);
            radix_sort(key_buffer.handle(), val_buffer.handle(), n);
            uint64_t ns = measure_gl_elapsed_time([&]() { radix_sort(key_buffer.handle(), val_buffer.handle(), n); });
            // 80 B/pair: what the 4 passes of 8-bit digits move (the reference's 8 x 4-bit structure would be 160 B/pair).
            // The reference's input here is all-zero keys: from 2^22 elements up every pass has a constant digit and its
            // scatter is skipped on the device, only the 4 count kernels read the keys (16 B/pair).
            const bool skipped = n >= (size_t(1) << 22);
            printf("Radix sort; Num elements: %zu, Elapsed: %s%s%s\n", n, test_timing::human_time(ns).c_str(),
                   rate_columns((skipped ? 16.0 : 80.0) * double(n), ns).c_str(), skipped ? " [constant-digit passes skipped]" : "");

            std::mt19937 gen(0x5EED);
            for (auto& k : keys) k = gen();
            key_buffer.write_data(keys.data(), n * sizeof(GLuint));
            ns = measure_gl_elapsed_time([&]() { radix_sort(key_buffer.handle(), val_buffer.handle(), n); });
            printf("Radix sort (uniform random keys); Num elements: %zu, Elapsed: %s, %.1f Mkeys/s%s\n", n,
                   test_timing::human_time(ns).c_str(), double(n) / double(ns) * 1e3, rate_columns(80.0 * double(n), ns).c_str());
            if (with_cpu && n <= (size_t(1) << 22))
            {
                struct Pair { GLuint key, val; };
                std::vector<Pair> pairs(n);
                for (size_t i = 0; i < n; i++) pairs[i] = {keys[i], GLuint(i)};
                const auto t0 = std::chrono::steady_clock::now();
                std::sort(pairs.begin(), pairs.end(), [](const Pair& a, const Pair& b) { return a.key < b.key; });
                const double cpu_ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
                printf("CPU std::sort (1 of %u hardware threads); Num elements: %zu, Elapsed: %.3f ms, %.1f Mkeys/s\n",
                       std::thread::hardware_concurrency(), n, cpu_ns * 1e-6, double(n) / cpu_ns * 1e3);
            }
        }
    return 0;
}
