// The reference's [benchmark] cases as a plain program (reference test/radix_sort_tests.cpp:160-193,
// test/blelloch_scan_tests.cpp:84-108, test/reduce_tests.cpp:185-209): same size ladders, same one-line output
// format, zero-initialised input as in the reference, plus a uniform-random line for the sort.
//   ./benchmark [radix|scan|reduce] [max_elements]
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "glu/BlellochScan.hpp"
#include "glu/RadixSort.hpp"
#include "glu/Reduce.hpp"
#include "util/timing.hpp"

using namespace glu;

static const size_t k_sort_sizes[] = {1024,    16384,   65536,    131072,   524288,   1048576,   2097152,
                                      4194304, 8388608, 16777216, 33554432, 67108864, 134217728, 268435456};
static const size_t k_other_sizes[] = {1024, 16384, 65536, 131072, 524288, 1048576, 16777216, 67108864, 134217728, 268435456};

int main(int argc, char** argv)
{
    const char* which = argc > 1 ? argv[1] : "all";
    const size_t max_elements = argc > 2 ? strtoull(argv[2], nullptr, 10) : 268435456ull;
    const bool all = !strcmp(which, "all");
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
            printf("Reduce; Num elements: %zu, Elapsed: %s\n", n, test_timing::human_time(ns).c_str());
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
            printf("BlellochScan; Num elements: %zu, Elapsed: %s\n", n, test_timing::human_time(ns).c_str());
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
            printf("Radix sort; Num elements: %zu, Elapsed: %s\n", n, test_timing::human_time(ns).c_str());

            std::mt19937 gen(0x5EED);
            for (auto& k : keys) k = gen();
            key_buffer.write_data(keys.data(), n * sizeof(GLuint));
            ns = measure_gl_elapsed_time([&]() { radix_sort(key_buffer.handle(), val_buffer.handle(), n); });
            printf("Radix sort (uniform random keys); Num elements: %zu, Elapsed: %s, %.1f Mkeys/s\n", n,
                   test_timing::human_time(ns).c_str(), double(n) / double(ns) * 1e3);
        }
    return 0;
}
