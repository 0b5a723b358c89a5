// C++ API tests of glu::BlellochScan -- the reference's cases (reference test/blelloch_scan_tests.cpp:12-82:
// same sizes, seed 123, values in [0,100), same exact comparison with std::exclusive_scan) + other data types.
#include <cstring>
#include <numeric>
#include <vector>

#include "glu/BlellochScan.hpp"
#include "util/golden_vectors.hpp"
#include "util/minstd_inputs.hpp"
#include "util/mini_test.hpp"

using namespace glu;

TEST_CASE("BlellochScan-simple")
{
    ShaderStorageBuffer buffer(golden::k_scan_simple_input);
    BlellochScan blelloch_scan(DataType_Uint);
    blelloch_scan(buffer.handle(), golden::k_scan_simple_input.size());
    CHECK(buffer.get_data<GLuint>() == golden::k_scan_simple_expected);
}

TEST_CASE("BlellochScan-multiple-sizes")
{
    for (size_t n : golden::k_scan_sizes)
    {
        std::vector<GLuint> data = test_inputs::minstd_vector<GLuint>(123, n, 0, 100);
        ShaderStorageBuffer buffer(data);

        BlellochScan blelloch_scan(DataType_Uint);
        blelloch_scan(buffer.handle(), data.size());

        std::vector<GLuint> expected(n);
        std::exclusive_scan(data.begin(), data.end(), expected.begin(), 0u);
        REQUIRE(buffer.get_data<GLuint>() == expected);
    }
}

TEST_CASE("BlellochScan-multiple-partitions")
{
    const size_t n = 1024;
    for (size_t partitions : golden::k_scan_partition_counts)
    {
        std::vector<GLuint> data = test_inputs::minstd_vector<GLuint>(123, n * partitions, 0, 100);
        ShaderStorageBuffer buffer(data);

        BlellochScan blelloch_scan(DataType_Uint);
        blelloch_scan(buffer.handle(), n, partitions);

        std::vector<GLuint> result = buffer.get_data<GLuint>();
        for (size_t p = 0; p < partitions; p++)
        {
            std::vector<GLuint> expected(n);
            std::exclusive_scan(data.begin() + p * n, data.begin() + (p + 1) * n, expected.begin(), 0u);
            REQUIRE(std::memcmp(expected.data(), result.data() + p * n, n * sizeof(GLuint)) == 0);
        }
    }
}

TEST_CASE("BlellochScan-small-partitions")
{
    // count = 1 and 2 are what RadixSort feeds the scan for tiny inputs in the reference (nbp2 = 1, 2)
    for (size_t n : {1, 2, 4, 64})
    {
        const size_t partitions = 16;
        std::vector<GLuint> data = test_inputs::minstd_vector<GLuint>(3, n * partitions, 0, 1000);
        ShaderStorageBuffer buffer(data);
        BlellochScan blelloch_scan(DataType_Uint);
        blelloch_scan(buffer.handle(), n, partitions);
        std::vector<GLuint> result = buffer.get_data<GLuint>();
        for (size_t p = 0; p < partitions; p++)
        {
            std::vector<GLuint> expected(n);
            std::exclusive_scan(data.begin() + p * n, data.begin() + (p + 1) * n, expected.begin(), 0u);
            CHECK(std::memcmp(expected.data(), result.data() + p * n, n * sizeof(GLuint)) == 0);
        }
    }
}

TEST_CASE("BlellochScan-int-float-double")
{
    const size_t n = 1 << 16;
    std::vector<GLuint> raw = test_inputs::minstd_vector<GLuint>(11, n, 0, 2000);
    {
        std::vector<int32_t> data(n), expected(n);
        for (size_t i = 0; i < n; i++) data[i] = int32_t(raw[i]) - 1000;
        std::exclusive_scan(data.begin(), data.end(), expected.begin(), 0);
        ShaderStorageBuffer buffer(data);
        BlellochScan scan(DataType_Int);
        scan(buffer.handle(), n);
        CHECK(buffer.get_data<int32_t>() == expected);
    }
    {
        std::vector<float> data(n);
        std::vector<double> expected(n);
        for (size_t i = 0; i < n; i++) data[i] = float(raw[i] % 16) * 0.25f; // partial sums < 2^18, exact in float32
        double acc = 0;
        for (size_t i = 0; i < n; i++) { expected[i] = acc; acc += data[i]; }
        ShaderStorageBuffer buffer(data);
        BlellochScan scan(DataType_Float);
        scan(buffer.handle(), n);
        std::vector<float> got = buffer.get_data<float>();
        bool ok = true;
        for (size_t i = 0; i < n; i++) ok = ok && got[i] == float(expected[i]);
        CHECK(ok);
    }
    {
        std::vector<double> data(n), expected(n);
        for (size_t i = 0; i < n; i++) data[i] = double(raw[i]) * 0.125;
        std::exclusive_scan(data.begin(), data.end(), expected.begin(), 0.0);
        ShaderStorageBuffer buffer(data);
        BlellochScan scan(DataType_Double);
        scan(buffer.handle(), n);
        CHECK(buffer.get_data<double>() == expected);
    }
}

int main(int argc, char** argv) { return mini_test::run(argc, argv); }
