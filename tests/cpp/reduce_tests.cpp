// C++ API tests of glu::Reduce -- the reference's known-answer cases and size sweeps
// (reference test/reduce_tests.cpp:14-183; vectors transcribed as test data, glm types replaced by plain
// structs with the same std430 layout).
#include <numeric>
#include <vector>

#include "glu/Reduce.hpp"
#include "util/Random.hpp"
#include "util/mini_test.hpp"

using namespace glu;

namespace
{
    struct vec2 { float x, y; };
    struct alignas(16) vec4 { float x, y, z, w; };
    struct ivec2 { int32_t x, y; };
    struct alignas(16) ivec4 { int32_t x, y, z, w; };

    const uint32_t k_simple_data[]{32, 35, 1,  3,  95, 10, 22, 24, 44, 37, 7,  80, 33, 54, 46, 23, 14, 84, 11, 67,
                                   4,  58, 70, 61, 16, 36, 83, 9,  56, 99, 28, 98, 69, 21, 51, 34, 48, 91, 62, 19,
                                   59, 79, 39, 92, 97, 78, 52, 40, 66, 47, 89, 88, 74, 49, 31, 20, 45, 13, 26, 72,
                                   43, 30, 65, 94, 63, 8,  60, 15, 93, 86, 41, 75, 12, 73, 55, 90, 64, 96, 53, 1,
                                   57, 71, 50, 42, 29, 2,  77, 25, 82, 18, 81, 85, 27, 5,  6,  68, 17, 38, 87, 76};
    const size_t k_simple_length = sizeof(k_simple_data) / sizeof(k_simple_data[0]);

    uint32_t reduce_simple(ReduceOperator op, size_t count)
    {
        ShaderStorageBuffer buffer(k_simple_data, k_simple_length * sizeof(uint32_t));
        Reduce reduce(DataType_Uint, op);
        reduce(buffer.handle(), count);
        return buffer.get_data<uint32_t>()[0];
    }
} // namespace

TEST_CASE("Reduce-simple-uint")
{
    CHECK(reduce_simple(ReduceOperator_Sum, k_simple_length) == 4951);
    CHECK(reduce_simple(ReduceOperator_Mul, 5) == 319200);
    CHECK(reduce_simple(ReduceOperator_Min, k_simple_length) == 1);
    CHECK(reduce_simple(ReduceOperator_Max, k_simple_length) == 99);
}

TEST_CASE("Reduce-all")
{
    {
        const std::vector<uint32_t> data{1, 11, 80, 73, 48, 40, 89, 36, 70, 57};
        Reduce reduce(DataType_Uint, ReduceOperator_Sum);
        ShaderStorageBuffer buffer(data);
        reduce(buffer.handle(), data.size());
        CHECK(buffer.get_data<uint32_t>()[0] == 505);
    }
    {
        const std::vector<float> data{42.138f, 18.228f, -19.127f, 86.564f, 11.904f, 48.538f, 30.606f, 11.338f, -32.699f, -29.587f};
        Reduce reduce(DataType_Float, ReduceOperator_Sum);
        ShaderStorageBuffer buffer(data);
        reduce(buffer.handle(), data.size());
        CHECK_WITHIN_ABS(buffer.get_data<float>()[0], 167.9f, 0.1f);
    }
    {
        const std::vector<double> data{-6.20, -56.02, 49.42, 52.38, -23.81, -29.72, 95.46, 77.37, -85.00, 81.74};
        Reduce reduce(DataType_Double, ReduceOperator_Sum);
        ShaderStorageBuffer buffer(data);
        reduce(buffer.handle(), data.size());
        CHECK_WITHIN_ABS(buffer.get_data<double>()[0], 155.6, 0.1);
    }
    {
        const std::vector<vec2> data{{-77.08f, 19.54f}, {98.89f, -16.09f},  {10.53f, 91.17f}, {43.06f, -94.18f}, {-19.18f, 0.86f},
                                     {-49.99f, -92.53f}, {-4.68f, 42.34f}, {2.79f, -4.26f},  {-17.49f, 43.99f}, {79.45f, -14.58f}};
        Reduce reduce(DataType_Vec2, ReduceOperator_Sum);
        ShaderStorageBuffer buffer(data);
        reduce(buffer.handle(), data.size());
        vec2 sum = buffer.get_data<vec2>()[0];
        CHECK_WITHIN_ABS(sum.x, 66.29f, 0.1f);
        CHECK_WITHIN_ABS(sum.y, -23.75f, 0.1f);
    }
    {
        const std::vector<vec4> data{{-17.04f, 1.79f, 82.67f, 39.72f},    {52.66f, 24.75f, -19.05f, 91.92f},
                                     {19.15f, 44.93f, -52.13f, 18.85f},   {-84.25f, 69.53f, -11.43f, 33.17f},
                                     {19.46f, -14.30f, -15.20f, -63.83f}, {-20.51f, -56.75f, -2.70f, 82.66f},
                                     {3.86f, 55.48f, -12.37f, -11.02f},   {-30.62f, -67.54f, -29.89f, -77.30f},
                                     {-21.55f, 50.46f, 39.34f, 81.08f},   {-56.40f, 84.61f, 90.26f, 13.35f}};
        Reduce reduce(DataType_Vec4, ReduceOperator_Sum);
        ShaderStorageBuffer buffer(data);
        reduce(buffer.handle(), data.size());
        vec4 sum = buffer.get_data<vec4>()[0];
        CHECK_WITHIN_ABS(sum.x, -135.24f, 0.1f);
        CHECK_WITHIN_ABS(sum.y, 192.97f, 0.1f);
        CHECK_WITHIN_ABS(sum.z, 69.49f, 0.1f);
        CHECK_WITHIN_ABS(sum.w, 208.59f, 0.1f);
    }
    {
        const std::vector<ivec2> data{{-38, -88}, {57, -34}, {61, 60}, {-90, 73}, {-23, -17}, {34, -79}, {-80, 53}, {24, -23}, {-88, 69}, {-83, -67}};
        Reduce reduce(DataType_IVec2, ReduceOperator_Sum);
        ShaderStorageBuffer buffer(data);
        reduce(buffer.handle(), data.size());
        ivec2 sum = buffer.get_data<ivec2>()[0];
        CHECK(sum.x == -226);
        CHECK(sum.y == -53);
    }
    {
        const std::vector<ivec4> data{{-95, 99, -30, 2},   {-69, 33, 78, 20},  {33, -43, -38, -26}, {69, -67, -17, -57}, {18, -23, -2, -53},
                                      {88, -96, 40, -48}, {-93, -47, -91, 59}, {-89, 82, 10, 94},  {-15, 7, 41, 14},    {63, 53, -40, 53}};
        Reduce reduce(DataType_IVec4, ReduceOperator_Sum);
        ShaderStorageBuffer buffer(data);
        reduce(buffer.handle(), data.size());
        ivec4 sum = buffer.get_data<ivec4>()[0];
        CHECK(sum.x == -90);
        CHECK(sum.y == -2);
        CHECK(sum.z == -49);
        CHECK(sum.w == 58);
    }
}

namespace
{
    void run_sum_case(size_t n)
    {
        Random random(1);
        std::vector<GLuint> data = random.sample_int_vector<GLuint>(n, 0, 100);
        GLuint sum = std::accumulate(data.begin(), data.end(), GLuint(0));
        ShaderStorageBuffer buffer(data.data(), data.size() * sizeof(GLuint));
        Reduce reduce(DataType_Uint, ReduceOperator_Sum);
        reduce(buffer.handle(), data.size());
        CHECK(buffer.get_data<GLuint>()[0] == sum);
    }
} // namespace

TEST_CASE("Reduce-subgroup-fitting-size")
{
    for (size_t n : {32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072}) run_sum_case(n);
}

TEST_CASE("Reduce-subgroup-non-fitting-size")
{
    for (size_t n : {1, 31, 93, 201, 693, 2087, 7358, 88289, 345897, 6094798, 5238082, 10043898}) run_sum_case(n);
}

TEST_CASE("Reduce-min-max-mul-other-types")
{
    Random random(21);
    const size_t n = 100003;
    std::vector<GLuint> raw = random.sample_int_vector<GLuint>(n, 0, 100000);
    std::vector<int32_t> idata(n);
    std::vector<float> fdata(n);
    std::vector<double> ddata(n);
    for (size_t i = 0; i < n; i++)
    {
        idata[i] = int32_t(raw[i]) - 50000;
        fdata[i] = float(idata[i]) * 0.5f;
        ddata[i] = double(idata[i]) * 0.25;
    }
    {
        ShaderStorageBuffer b(idata);
        Reduce r(DataType_Int, ReduceOperator_Min);
        r(b.handle(), n);
        CHECK(b.get_data<int32_t>()[0] == *std::min_element(idata.begin(), idata.end()));
    }
    {
        ShaderStorageBuffer b(idata);
        Reduce r(DataType_Int, ReduceOperator_Max);
        r(b.handle(), n);
        CHECK(b.get_data<int32_t>()[0] == *std::max_element(idata.begin(), idata.end()));
    }
    {
        ShaderStorageBuffer b(fdata);
        Reduce r(DataType_Float, ReduceOperator_Min);
        r(b.handle(), n);
        CHECK(b.get_data<float>()[0] == *std::min_element(fdata.begin(), fdata.end()));
    }
    {
        ShaderStorageBuffer b(ddata);
        Reduce r(DataType_Double, ReduceOperator_Max);
        r(b.handle(), n);
        CHECK(b.get_data<double>()[0] == *std::max_element(ddata.begin(), ddata.end()));
    }
    {
        std::vector<GLuint> ones(n, 1u);
        ones[17] = 3;
        ones[n - 1] = 5;
        ones[n / 2] = 7;
        ShaderStorageBuffer b(ones);
        Reduce r(DataType_Uint, ReduceOperator_Mul);
        r(b.handle(), n);
        CHECK(b.get_data<GLuint>()[0] == 105u);
    }
}

int main(int argc, char** argv) { return mini_test::run(argc, argv); }
