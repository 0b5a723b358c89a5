// C++ API tests of glu::Reduce, data-driven: the known-answer vectors and size ladders of the reference's reduce tests
// live in util/golden_vectors.hpp (generated from tests/golden/reference_vectors.json; reference test/reduce_tests.cpp:14-183).
#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

#include "glu/Reduce.hpp"
#include "util/golden_vectors.hpp"
#include "util/minstd_inputs.hpp"
#include "util/mini_test.hpp"

using namespace glu;

namespace
{
    /// Uploads `values` converted to the scalar type of `data_type`, reduces `count` elements, returns element 0 as doubles.
    template<typename Scalar>
    std::vector<double> reduce_typed(DataType data_type, ReduceOperator op, const std::vector<double>& values, int components)
    {
        std::vector<Scalar> host(values.size());
        for (size_t i = 0; i < values.size(); i++) host[i] = static_cast<Scalar>(values[i]);
        ShaderStorageBuffer buffer(host);
        Reduce reduce(data_type, op);
        reduce(buffer.handle(), values.size() / components);
        std::vector<Scalar> out = buffer.get_data<Scalar>();
        return std::vector<double>(out.begin(), out.begin() + components);
    }

    std::vector<double> reduce_any(int data_type, ReduceOperator op, const std::vector<double>& values, int components)
    {
        const DataType dt = static_cast<DataType>(data_type);
        switch (dt)
        {
        case DataType_Float: case DataType_Vec2: case DataType_Vec4: return reduce_typed<float>(dt, op, values, components);
        case DataType_Double: case DataType_DVec2: case DataType_DVec4: return reduce_typed<double>(dt, op, values, components);
        case DataType_Int: case DataType_IVec2: case DataType_IVec4: return reduce_typed<int32_t>(dt, op, values, components);
        default: return reduce_typed<uint32_t>(dt, op, values, components);
        }
    }

    void check_uint_sum(size_t n)
    {
        std::vector<GLuint> data = test_inputs::minstd_vector<GLuint>(1, n, 0, 100);
        const GLuint expected = std::accumulate(data.begin(), data.end(), GLuint(0));
        ShaderStorageBuffer buffer(data);
        Reduce reduce(DataType_Uint, ReduceOperator_Sum);
        reduce(buffer.handle(), n);
        CHECK(buffer.get_data<GLuint>()[0] == expected);
    }
} // namespace

TEST_CASE("Reduce-simple-uint")
{
    for (const golden::SimpleCase& c : golden::k_reduce_simple_cases)
    {
        ShaderStorageBuffer buffer(golden::k_reduce_simple_input);
        Reduce reduce(DataType_Uint, static_cast<ReduceOperator>(c.op));
        reduce(buffer.handle(), c.count);
        std::vector<uint32_t> out = buffer.get_data<uint32_t>();
        CHECK(out[0] == c.expected);
        // only element 0 is written
        CHECK(std::equal(out.begin() + 1, out.end(), golden::k_reduce_simple_input.begin() + 1));
    }
}

TEST_CASE("Reduce-all")
{
    for (const golden::TypedCase& c : golden::k_reduce_all_cases)
    {
        std::vector<double> got = reduce_any(c.data_type, ReduceOperator_Sum, c.input, c.components);
        for (int k = 0; k < c.components; k++) CHECK_WITHIN_ABS(got[k], c.expected[k], c.abs_tol > 0 ? c.abs_tol : 1e-9);
    }
}

TEST_CASE("Reduce-subgroup-fitting-size")
{
    for (size_t n : golden::k_reduce_fitting_sizes) check_uint_sum(n);
}

TEST_CASE("Reduce-subgroup-non-fitting-size")
{
    for (size_t n : golden::k_reduce_non_fitting_sizes) check_uint_sum(n);
}

TEST_CASE("Reduce-min-max-mul-other-types")
{
    const size_t n = 100003;
    std::vector<GLuint> raw = test_inputs::minstd_vector<GLuint>(21, n, 0, 100000);
    std::vector<double> centred(n);
    for (size_t i = 0; i < n; i++) centred[i] = double(raw[i]) - 50000.0;
    const double lo = *std::min_element(centred.begin(), centred.end()), hi = *std::max_element(centred.begin(), centred.end());
    for (int dt : {int(DataType_Int), int(DataType_Float), int(DataType_Double)})
    {
        CHECK(reduce_any(dt, ReduceOperator_Min, centred, 1)[0] == lo);
        CHECK(reduce_any(dt, ReduceOperator_Max, centred, 1)[0] == hi);
    }
    std::vector<double> ones(n, 1.0);
    ones[17] = 3;
    ones[n / 2] = 7;
    ones[n - 1] = 5;
    CHECK(reduce_any(DataType_Uint, ReduceOperator_Mul, ones, 1)[0] == 105.0);
    // component-wise on a 4-vector type: each component reduces independently
    std::vector<double> quad(4 * 1000);
    for (size_t i = 0; i < quad.size(); i++) quad[i] = double((i % 4 + 1) * (i / 4 % 7));
    std::vector<double> got = reduce_any(DataType_UVec4, ReduceOperator_Max, quad, 4);
    CHECK(got[0] == 6.0);
    CHECK(got[1] == 12.0);
    CHECK(got[2] == 18.0);
    CHECK(got[3] == 24.0);
}

int main(int argc, char** argv) { return mini_test::run(argc, argv); }
