// Seeded test inputs.  The reference's tests draw their inputs from std::minstd_rand (its test/util/Random.hpp:15-39:
// seed 0 means the default seed, value = engine() % (max - min) + min); these free functions reproduce that stream so
// the C++ API tests run on exactly the reference's inputs.
#pragma once

#include <cstdint>
#include <random>
#include <vector>

namespace test_inputs
{
    inline std::minstd_rand make_engine(uint64_t seed)
    {
        return seed == 0 ? std::minstd_rand() : std::minstd_rand(static_cast<std::minstd_rand::result_type>(seed));
    }

    /// `count` draws of engine() % (max - min) + min from a fresh engine seeded with `seed`.
    template<typename IntegerT>
    std::vector<IntegerT> minstd_vector(uint64_t seed, size_t count, IntegerT min, IntegerT max)
    {
        std::minstd_rand engine = make_engine(seed);
        std::vector<IntegerT> out;
        out.reserve(count);
        for (size_t i = 0; i < count; i++) out.push_back(static_cast<IntegerT>(engine() % (max - min)) + min);
        return out;
    }
} // namespace test_inputs
