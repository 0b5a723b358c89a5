// Input generator of the C++ API tests: same stream as the reference's test generator
// (reference test/util/Random.hpp:15-39: std::minstd_rand, seed 0 = default seed, value = engine() % (max-min) + min),
// so the reference's seeded test inputs are reproduced exactly.
#pragma once

#include <cstdint>
#include <random>
#include <vector>

namespace glu
{
    class Random
    {
    public:
        explicit Random(uint64_t seed = 0) :
            m_engine(seed != 0 ? std::minstd_rand(static_cast<std::minstd_rand::result_type>(seed)) : std::minstd_rand())
        {
        }

        template<typename IntegerT>
        IntegerT sample_int(IntegerT min, IntegerT max)
        {
            return static_cast<IntegerT>(m_engine() % (max - min)) + min;
        }

        template<typename IntegerT>
        std::vector<IntegerT> sample_int_vector(size_t num_elements, IntegerT min, IntegerT max)
        {
            std::vector<IntegerT> out(num_elements);
            for (auto& x : out) x = sample_int(min, max);
            return out;
        }

    private:
        std::minstd_rand m_engine;
    };
} // namespace glu
