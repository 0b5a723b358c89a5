// "0.123 ms" formatting of the benchmark lines (same shape as the reference's README table: seconds from 0.1 s up,
// milliseconds down to 1 us, nanoseconds below).
#pragma once

#include <cstdint>
#include <cstdio>
#include <string>

namespace test_timing
{
    inline std::string human_time(uint64_t ns)
    {
        char buf[64];
        if (ns >= 100000000ull)
            std::snprintf(buf, sizeof(buf), "%.3f s", double(ns) * 1e-9);
        else if (ns >= 1000ull)
            std::snprintf(buf, sizeof(buf), "%.3f ms", double(ns) * 1e-6);
        else
            std::snprintf(buf, sizeof(buf), "%llu ns", (unsigned long long) ns);
        return buf;
    }
} // namespace test_timing
