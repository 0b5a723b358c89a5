// Wall-clock helper + the "0.123 ms" formatter the reference's benchmark lines use
// (reference test/util/StopWatch.hpp:11-59).
#pragma once

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <string>

namespace glu
{
    inline std::string ns_to_human_string(uint64_t ns)
    {
        const double ms = double(ns) / 1.0e6;
        const double s = ms / 1.0e3;
        char buf[64];
        if (s >= 0.1)
            std::snprintf(buf, sizeof(buf), "%.3f s", s);
        else if (ms >= 0.001)
            std::snprintf(buf, sizeof(buf), "%.3f ms", ms);
        else
            std::snprintf(buf, sizeof(buf), "%llu ns", (unsigned long long) ns);
        return buf;
    }

    class StopWatch
    {
        using Clock = std::chrono::steady_clock;

    public:
        StopWatch() { reset(); }
        void reset() { m_start = Clock::now(); }
        uint64_t elapsed_nanos() const
        {
            return (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - m_start).count();
        }
        uint64_t elapsed_millis() const { return elapsed_nanos() / 1000000ull; }
        std::string elapsed_time_str() const { return ns_to_human_string(elapsed_nanos()); }

    private:
        Clock::time_point m_start;
    };
} // namespace glu
