// Minimal test harness for the C++ API tests (Catch2 is not available in this image).
// TEST_CASE(name) registers a function; CHECK / REQUIRE count failures; main() runs everything whose name
// contains argv[1] (all when absent) and returns non-zero on any failure.
#pragma once

#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

namespace mini_test
{
    struct Case
    {
        std::string name;
        std::function<void()> body;
    };
    inline std::vector<Case>& registry()
    {
        static std::vector<Case> r;
        return r;
    }
    inline int& failures()
    {
        static int f = 0;
        return f;
    }
    inline int& checks()
    {
        static int c = 0;
        return c;
    }
    struct Registrar
    {
        Registrar(const char* name, std::function<void()> body) { registry().push_back({name, std::move(body)}); }
    };
    struct RequireFailed
    {
    };
    inline int run(int argc, char** argv)
    {
        const char* filter = argc > 1 ? argv[1] : "";
        int ran = 0;
        for (auto& c : registry())
        {
            if (*filter && c.name.find(filter) == std::string::npos) continue;
            int before = failures();
            try
            {
                c.body();
            }
            catch (const RequireFailed&)
            {
            }
            std::printf("[%s] %s\n", failures() == before ? " OK " : "FAIL", c.name.c_str());
            std::fflush(stdout);
            ran++;
        }
        std::printf("%d test case(s), %d check(s), %d failure(s)\n", ran, checks(), failures());
        return failures() == 0 && ran > 0 ? 0 : 1;
    }
} // namespace mini_test

#define MT_CAT2(a, b) a##b
#define MT_CAT(a, b) MT_CAT2(a, b)
#define TEST_CASE(name)                                                                                                \
    static void MT_CAT(mt_case_, __LINE__)();                                                                          \
    static mini_test::Registrar MT_CAT(mt_reg_, __LINE__)(name, MT_CAT(mt_case_, __LINE__));                           \
    static void MT_CAT(mt_case_, __LINE__)()

#define CHECK(cond)                                                                                                    \
    do                                                                                                                 \
    {                                                                                                                  \
        mini_test::checks()++;                                                                                         \
        if (!(cond))                                                                                                   \
        {                                                                                                              \
            mini_test::failures()++;                                                                                   \
            std::printf("  %s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #cond);                                     \
        }                                                                                                              \
    } while (0)

#define REQUIRE(cond)                                                                                                  \
    do                                                                                                                 \
    {                                                                                                                  \
        mini_test::checks()++;                                                                                         \
        if (!(cond))                                                                                                   \
        {                                                                                                              \
            mini_test::failures()++;                                                                                   \
            std::printf("  %s:%d: REQUIRE(%s) failed\n", __FILE__, __LINE__, #cond);                                   \
            throw mini_test::RequireFailed();                                                                          \
        }                                                                                                              \
    } while (0)

#define CHECK_WITHIN_ABS(value, target, margin) CHECK(std::fabs(double(value) - double(target)) <= double(margin))
