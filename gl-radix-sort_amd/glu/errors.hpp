// glu/errors.hpp -- the reference's error convention (reference glu/errors.hpp:8-18): a failed check prints
// to stderr and terminates the process with exit(1).  No exceptions, no status returns at this level.
// The C ABI underneath returns status codes; GLU_CHECK_STATUS turns a non-zero status into the same
// print-and-exit behaviour.
#ifndef GLU_ERRORS_HPP
#define GLU_ERRORS_HPP

#include <cstdio>
#include <cstdlib>

#include "glu_hip.h"

#define GLU_CHECK_STATE(condition_, ...)                                                                               \
    do                                                                                                                 \
    {                                                                                                                  \
        if (__builtin_expect(!(condition_), 0))                                                                        \
        {                                                                                                              \
            std::fprintf(stderr, __VA_ARGS__);                                                                         \
            std::fprintf(stderr, "\n");                                                                                \
            std::exit(1);                                                                                              \
        }                                                                                                              \
    } while (0)

#define GLU_CHECK_ARGUMENT(condition_, ...) GLU_CHECK_STATE(condition_, __VA_ARGS__)
#define GLU_FAIL(...) GLU_CHECK_STATE(false, __VA_ARGS__)

/// Wraps a call into libglu_hip.so: any status other than GLU_OK is fatal, as every GL/driver failure is in
/// the reference (e.g. reference glu/gl_utils.hpp:86,131,140).
#define GLU_CHECK_STATUS(call_)                                                                                        \
    do                                                                                                                 \
    {                                                                                                                  \
        glu_status glu_status_ = (call_);                                                                              \
        GLU_CHECK_STATE(glu_status_ == GLU_OK, "%s", glu_last_error());                                                \
    } while (0)

#endif // GLU_ERRORS_HPP
