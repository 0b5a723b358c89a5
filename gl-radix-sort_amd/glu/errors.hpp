// glu/errors.hpp -- failure convention of the glu:: operators.
//
// Kept from the reference (its glu/errors.hpp:8-18): a failed GLU_CHECK_ARGUMENT / GLU_CHECK_STATE / GLU_FAIL prints a
// printf-style message to stderr and ends the process with exit code 1.  There are no exceptions and no status returns
// at this level.  The C ABI below (glu_hip.h) reports failures as status codes + glu_last_error(); GLU_CHECK_STATUS
// converts those into the same print-and-exit behaviour.
#ifndef GLU_ERRORS_HPP
#define GLU_ERRORS_HPP

#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "glu_hip.h"

namespace glu
{
    namespace detail
    {
        /// Prints the formatted message (plus a newline) to stderr and terminates with exit(1).
        [[noreturn]] inline void fatal(const char* format, ...)
        {
            va_list args;
            va_start(args, format);
            std::vfprintf(stderr, format, args);
            va_end(args);
            std::fputc('\n', stderr);
            std::exit(1);
        }

        /// Status check for calls into libglu_hip.so.
        inline void require_ok(glu_status status)
        {
            if (status != GLU_OK) fatal("%s", glu_last_error());
        }
    } // namespace detail
} // namespace glu

#define GLU_FAIL(...) ::glu::detail::fatal(__VA_ARGS__)
#define GLU_CHECK_STATE(condition_, ...) ((condition_) ? (void) 0 : ::glu::detail::fatal(__VA_ARGS__))
#define GLU_CHECK_ARGUMENT(condition_, ...) GLU_CHECK_STATE(condition_, __VA_ARGS__)
#define GLU_CHECK_STATUS(call_) ::glu::detail::require_ok(call_)

#endif // GLU_ERRORS_HPP
