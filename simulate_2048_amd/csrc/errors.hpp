// errors.hpp -- error reporting and argument checks of the C-ABI (include/env2048.h); plain C++, no HIP types.
#pragma once
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "env2048.h"
#include "tiling.hpp"

namespace env2048 {
namespace host {

// what env2048_last_error() returns: per thread, overwritten by the next failing call of that thread
inline thread_local char g_error[512] = "";

inline int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

inline bool misaligned(const void *p, uintptr_t a) { return p && (reinterpret_cast<uintptr_t>(p) & (a - 1)) != 0; }

inline int check_range(uint64_t n, uint64_t env_base)
{
    // env_hi is folded into counter word 3 as 2*env_hi + stream: keep env below 2^63.
    if (env_base + n < env_base || (env_base + n) >> 63)
        return fail(ENV2048_ERR_RANGE, "env_base + n = %llu + %llu exceeds the 2^63 environment index range",
                    (unsigned long long)env_base, (unsigned long long)n);
    return ENV2048_OK;
}

inline int check_count(uint64_t n)
{
    if (n > ((uint64_t)0x7FFFFFFF) * BLOCK)
        return fail(ENV2048_ERR_RANGE, "n = %llu exceeds the largest single launch (%llu boards)",
                    (unsigned long long)n, (unsigned long long)0x7FFFFFFF * BLOCK);
    return ENV2048_OK;
}

}  // namespace host
}  // namespace env2048
