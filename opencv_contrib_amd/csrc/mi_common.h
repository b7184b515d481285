// Internal helpers shared by the miflow translation units (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "miflow/c_api.h"

namespace mi {

void set_error(const char *fmt, ...);

#define MI_HIP_TRY(expr)                                                                  \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            mi::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return _e == hipErrorOutOfMemory ? MI_ERR_OOM : MI_ERR_HIP;                   \
        }                                                                                 \
    } while (0)

#define MI_REQUIRE(cond, code, ...)        \
    do {                                   \
        if (!(cond)) {                     \
            mi::set_error(__VA_ARGS__);    \
            return (code);                 \
        }                                  \
    } while (0)

static inline int div_up(int a, int b) { return (a + b - 1) / b; }
static inline int align_up(int a, int b) { return div_up(a, b) * b; }

// Dense-row float plane in scratch memory: row pitch `ld` floats (multiple of 64 = 256 B),
// base 256-B aligned, so every row start is dwordx4-aligned.
struct PlaneF {
    float *p;
    int ld;
};

}  // namespace mi
