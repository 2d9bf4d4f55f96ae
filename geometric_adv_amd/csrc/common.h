// Internal helpers shared by the HIP translation units of libgeoadv.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/geoadv.h"

namespace geoadv {

void set_error(const char *fmt, ...);

#define GA_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) {                                          \
            ::geoadv::set_error(__VA_ARGS__);                   \
            return GEOADV_EINVAL;                               \
        }                                                       \
    } while (0)

#define GA_HIP(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            ::geoadv::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return GEOADV_EHIP;                                                                \
        }                                                                                      \
    } while (0)

#define GA_LAUNCH_CHECK()  GA_HIP(hipGetLastError())

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

constexpr int kWave = 64;   // gfx950 wavefront

}  // namespace geoadv
