// Internal helpers shared by the HIP translation units of libgeoadv.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <algorithm>
#include <mutex>
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
constexpr int kCUs = 256;   // MI355X: 8 XCDs x 32 CUs -- launch-SHAPE decisions only (grid sizes, slice widths): nothing is wrong on a smaller
                            // part or partition, just slower.  Anything that needs workgroups RESIDENT together (in-launch flags) asks the
                            // device instead (geoadv_attack::cus)

// One-time kernel attribute setup (hipFuncSetAttribute applies to the CURRENT device only): run(f) calls f once per device
// ordinal, under a lock, so handles created on cuda:1 after cuda:0, or from several host threads, all get their >64 KB LDS opt-in.
struct DeviceOnce {
    std::mutex mu;
    unsigned long long done = 0;
    template <class F> int run(F f) {
        int dev = 0;
        GA_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> g(mu);
        if ((done >> (dev & 63)) & 1ull) return GEOADV_OK;
        if (int rc = f()) return rc;
        done |= 1ull << (dev & 63);
        return GEOADV_OK;
    }
};

#ifdef __HIPCC__
// In-kernel stamps for a DIAGNOSTIC build only (-DGA_STAMPS, tools/debug/build_variants.sh; the shipped library contains none
// of this): thread 0 of a workgroup writes the 100 MHz wall clock (s_memrealtime) at phase boundaries into a per-translation-unit
// array -- kernel slot K, linear block id, stamp index 0..7 -- read back through the getter GA_STAMPS_GETTER defines.
#ifdef GA_STAMPS
#define GA_STAMP_BLOCKS 1024
static __device__ unsigned long long ga_stamps[8 * GA_STAMP_BLOCKS * 8];
#define GA_STAMP(K, I) GA_STAMP_T(K, I, 0)
#define GA_STAMP_T(K, I, TID)                                                                            \
    do {                                                                                                 \
        const unsigned blk_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);            \
        if (threadIdx.x == (TID) && blk_ < GA_STAMP_BLOCKS) {                                                \
            unsigned long long t_;                                                                       \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
            ga_stamps[(((K) & 7) * GA_STAMP_BLOCKS + blk_) * 8 + (I)] = t_;                               \
        }                                                                                                \
    } while (0)
#define GA_STAMPS_GETTER(NAME)                                                                           \
    extern "C" int NAME(unsigned long long *host_out) {                                                  \
        return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(::geoadv::ga_stamps), sizeof(unsigned long long) * 8 * GA_STAMP_BLOCKS * 8) == hipSuccess ? 0 : 1; \
    }
#else
#define GA_STAMP(K, I)
#define GA_STAMP_T(K, I, TID)
#define GA_STAMPS_GETTER(NAME)
#endif

// Sum over the 64 lanes of a wave in a FIXED order (every lane gets the result): four DPP steps fold each row of 16 lanes
// (xor 1, xor 2, half mirror, mirror -- register to register), v_readlane collects the four rows.
__device__ __forceinline__ float wave_sum(float v) {
#define GA_SUM_DPP(CTRL) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false))
    GA_SUM_DPP(0xB1);      // quad_perm [1,0,3,2]
    GA_SUM_DPP(0x4E);      // quad_perm [2,3,0,1]
    GA_SUM_DPP(0x141);     // row_half_mirror
    GA_SUM_DPP(0x140);     // row_mirror
#undef GA_SUM_DPP
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// Lexicographic (value, index) minimum over the 64 lanes of a wave; every lane ends up with the result.  Four DPP steps
// reduce each row of 16 lanes (quad_perm xor 1, xor 2, row_half_mirror, row_mirror -- register-to-register, a few cycles
// each, where a __shfl_xor is a ds_bpermute round trip of ~150), v_readlane collects the four rows.
__device__ __forceinline__ void wave_lexmin(float &d, int &k) {
#define GA_LEXMIN_DPP(CTRL)                                                                              \
    do {                                                                                                 \
        const float d2_ = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), CTRL, 0xf, 0xf, false)); \
        const int k2_ = __builtin_amdgcn_update_dpp(0, k, CTRL, 0xf, 0xf, false);                        \
        if (d2_ < d || (d2_ == d && k2_ < k)) { d = d2_; k = k2_; }                                      \
    } while (0)
    GA_LEXMIN_DPP(0xB1);      // quad_perm [1,0,3,2]
    GA_LEXMIN_DPP(0x4E);      // quad_perm [2,3,0,1]
    GA_LEXMIN_DPP(0x141);     // row_half_mirror
    GA_LEXMIN_DPP(0x140);     // row_mirror
#undef GA_LEXMIN_DPP
    float bd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 0));
    int bk = __builtin_amdgcn_readlane(k, 0);
#pragma unroll
    for (int row = 1; row < 4; ++row) {
        const float d2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 16 * row));
        const int k2 = __builtin_amdgcn_readlane(k, 16 * row);
        if (d2 < bd || (d2 == bd && k2 < bk)) { bd = d2; bk = k2; }
    }
    d = bd;
    k = bk;
}
#endif

}  // namespace geoadv
