// Roofline calibration micro-benchmarks (not part of the reference's surface): how many fp32
// VALU lane-operations per second the box sustains with plain vs packed instructions.  The
// Chamfer / kNN scans are VALU-issue bound, so DESIGN.md prices them against these numbers.
#include "common.h"

namespace geoadv {

typedef float f2 __attribute__((ext_vector_type(2)));

// which: 0 = v_mul_f32 + v_add_f32 (independent chains), 1 = v_pk_mul_f32 + v_pk_add_f32,
//        2 = v_min_f32, 3 = v_fma_f32, 4 = v_pk_fma_f32
template <int WHICH>
__global__ __launch_bounds__(256) void mb_valu_kernel(float *out, int iters) {
    float s = 1.0f + 1e-7f * threadIdx.x;
    float a[8];
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = 0.5f + i; p[i] = f2{0.5f + i, 1.5f + i}; }
    f2 s2 = {s, s};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (WHICH == 0) {
                asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            } else if (WHICH == 1) {
                asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
            } else if (WHICH == 2) {
                asm volatile("v_min_f32 %0, %0, %1\n\tv_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            } else if (WHICH == 3) {
                asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
            } else {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(s2));
            }
        }
    }
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[i] + p[i].x + p[i].y;
    if (acc == 123.456f) out[0] = acc;   // keep everything live
}

// MFMA issue-rate probes (which = 5..9): every wave issues 16 * iters MFMAs with no memory traffic at all.
//   5: v_mfma_f32_32x32x2_f32, ONE accumulator (every MFMA depends on the previous one -- the encoder's chain shape)
//   6: the same with 2 independent accumulators, 7: with 4
//   8: v_mfma_f32_16x16x4_f32, one accumulator;  9: with 4 independent accumulators
// Launched with 4096 workgroups x 256 threads (waves per SIMD limited only by registers).
typedef float mb16 __attribute__((ext_vector_type(16)));
typedef float mb4 __attribute__((ext_vector_type(4)));
template <int WHICH>
__global__ __launch_bounds__(256) void mb_mfma_kernel(float *out, int iters) {
    const float a = 1.0f + 1e-7f * threadIdx.x, b = 0.5f;
    float r = 0.f;
    if (WHICH <= 7) {
        constexpr int NA = WHICH == 5 ? 1 : (WHICH == 6 ? 2 : 4);
        mb16 acc[NA] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u % NA] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % NA], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NA; ++q) r += acc[q][0] + acc[q][15];
    } else {
        constexpr int NA = WHICH == 8 ? 1 : 4;
        mb4 acc[NA] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u % NA] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % NA], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NA; ++q) r += acc[q][0] + acc[q][3];
    }
    if (r == 123.456f) out[0] = r;
}

}  // namespace geoadv
using namespace geoadv;

// ms = time of one launch of 2048 workgroups x 256 threads, each thread issuing 16*iters VALU
// instructions of the selected kind.
extern "C" int geoadv_microbench(int which, int iters, float *ms, void *stream) {
    GA_REQUIRE(which >= 0 && which <= 9 && iters > 0 && ms, "microbench: bad arguments");
    hipStream_t st = as_stream(stream);
    float *out = nullptr;
    GA_HIP(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    GA_HIP(hipEventCreate(&e0));
    GA_HIP(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {   // first launch warms up
        GA_HIP(hipEventRecord(e0, st));
        switch (which) {
            case 0: mb_valu_kernel<0><<<2048, 256, 0, st>>>(out, iters); break;
            case 1: mb_valu_kernel<1><<<2048, 256, 0, st>>>(out, iters); break;
            case 2: mb_valu_kernel<2><<<2048, 256, 0, st>>>(out, iters); break;
            case 3: mb_valu_kernel<3><<<2048, 256, 0, st>>>(out, iters); break;
            case 4: mb_valu_kernel<4><<<2048, 256, 0, st>>>(out, iters); break;
            case 5: mb_mfma_kernel<5><<<4096, 256, 0, st>>>(out, iters); break;
            case 6: mb_mfma_kernel<6><<<4096, 256, 0, st>>>(out, iters); break;
            case 7: mb_mfma_kernel<7><<<4096, 256, 0, st>>>(out, iters); break;
            case 8: mb_mfma_kernel<8><<<4096, 256, 0, st>>>(out, iters); break;
            default: mb_mfma_kernel<9><<<4096, 256, 0, st>>>(out, iters); break;
        }
        GA_LAUNCH_CHECK();
        GA_HIP(hipEventRecord(e1, st));
        GA_HIP(hipEventSynchronize(e1));
    }
    GA_HIP(hipEventElapsedTime(ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(out);
    return GEOADV_OK;
}
