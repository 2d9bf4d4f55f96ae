// Roofline calibration micro-benchmarks (not part of the reference's surface): how many fp32
// VALU lane-operations per second the box sustains with plain vs packed instructions.  The
// Chamfer / kNN scans are VALU-issue bound, so DESIGN.md prices them against these numbers.
#include "../../geometric_adv_amd/csrc/common.h"
#include <string.h>

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

// Plain VALU issue rates by opcode (which = 18..27): 16 * iters instructions per lane, eight independent chains.
//   18 v_add_f32  19 v_mul_f32  20 v_sub_f32  21 v_min_u32  22 v_min3_f32  23 v_min3_u32  24 v_max_f32  25 v_cndmask_b32
//   26 v_cmp_lt_f32 (to SGPR pair)  27 v_mov_b32  28 v_cndmask_b32_e64 (SGPR-pair mask)  29 v_writelane_b32  30 v_fma_f32
//   31 v_lshl_add_u32 + v_add_u32
template <int WHICH>
__global__ __launch_bounds__(256) void mb_op_kernel(float *out, int iters) {
    float s = 1.0f + 1e-7f * threadIdx.x;
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 0.5f + i;
    const unsigned long long msk = __ballot((threadIdx.x & 3) == 1) + (unsigned long long)iters;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (WHICH == 18) asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            else if (WHICH == 19) asm volatile("v_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            else if (WHICH == 20) asm volatile("v_sub_f32 %0, %0, %1\n\tv_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            else if (WHICH == 21) asm volatile("v_min_u32 %0, %0, %1\n\tv_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            else if (WHICH == 22) asm volatile("v_min3_f32 %0, %0, %1, %1\n\tv_min3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
            else if (WHICH == 23) asm volatile("v_min3_u32 %0, %0, %1, %1\n\tv_min3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
            else if (WHICH == 24) asm volatile("v_max_f32 %0, %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            else if (WHICH == 25) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s) : "vcc");
            else if (WHICH == 26) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n\tv_cmp_lt_f32 s[22:23], %1, %0" : "+v"(a[i]) : "v"(s) : "s20", "s21", "s22", "s23");
            else if (WHICH == 27) asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %0, %1" : "+v"(a[i]) : "v"(s));
            else if (WHICH == 28) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2\n\tv_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "s"(msk));
            else if (WHICH == 29) asm volatile("v_writelane_b32 %0, %1, 3\n\tv_writelane_b32 %0, %1, 5" : "+v"(a[i]) : "s"((unsigned)msk));
            else if (WHICH == 30) asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
            else asm volatile("v_lshl_add_u32 %0, %0, 1, %1\n\tv_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
        }
    }
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[i];
    if (acc == 123.456f) out[0] = acc;
}

// fp64 / conversion / transcendental issue rates (which = 32..39), same harness:
//   32 v_add_f64  33 v_mul_f64  34 v_fma_f64  35 v_cvt_f64_f32  36 v_cvt_f32_f64  37 v_exp_f32  38 v_ldexp_f32  39 v_rndne_f32
template <int WHICH>
__global__ __launch_bounds__(256) void mb_op64_kernel(float *out, int iters) {
    const double s = 1.0 + 1e-7 * threadIdx.x;
    const float sf = (float)s;
    double a[8];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = 0.5 + i; f[i] = 0.25f + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (WHICH == 32) asm volatile("v_add_f64 %0, %0, %1\n\tv_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            else if (WHICH == 33) asm volatile("v_mul_f64 %0, %0, %1\n\tv_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            else if (WHICH == 34) asm volatile("v_fma_f64 %0, %0, %1, %1\n\tv_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
            else if (WHICH == 35) asm volatile("v_cvt_f64_f32 %0, %1\n\tv_cvt_f64_f32 %0, %1" : "+v"(a[i]) : "v"(f[i]));
            else if (WHICH == 36) asm volatile("v_cvt_f32_f64 %0, %1\n\tv_cvt_f32_f64 %0, %1" : "+v"(f[i]) : "v"(a[i]));
            else if (WHICH == 37) asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %0, %0" : "+v"(f[i]));
            else if (WHICH == 38) asm volatile("v_ldexp_f32 %0, %0, %1\n\tv_ldexp_f32 %0, %0, %1" : "+v"(f[i]) : "v"(1));
            else asm volatile("v_rndne_f32 %0, %0\n\tv_rndne_f32 %0, %0" : "+v"(f[i]));
        }
    }
    double acc = sf;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[i] + f[i];
    if (acc == 123.456) out[0] = (float)acc;
}

// Issue-slot probe (which = 15..17): 16 dependent 32x32x2 MFMAs per iteration plus NV independent v_add_f32 per group of four
// (NV = 4, 8, 16).  If plain VALU instructions of the same or of other waves overlapped with the matrix pipe, the time
// would not move.
template <int NV>
__global__ __launch_bounds__(256) void mb_mix_kernel(float *out, int iters) {
    const float a = 1.0f + 1e-7f * threadIdx.x, b = 0.5f;
    mb16 acc = {};
    float s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = a + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[v]) : "v"(b));
        }
    float r = acc[0] + acc[15];
#pragma unroll
    for (int i = 0; i < 16; ++i) r += s[i];
    if (r == 123.456f) out[0] = r;
}

// Operand-delivery probes (which = 10..12): the encoder's inner loop shape -- per k-group one ds_read_b128 (A fragment, LDS)
// and one 1 KiB global_load_dwordx4 per wave (B fragment, L2-resident weights, 4-deep register ring), then 4 dependent
// v_mfma_f32_32x32x2_f32 -- with the epilogues, barriers and prologue taken away.  512 threads, 72 KB of LDS per
// workgroup (two per CU, 4 waves per SIMD, like encoder_fwd2_kernel).
//   10: MFMAs + LDS reads, 11: MFMAs + global ring, 12: both
template <int WHICH>
__global__ __launch_bounds__(512, 4) void mb_feed_kernel(float *out, const float4 *wts, int wts_groups, int iters) {
    extern __shared__ __attribute__((aligned(16))) float mb_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < 64 * 132; e += 512) mb_lds[e] = 1.0f + 1e-6f * e;
    __syncthreads();
    const float *ar = mb_lds + ((wave >> 2) * 32 + (lane & 31)) * 132 + 4 * (lane >> 5);
    const float4 *bp = wts + lane + (size_t)(wave & 3) * 16 * 64;
    mb16 acc = {};
    float4 ring[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ring[u] = bp[(size_t)u * 64];
    float4 a = *reinterpret_cast<const float4 *>(ar);
    int g = 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float4 an = a;
            if (WHICH != 11) an = *reinterpret_cast<const float4 *>(ar + 8 * ((it * 4 + u + 1) & 15));
            __builtin_amdgcn_sched_barrier(0);
            const float4 b = ring[u];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            if (WHICH != 10) {
                typedef float v4 __attribute__((ext_vector_type(4)));
                const v4 t = *reinterpret_cast<const v4 *>(bp + (size_t)g * 64);
                ring[u] = make_float4(t.x, t.y, t.z, t.w);
                g = g + 1 < wts_groups ? g + 1 : 0;
            }
            __builtin_amdgcn_sched_barrier(0);
            a = an;
        }
    }
    if (acc[0] + acc[15] == 123.456f) out[0] = acc[0];
}

static thread_local char g_probe_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_probe_err, sizeof(g_probe_err), fmt, ap);
    va_end(ap);
}
}  // namespace geoadv
using namespace geoadv;
extern "C" const char *geoadv_probe_last_error(void) { return geoadv::g_probe_err; }


// ms = time of one launch of 2048 workgroups x 256 threads, each thread issuing 16*iters VALU
// instructions of the selected kind.
extern "C" int geoadv_probe_microbench(int which, int iters, float *ms, void *stream) {
    GA_REQUIRE(which >= 0 && which <= 39 && iters > 0 && ms, "microbench: bad arguments");
    hipStream_t st = as_stream(stream);
    float *out = nullptr;
    GA_HIP(hipMalloc(&out, 64));
    const int groups = 352;                                  // 352 KiB of "weights": the encoder's 361 KB, L2-resident (13 / 14: 16 KB / 4 MB)
    float4 *wts = nullptr;
    if (which >= 10) {
        GA_HIP(hipMalloc(&wts, sizeof(float4) * 64 * (size_t)(4096 + 64)));
        GA_HIP(hipMemset(wts, 0, sizeof(float4) * 64 * (size_t)(4096 + 64)));
        GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(mb_feed_kernel<10>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
        GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(mb_feed_kernel<11>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
        GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(mb_feed_kernel<12>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    }
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
            case 9: mb_mfma_kernel<9><<<4096, 256, 0, st>>>(out, iters); break;
            case 10: mb_feed_kernel<10><<<1024, 512, 72 * 1024, st>>>(out, wts, groups, iters); break;
            case 11: mb_feed_kernel<11><<<1024, 512, 72 * 1024, st>>>(out, wts, groups, iters); break;
            case 12: mb_feed_kernel<12><<<1024, 512, 72 * 1024, st>>>(out, wts, groups, iters); break;
            case 13: mb_feed_kernel<11><<<1024, 512, 72 * 1024, st>>>(out, wts, 4, iters); break;        // 16 KB of weights: L1 hits
            case 14: mb_feed_kernel<11><<<1024, 512, 72 * 1024, st>>>(out, wts, 4096, iters); break;     // 4 MB: streams through L2
            case 15: mb_mix_kernel<4><<<4096, 256, 0, st>>>(out, iters); break;
            case 16: mb_mix_kernel<8><<<4096, 256, 0, st>>>(out, iters); break;
            case 17: mb_mix_kernel<16><<<4096, 256, 0, st>>>(out, iters); break;
            case 18: mb_op_kernel<18><<<2048, 256, 0, st>>>(out, iters); break;
            case 19: mb_op_kernel<19><<<2048, 256, 0, st>>>(out, iters); break;
            case 20: mb_op_kernel<20><<<2048, 256, 0, st>>>(out, iters); break;
            case 21: mb_op_kernel<21><<<2048, 256, 0, st>>>(out, iters); break;
            case 22: mb_op_kernel<22><<<2048, 256, 0, st>>>(out, iters); break;
            case 23: mb_op_kernel<23><<<2048, 256, 0, st>>>(out, iters); break;
            case 24: mb_op_kernel<24><<<2048, 256, 0, st>>>(out, iters); break;
            case 25: mb_op_kernel<25><<<2048, 256, 0, st>>>(out, iters); break;
            case 26: mb_op_kernel<26><<<2048, 256, 0, st>>>(out, iters); break;
            case 27: mb_op_kernel<27><<<2048, 256, 0, st>>>(out, iters); break;
            case 28: mb_op_kernel<28><<<2048, 256, 0, st>>>(out, iters); break;
            case 29: mb_op_kernel<29><<<2048, 256, 0, st>>>(out, iters); break;
            case 30: mb_op_kernel<30><<<2048, 256, 0, st>>>(out, iters); break;
            case 31: mb_op_kernel<31><<<2048, 256, 0, st>>>(out, iters); break;
            case 32: mb_op64_kernel<32><<<2048, 256, 0, st>>>(out, iters); break;
            case 33: mb_op64_kernel<33><<<2048, 256, 0, st>>>(out, iters); break;
            case 34: mb_op64_kernel<34><<<2048, 256, 0, st>>>(out, iters); break;
            case 35: mb_op64_kernel<35><<<2048, 256, 0, st>>>(out, iters); break;
            case 36: mb_op64_kernel<36><<<2048, 256, 0, st>>>(out, iters); break;
            case 37: mb_op64_kernel<37><<<2048, 256, 0, st>>>(out, iters); break;
            case 38: mb_op64_kernel<38><<<2048, 256, 0, st>>>(out, iters); break;
            default: mb_op64_kernel<39><<<2048, 256, 0, st>>>(out, iters); break;
        }
        GA_LAUNCH_CHECK();
        GA_HIP(hipEventRecord(e1, st));
        GA_HIP(hipEventSynchronize(e1));
    }
    GA_HIP(hipEventElapsedTime(ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(out);
    if (wts) (void)hipFree(wts);
    return GEOADV_OK;
}
