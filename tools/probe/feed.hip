// Operand-delivery probe for the fp32 MFMA chains (measurement tooling, NOT part of libgeoadv.so; tools/feed_probe.py).
// The encoder / training layer kernels feed v_mfma_f32_32x32x2_f32 with A fragments from LDS (ds_read_b128) and B fragments
// (packed weights) from L2 through a register ring of buffer loads.  Pure MFMA chains reach 0.985 of the fp32 peak on this part,
// the encoder's loop shape 0.85 (tools/mfma_probe.py): what does each way of delivering the B operand cost, and how much does
// sharing a B fragment between more row blocks (RM) buy?  No epilogues, no barriers, no HBM traffic.
//   mode 0: B by raw buffer loads, ring of DEPTH fragments        mode 1: B by ds_read_b128 from an LDS copy of the weights
//   mode 2: no B loads at all (A from LDS only: the ceiling of this loop shape)
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

template <int MODE, int RM, int DEPTH>
__global__ __launch_bounds__(512) void feed_kernel(float *out, const float *wts, int kgroups, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int ROWS = 64 * RM, STRIDE = 132;
    for (int e = threadIdx.x; e < ROWS * STRIDE; e += 512) lds[e] = 1.0f + 1e-6f * e;
    float *bl = lds + ROWS * STRIDE;                          // mode 1: 4 column blocks x 16 k-groups x 256 floats
    if (MODE == 1)
        for (int e = threadIdx.x; e < 4 * 16 * 256; e += 512) bl[e] = wts[e];
    __syncthreads();
    const float *ar[RM];
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) ar[rm] = lds + (((wave >> 2) * RM + rm) * 32 + (lane & 31)) * STRIDE + 4 * (lane >> 5);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wts), 0, 0x7fffffff, 0x00020000);
    const unsigned lb = lane * 16u;
    const int base = (wave & 3) * 16 * 1024;                  // bytes: this wave's column block
    const float *blw = bl + (wave & 3) * 16 * 256 + lane * 4;
    f16v acc[RM] = {};
    float4 ring[DEPTH];
    auto ldb = [&](int g) -> float4 {
        if (MODE == 0) {
            const u4v t = __builtin_amdgcn_raw_buffer_load_b128(rs, lb, base + g * 1024, 0);
            return make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
        }
        if (MODE == 1) return *reinterpret_cast<const float4 *>(blw + (g & 15) * 256);
        return make_float4(1.f, 2.f, 3.f, 4.f);
    };
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) ring[u] = ldb(u);
    float4 a[RM];
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) a[rm] = *reinterpret_cast<const float4 *>(ar[rm]);
    int g = DEPTH;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            float4 an[RM];
#pragma unroll
            for (int rm = 0; rm < RM; ++rm) an[rm] = *reinterpret_cast<const float4 *>(ar[rm] + 8 * ((u + 1) & 15));
            __builtin_amdgcn_sched_barrier(0);
            const float4 b = ring[u];
#pragma unroll
            for (int rm = 0; rm < RM; ++rm) {
                acc[rm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rm].x, b.x, acc[rm], 0, 0, 0);
                acc[rm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rm].y, b.y, acc[rm], 0, 0, 0);
                acc[rm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rm].z, b.z, acc[rm], 0, 0, 0);
                acc[rm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rm].w, b.w, acc[rm], 0, 0, 0);
            }
            if (MODE != 2) ring[u] = ldb(g);
            g = g + 1 < kgroups ? g + 1 : 0;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rm = 0; rm < RM; ++rm) a[rm] = an[rm];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) s += acc[rm][0] + acc[rm][15];
    if (s == 123.456f) out[0] = s;
}

template <int MODE, int RM, int DEPTH>
static int run(int wgs_per_cu, int iters, float *ms, double *tflops) {
    const int kgroups = 16;                                   // 16 k-groups x 4 column blocks x 1 KiB = 64 KB of weights (L2 / L1)
    float *out = nullptr, *wts = nullptr;
    if (hipMalloc(&out, 64) != hipSuccess || hipMalloc(&wts, 4 * 16 * 1024 + 4096) != hipSuccess) return 1;
    (void)hipMemset(wts, 0, 4 * 16 * 1024 + 4096);
    // LDS per workgroup chosen so that exactly wgs_per_cu fit (160 KB per CU)
    const size_t need = sizeof(float) * (64 * RM * 132 + (MODE == 1 ? 4 * 16 * 256 : 0));
    size_t lds = wgs_per_cu == 1 ? 100 * 1024 : (wgs_per_cu == 2 ? 72 * 1024 : 40 * 1024);
    if (lds < need) { (void)hipFree(out); (void)hipFree(wts); return 2; }
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(feed_kernel<MODE, RM, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 3;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu * 4;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        feed_kernel<MODE, RM, DEPTH><<<grid, 512, lds, 0>>>(out, wts, kgroups, iters);
        (void)hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) return 4;
    }
    (void)hipEventElapsedTime(ms, e0, e1);
    *tflops = (double)grid * 8 * iters * DEPTH * 4 * RM * 4096.0 / (*ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(out); (void)hipFree(wts);
    return 0;
}

// Layer-shaped probe: chains of KG k-groups separated by what a fused MLP layer puts between them -- an epilogue of NV VALU
// instructions per accumulator register on the 16 accumulators, 16 ds_write_b32 of the results into the tile the next chain
// reads, and (BAR) a workgroup barrier.  How much of the matrix pipe does that cost with one and with two workgroups per CU?
template <int KG, int NV, bool BAR>
__global__ __launch_bounds__(512) void feed_layers_kernel(float *out, const float *wts, int chains, int skew) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int STRIDE = 260;
    for (int e = threadIdx.x; e < 64 * STRIDE; e += 512) lds[e] = 1.0f + 1e-6f * e;
    __syncthreads();
    const float *ar = lds + ((wave >> 2) * 32 + (lane & 31)) * STRIDE + 4 * (lane >> 5);
    float *wr = lds + ((wave >> 2) * 32 + 4 * (lane >> 5)) * STRIDE + (wave & 3) * 32 + (lane & 31);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wts), 0, 0x7fffffff, 0x00020000);
    const unsigned lb = lane * 16u;
    const int base = (wave & 3) * 32 * 1024;
    float4 ring[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const u4v t = __builtin_amdgcn_raw_buffer_load_b128(rs, lb, base + u * 1024, 0);
        ring[u] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
    }
    float sc = 1.0001f, sh = 0.001f;
    // skew: start some workgroups half a chain late (which ones share a CU is the question: 1 = odd workgroups, 2 = the second
    // half of the grid, 3 = bit 3 of the index, i.e. every other workgroup of an XCD)
    const bool late = skew == 1 ? (blockIdx.x & 1) : skew == 2 ? (blockIdx.x >= gridDim.x / 2) : skew == 3 ? ((blockIdx.x >> 3) & 1) : false;
    if (late) { __builtin_amdgcn_s_sleep(KG * 2); }          // KG * 2 * 64 cycles = half of the chain's 2 x KG x 4 x 64
    for (int c = 0; c < chains; ++c) {
        f16v acc = {};
        float4 a = *reinterpret_cast<const float4 *>(ar);
#pragma unroll
        for (int t = 0; t < KG; t += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 an = *reinterpret_cast<const float4 *>(ar + 8 * ((t + u + 1) % KG));
                __builtin_amdgcn_sched_barrier(0);
                const float4 b = ring[u];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
                const int g = (t + u + 4) % KG;
                const u4v tt = __builtin_amdgcn_raw_buffer_load_b128(rs, lb, base + g * 1024, 0);
                ring[u] = make_float4(__uint_as_float(tt.x), __uint_as_float(tt.y), __uint_as_float(tt.z), __uint_as_float(tt.w));
                __builtin_amdgcn_sched_barrier(0);
                a = an;
            }
        }
        if (BAR) __syncthreads();                            // every wave has read the tile
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r];
#pragma unroll
            for (int q = 0; q < NV; ++q) v = fmaf(v, sc, sh);
            v = fminf(v, 2.0f);
            wr[((r & 3) + 8 * (r >> 2)) * STRIDE] = v;
        }
        if (BAR) __syncthreads();                            // the next chain's tile is written
    }
    if (lds[threadIdx.x] == 123.456f) out[0] = 1.f;
}

template <int KG, int NV, bool BAR>
static int run_layers(int wgs_per_cu, int chains, int skew, float *ms, double *tflops) {
    float *out = nullptr, *wts = nullptr;
    if (hipMalloc(&out, 64) != hipSuccess || hipMalloc(&wts, 4 * 32 * 1024 + 4096) != hipSuccess) return 1;
    (void)hipMemset(wts, 0, 4 * 32 * 1024 + 4096);
    const size_t lds = wgs_per_cu == 1 ? 100 * 1024 : 72 * 1024;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(feed_layers_kernel<KG, NV, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 3;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;                      // ONE round: the workgroups that start together stay together
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        feed_layers_kernel<KG, NV, BAR><<<grid, 512, lds, 0>>>(out, wts, chains, skew);
        (void)hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) return 4;
    }
    (void)hipEventElapsedTime(ms, e0, e1);
    *tflops = (double)grid * 8 * chains * KG * 4 * 4096.0 / (*ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(out); (void)hipFree(wts);
    return 0;
}

extern "C" int geoadv_probe_feed_layers(int kg, int nv, int bar, int wgs_per_cu, int chains, int skew, float *ms, double *tflops) {
#define LCASE(K, V, B) if (kg == K && nv == V && bar == B) return run_layers<K, V, (B != 0)>(wgs_per_cu, chains, skew, ms, tflops);
    LCASE(16, 0, 0) LCASE(16, 0, 1) LCASE(16, 2, 0) LCASE(16, 2, 1) LCASE(16, 6, 1) LCASE(32, 2, 1) LCASE(8, 2, 1) LCASE(32, 6, 1) LCASE(8, 6, 1)
#undef LCASE
    return 9;
}

extern "C" int geoadv_probe_feed(int mode, int rm, int depth, int wgs_per_cu, int iters, float *ms, double *tflops) {
#define CASE(M, R, D) if (mode == M && rm == R && depth == D) return run<M, R, D>(wgs_per_cu, iters, ms, tflops);
    CASE(0, 1, 4) CASE(0, 2, 4) CASE(0, 4, 4) CASE(0, 1, 8) CASE(0, 2, 8)
    CASE(1, 1, 4) CASE(1, 2, 4) CASE(1, 4, 4)
    CASE(2, 1, 4) CASE(2, 2, 4) CASE(2, 4, 4)
#undef CASE
    return 9;
}
