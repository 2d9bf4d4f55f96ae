// FC decoder of the victim auto-encoder (src/encoders_decoders.py:100-132 with the widths of
// src/ae_templates.py:29-33: bneck -> 256 -> 256 -> 3N, ReLU after the first two, no BN), forward
// and backward-to-latent, plus the reduction of the encoder's per-tile max-pool partials.
//
// The first two layers are tiny (98 K MACs per cloud) and run on the VALU, one workgroup per
// cloud.  The last layer (256 x 3N, 6.3 MB of weights at N = 2048) is a skinny GEMM with M = batch:
// it runs on v_mfma_f32_32x32x2_f32 with the weights pre-packed in fragment order, each workgroup
// streaming one 32-column block once (HBM/L2-bandwidth shaped, as SURVEY 8(a5) notes).
#include "ae.h"
#include <limits.h>

namespace geoadv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int acc_row16(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// ------------------------------------------------------------------------------------------
// Pool reduce + FC0 + FC1.  grid = clouds, 256 threads.
//   z[b][c]     = max over tiles (encoders_decoders.py:72: reduce_max over the point axis)
//   crit[b][c]  = lowest point index attaining it; zcnt[b][c] = number of points attaining it
//   dense[b]    = 1 if some channel has a positive maximum attained more than once (exact tie)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void latent_decode_kernel(DeviceAE A, int tiles, const float *pmax, const int *parg,
                                                            const int *pcnt, float *z, int *crit, int *zcnt,
                                                            int *dense, float *d1, float *d2) {
    __shared__ float zs[128];
    __shared__ float hs[256];
    __shared__ int tie;
    const int b = blockIdx.x, t = threadIdx.x;
    if (t == 0) tie = 0;
    __syncthreads();
    if (t < 128) {
        float m = -1.f;
        int a = INT_MAX, k = 0;
        for (int tl = 0; tl < tiles; ++tl) {
            const size_t o = ((size_t)b * tiles + tl) * 128 + t;
            const float pm = pmax[o];
            if (pm > m) { m = pm; a = parg[o]; k = pcnt[o]; }
            else if (pm == m) k += pcnt[o];
        }
        zs[t] = m;
        z[(size_t)b * 128 + t] = m;
        crit[(size_t)b * 128 + t] = a;
        zcnt[(size_t)b * 128 + t] = k;
        if (m > 0.f && k > 1) atomicOr(&tie, 1);
    }
    __syncthreads();
    if (t == 0) dense[b] = tie;
    if (!d1) return;
    {   // FC0 + ReLU: 128 -> 256
        float s = 0.f;
        for (int k = 0; k < 128; ++k) s = fmaf(zs[k], A.v0[k * 256 + t], s);
        s = fmaxf(s + A.c0[t], 0.f);
        hs[t] = s;
        d1[(size_t)b * 256 + t] = s;
    }
    __syncthreads();
    {   // FC1 + ReLU: 256 -> 256
        float s = 0.f;
        for (int k = 0; k < 256; ++k) s = fmaf(hs[k], A.v1[k * 256 + t], s);
        s = fmaxf(s + A.c1[t], 0.f);
        d2[(size_t)b * 256 + t] = s;
    }
}

// ------------------------------------------------------------------------------------------
// FC2 forward: out[b][3N] = d2[b][256] @ V2 + c2.  grid = (column blocks of 32, row blocks of 32),
// 256 threads = 4 waves splitting K = 256 four ways; partials are summed in a fixed order.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decoder_fc2_kernel(DeviceAE A, int batch, const float *d2, float *out) {
    __shared__ float part[3][16][64];
    const int cb = blockIdx.x, rb = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = lane >> 5, i = lane & 31;
    const PackedLayer &L = A.dec2_fwd;
    const int kg = L.K >> 3;                 // 32
    const int per = kg / 4;
    int arow = rb * 32 + i;
    arow = arow < batch ? arow : batch - 1;
    const float *ap = d2 + (size_t)arow * 256 + 4 * h;
    const float4 *bp = reinterpret_cast<const float4 *>(L.w) + (size_t)cb * kg * 64 + lane;
    f32x16 acc = {};
#pragma unroll 2
    for (int t = wave * per; t < (wave + 1) * per; ++t) {
        const float4 a = *reinterpret_cast<const float4 *>(ap + 8 * t);
        const float4 w = bp[(size_t)t * 64];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, acc, 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) part[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
        const int ncols = A.dec_dims[GEOADV_DEC_LAYERS];
        const int col = cb * 32 + i;
        const float bias = col < ncols ? A.c2[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = ((acc[r] + part[0][r][lane]) + part[1][r][lane]) + part[2][r][lane] + bias;
            const int row = rb * 32 + acc_row16(r, h);
            if (row < batch && col < ncols) out[(size_t)row * ncols + col] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// FC2 backward: partial[ch][b][256] = g_out[b][k-chunk] @ V2^T.  grid = (k chunks of 128, row
// blocks of 32), 512 threads = 8 waves = the 8 column blocks of the 256 outputs.
// ------------------------------------------------------------------------------------------
constexpr int DB_KC = 128;

__global__ __launch_bounds__(512) void decoder_fc2_bwd_kernel(DeviceAE A, int batch, const float *g_out, float *partial) {
    __shared__ __attribute__((aligned(16))) float as[32 * (DB_KC + 4)];
    const int ch = blockIdx.x, rb = blockIdx.y;
    const int ncols = A.dec_dims[GEOADV_DEC_LAYERS];   // 3N = K of this product
    const PackedLayer &L = A.dec2_bwd;
    const int kg_total = L.K >> 3;
    const int k0 = ch * DB_KC;
    for (int e = threadIdx.x; e < 32 * DB_KC; e += 512) {
        const int r = e / DB_KC, k = e % DB_KC;
        const int row = rb * 32 + r, kk = k0 + k;
        as[r * (DB_KC + 4) + k] = (row < batch && kk < ncols) ? g_out[(size_t)row * ncols + kk] : 0.f;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = lane >> 5, i = lane & 31;
    const int cb = wave;
    const int t0 = k0 >> 3;
    const int nt = min(DB_KC >> 3, kg_total - t0);
    const float4 *bp = reinterpret_cast<const float4 *>(L.w) + ((size_t)cb * kg_total + t0) * 64 + lane;
    const float *ap = as + i * (DB_KC + 4) + 4 * h;
    f32x16 acc = {};
    for (int t = 0; t < nt; ++t) {
        const float4 a = *reinterpret_cast<const float4 *>(ap + 8 * t);
        const float4 w = bp[(size_t)t * 64];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, acc, 0, 0, 0);
    }
    const int col = cb * 32 + i;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = rb * 32 + acc_row16(r, h);
        if (row < batch) partial[((size_t)ch * batch + row) * 256 + col] = acc[r];
    }
}

// dd2 = sum of partials, masked by d2 > 0; dd1 = dd2 @ V1^T masked by d1 > 0; dz = dd1 @ V0^T.
// grid = clouds, 256 threads.  TF ReluGrad masks by the layer OUTPUT being > 0.
__global__ __launch_bounds__(256) void decoder_bwd_tail_kernel(DeviceAE A, int batch, int chunks, const float *partial,
                                                               const float *d1, const float *d2, float *dz) {
    __shared__ float g2[256];
    __shared__ float g1[256];
    const int b = blockIdx.x, t = threadIdx.x;
    float s = 0.f;
    for (int ch = 0; ch < chunks; ++ch) s += partial[((size_t)ch * batch + b) * 256 + t];
    g2[t] = d2[(size_t)b * 256 + t] > 0.f ? s : 0.f;
    __syncthreads();
    s = 0.f;
    for (int k = 0; k < 256; ++k) s = fmaf(g2[k], A.v1t[k * 256 + t], s);
    g1[t] = d1[(size_t)b * 256 + t] > 0.f ? s : 0.f;
    __syncthreads();
    if (t < 128) {
        s = 0.f;
        for (int k = 0; k < 256; ++k) s = fmaf(g1[k], A.v0t[k * 128 + t], s);
        dz[(size_t)b * 128 + t] = s;
    }
}

int launch_latent_decode(const DeviceAE &A, int b, const float *pmax, const int *parg, const int *pcnt, float *z,
                         int *crit, int *zcnt, int *dense, float *d1, float *d2, hipStream_t stream) {
    if (b <= 0) return GEOADV_OK;
    latent_decode_kernel<<<b, 256, 0, stream>>>(A, cdiv(A.n_points, ENC_ROWS), pmax, parg, pcnt, z, crit, zcnt, dense, d1, d2);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

int launch_decoder_fc2(const DeviceAE &A, int b, const float *d2, float *recon, hipStream_t stream) {
    if (b <= 0) return GEOADV_OK;
    decoder_fc2_kernel<<<dim3(A.dec2_fwd.N / 32, cdiv(b, 32)), 256, 0, stream>>>(A, b, d2, recon);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

int decoder_bwd_chunks(const DeviceAE &A) { return cdiv(A.dec2_bwd.K, DB_KC); }

int launch_decoder_bwd(const DeviceAE &A, int b, const float *g_recon, const float *d1, const float *d2, float *partial,
                       float *dz, hipStream_t stream) {
    if (b <= 0) return GEOADV_OK;
    const int chunks = decoder_bwd_chunks(A);
    decoder_fc2_bwd_kernel<<<dim3(chunks, cdiv(b, 32)), 512, 0, stream>>>(A, b, g_recon, partial);
    GA_LAUNCH_CHECK();
    decoder_bwd_tail_kernel<<<b, 256, 0, stream>>>(A, b, chunks, partial, d1, d2, dz);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

}  // namespace geoadv
