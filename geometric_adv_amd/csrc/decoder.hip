// FC decoder of the victim auto-encoder (src/encoders_decoders.py:100-132 with the widths of
// src/ae_templates.py:29-33: bneck -> 256 -> 256 -> 3N, ReLU after the first two, no BN), forward
// and backward-to-latent, plus the reduction of the encoder's per-tile max-pool partials.
//
// The first two layers are tiny (98 K MACs per cloud) and run on the VALU, one workgroup per
// cloud.  The last layer (256 x 3N, 6.3 MB of weights at N = 2048) is a skinny GEMM with M = batch:
// it runs on v_mfma_f32_32x32x2_f32 with the weights pre-packed in fragment order, each workgroup
// streaming one 32-column block once (HBM/L2-bandwidth shaped, as SURVEY 8(a5) notes).
#include "ae.h"
#include "decoder_tail.h"
#include "chamfer_grid.h"
#include <limits.h>

namespace geoadv {

int encoder_tiles(const DeviceAE &A, int b);

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int acc_row16(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// ------------------------------------------------------------------------------------------
// Pool reduce + FC0 + FC1.  grid = clouds, 1024 threads (the work per cloud is tiny, so it is
// spread over 16 waves and every thread issues all its loads before it starts summing).
//   z[b][c]     = max over tiles (encoders_decoders.py:72: reduce_max over the point axis)
//   crit[b][c]  = lowest point index attaining it; zcnt[b][c] = number of points attaining it
//   dense[b]    = 1 if some channel has a positive maximum attained more than once (exact tie)
// ------------------------------------------------------------------------------------------
// PARTS > 1 (1024-thread form only): PARTS workgroups per cloud.  Each repeats the pool reduce and FC0 (the chain's short,
// latency-bound head) and takes 256 / PARTS output columns of FC1, whose 262 KB of weights are what a cloud's ONE workgroup
// spends most of its time streaming through its CU; part 0 alone writes z / crit / zcnt / dense / d1.  Every output's partial
// sums and their order are unchanged: same bits.
template <int THREADS, int PARTS = 1>
__device__ __forceinline__ void latent_decode_block(const DeviceAE &A, int tiles, const float *pmax, const int *parg,
                                                    const int *pcnt, float *z, int *crit, int *zcnt,
                                                    int *dense, float *d1, float *d2, const int b, const int part_id = 0) {
    __shared__ float gm[8][128];
    __shared__ int ga[8][128], gk[8][128];
    __shared__ float zs[128];
    __shared__ float hs[256];
    __shared__ float part[4][256];
    __shared__ int tie;
    const int t = threadIdx.x;
    GA_STAMP(0, 0);
    if (t == 0) tie = 0;
    // 8 contiguous tile groups x 128 channels; ascending tiles inside a group, groups merged in order.  Every merge is
    // BRANCH-FREE: with `if (pm > m) { a = pa; k = pk; }` the compiler makes the arg / count loads lazy -- one dependent
    // global round trip per tile and array, 8-16 in a row (4.4 us of this kernel's 9.8 at B = 4, in-kernel stamps) -- where
    // selects need every operand, so all loads of a batch are requested together.
    for (int g = t >> 7; g < 8; g += THREADS / 128) {
        const int c = t & 127;
        const int tb = tiles * g / 8, te = tiles * (g + 1) / 8;
        float m = -1.f;
        int a = INT_MAX, k = 0;
        constexpr int U = 8;                                   // tiles per batch: 24 loads in flight per thread
        for (int t0 = tb; t0 < te; t0 += U) {
            float pm[U];
            int pa[U], pk[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int tl = t0 + u < te ? t0 + u : te - 1;  // (a repeated last tile is masked out below)
                const size_t o = ((size_t)b * tiles + tl) * 128 + c;
                pm[u] = pmax[o]; pa[u] = parg[o]; pk[u] = pcnt[o];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool live = t0 + u < te;
                const bool gt = live && pm[u] > m, eq = live && pm[u] == m;
                k = gt ? pk[u] : (eq ? k + pk[u] : k);
                a = gt ? pa[u] : a;
                m = gt ? pm[u] : m;
            }
        }
        gm[g][c] = m; ga[g][c] = a; gk[g][c] = k;
    }
    __syncthreads();
    GA_STAMP(0, 1);
    if (t < 128) {
        float gmv[8];
        int gav[8], gkv[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) { gmv[g] = gm[g][t]; gav[g] = ga[g][t]; gkv[g] = gk[g][t]; }
        float m = gmv[0];
        int a = gav[0], k = gkv[0];
#pragma unroll
        for (int g = 1; g < 8; ++g) {
            const bool gt = gmv[g] > m, eq = gmv[g] == m;
            k = gt ? gkv[g] : (eq ? k + gkv[g] : k);
            a = gt ? gav[g] : a;
            m = gt ? gmv[g] : m;
        }
        zs[t] = m;
        if (part_id == 0) {
            z[(size_t)b * 128 + t] = m;
            crit[(size_t)b * 128 + t] = a;
            zcnt[(size_t)b * 128 + t] = k;
        }
        if (m > 0.f && k > 1) atomicOr(&tie, 1);
    }
    __syncthreads();
    if (t == 0 && part_id == 0) dense[b] = tie;
    GA_STAMP(0, 2);
    if (!d1) return;
    {   // FC0 + ReLU: 128 -> 256
        const float s = fc256_split4<128, THREADS>(zs, A.v0, part);
        if (t < 256) {
            const float v = fmaxf(s + A.c0[t], 0.f);
            hs[t] = v;
            if (part_id == 0) d1[(size_t)b * 256 + t] = v;
        }
    }
    __syncthreads();
    GA_STAMP(0, 3);
    if (PARTS == 1) {   // FC1 + ReLU: 256 -> 256
        const float s = fc256_split4<256, THREADS>(hs, A.v1, part);
        if (t < 256) d2[(size_t)b * 256 + t] = fmaxf(s + A.c1[t], 0.f);
    } else {            // ... this workgroup's 256 / PARTS columns of it
        constexpr int NC = 256 / PARTS;
        const int col0 = part_id * NC;
        const float s = fc256_split4_cols<256, NC>(hs, A.v1, part, col0);
        if (t < NC) d2[(size_t)b * 256 + col0 + t] = fmaxf(s + A.c1[col0 + t], 0.f);
    }
    GA_STAMP(0, 7);
}

// AutoEncoder.decode (src/autoencoder.py:191-194): the decoder fed with a GIVEN latent code -- FC0 + FC1 exactly as in
// latent_decode_block (same split, same order of the partial sums: decode(transform(x)) equals reconstruct(x) bit for bit).
__global__ __launch_bounds__(LD_THREADS) void latent_fc_kernel(DeviceAE A, const float *z, float *d2) {
    __shared__ float zs[128];
    __shared__ float hs[256];
    __shared__ float part[4][256];
    const int t = threadIdx.x, b = blockIdx.x;
    if (t < 128) zs[t] = z[(size_t)b * 128 + t];
    __syncthreads();
    {
        const float s = fc256_split4<128, LD_THREADS>(zs, A.v0, part);
        if (t < 256) hs[t] = fmaxf(s + A.c0[t], 0.f);
    }
    __syncthreads();
    {
        const float s = fc256_split4<256, LD_THREADS>(hs, A.v1, part);
        if (t < 256) d2[(size_t)b * 256 + t] = fmaxf(s + A.c1[t], 0.f);
    }
}

template <int PARTS>
__global__ __launch_bounds__(LD_THREADS) void latent_decode_kernel(DeviceAE A, int tiles, const float *pmax, const int *parg,
                                                                   const int *pcnt, float *z, int *crit, int *zcnt,
                                                                   int *dense, float *d1, float *d2) {
    latent_decode_block<LD_THREADS, PARTS>(A, tiles, pmax, parg, pcnt, z, crit, zcnt, dense, d1, d2, blockIdx.x / PARTS, blockIdx.x % PARTS);
}

// The same launch with the workgroups of the attack's paired grid search nn_distance(adv, x) behind it (chamfer_grid.h):
// that search needs nothing the network produces, and this launch keeps 32 workgroups busy for 9 us.  Two shapes.
// THREADS = 1024 (the latent blocks' own shape; the grid blocks use the first eight of the sixteen waves): 16 waves x 112
// VGPRs admit ONE workgroup per CU, fine while the 9 B workgroups of the launch fit the 256 CUs (B <= 28).  THREADS = 512:
// two workgroups per CU, so a B = 32 batch (288 workgroups) still runs in one round -- with 1024 it took two (21 us instead
// of 15) --, and the latent blocks do their two K quarters in sequence (13 instead of 10 us, which is why small batches keep
// the first shape).  Same arithmetic in the same order either way.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void latent_decode_and_grid_kernel(DeviceAE A, int tiles, const float *pmax, const int *parg,
                                                                         const int *pcnt, float *z, int *crit, int *zcnt, int *dense,
                                                                         float *d1, float *d2, int batch, GridArgs G) {
    if ((int)blockIdx.x < batch) {
        latent_decode_block<THREADS>(A, tiles, pmax, parg, pcnt, z, crit, zcnt, dense, d1, d2, blockIdx.x);
        return;
    }
    if (THREADS > GR_THREADS && threadIdx.x >= GR_THREADS) return;
    const int g = blockIdx.x - batch;                   // (cloud, direction, slice), slice fastest
    GA_STAMP(1, 0);
    grid_nn_block<GR_MAX_N>(G, g / (2 * GR_QSPLIT), (g / GR_QSPLIT) % 2, g % GR_QSPLIT);
    GA_STAMP(1, 7);
}

// ------------------------------------------------------------------------------------------
// FC2 forward: out[b][3N] = d2[b][256] @ V2 + c2.  grid = (column blocks of 32, row blocks of 32),
// 256 threads = 4 waves splitting K = 256 four ways; partials are summed in a fixed order.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void decoder_fc2_block(const DeviceAE &A, int batch, const float *d2, float *out, const int cb, const int rb) {
    __shared__ float part[3][16][64];
    GA_STAMP(2, 0);
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
    // K = 256 = 32 k-groups, 8 per wave: request all 16 operand fragments first, then run the MFMAs
    float4 av[8], wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int t = wave * per + (u < per ? u : per - 1);
        av[u] = *reinterpret_cast<const float4 *>(ap + 8 * t);
        wv[u] = bp[(size_t)t * 64];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (u < per) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, wv[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, wv[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].z, wv[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].w, wv[u].w, acc, 0, 0, 0);
        }
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
    GA_STAMP(2, 7);
}

// fill / fill_count: words this launch also sets to all ones on its way (the symmetric Chamfer scan's packed row minima, which the
// scan of the SAME forward folds into with 64-bit atomic minima: chamfer_sym.hip) -- no launch of their own
__global__ __launch_bounds__(256) void decoder_fc2_kernel(DeviceAE A, int batch, const float *d2, float *out, unsigned long long *fill,
                                                         size_t fill_count) {
    if (fill) {
        const size_t stride = (size_t)gridDim.x * gridDim.y * 256;
        for (size_t e = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; e < fill_count; e += stride) fill[e] = ~0ull;
    }
    decoder_fc2_block(A, batch, d2, out, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------------
// FC2 backward: partial[ch][b][256] = g_out[b][k-chunk] @ V2^T.  grid = (k chunks of 128, row
// blocks of 32), 512 threads = 8 waves = the 8 column blocks of the 256 outputs.
// ------------------------------------------------------------------------------------------
constexpr int DB_KC = 64;

__global__ __launch_bounds__(512) void decoder_fc2_bwd_kernel(DeviceAE A, int batch, const float *g_out, float *partial) {
    __shared__ __attribute__((aligned(16))) float as[32 * (DB_KC + 4)];
    GA_STAMP(3, 0);
    const int ch = blockIdx.x, rb = blockIdx.y;
    const int ncols = A.dec_dims[GEOADV_DEC_LAYERS];   // 3N = K of this product
    const PackedLayer &L = A.dec2_bwd;
    const int kg_total = L.K >> 3;
    const int k0 = ch * DB_KC;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = lane >> 5, i = lane & 31;
    const int cb = wave;
    const int t0 = k0 >> 3;
    const int nt = min(DB_KC >> 3, kg_total - t0);
    const float4 *bp = reinterpret_cast<const float4 *>(L.w) + ((size_t)cb * kg_total + t0) * 64 + lane;
    constexpr int NT = DB_KC >> 3;              // 8 k-groups: all B fragments requested up front, BEFORE the A tile is
    float4 wv[NT];                              // staged (they do not depend on it: one global round trip instead of two)
#pragma unroll
    for (int t = 0; t < NT; ++t) wv[t] = bp[(size_t)(t < nt ? t : nt - 1) * 64];
    for (int e = threadIdx.x; e < 32 * DB_KC; e += 512) {
        const int r = e / DB_KC, k = e % DB_KC;
        const int row = rb * 32 + r, kk = k0 + k;
        as[r * (DB_KC + 4) + k] = (row < batch && kk < ncols) ? g_out[(size_t)row * ncols + kk] : 0.f;
    }
    __syncthreads();
    const float *ap = as + i * (DB_KC + 4) + 4 * h;
    f32x16 acc = {};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < nt) {
            const float4 a = *reinterpret_cast<const float4 *>(ap + 8 * t);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wv[t].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wv[t].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wv[t].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wv[t].w, acc, 0, 0, 0);
        }
    }
    const int col = cb * 32 + i;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = rb * 32 + acc_row16(r, h);
        if (row < batch) partial[((size_t)ch * batch + row) * 256 + col] = acc[r];
    }
    GA_STAMP(3, 7);
}

// the decoder backward's tail (decoder_tail.h) as a launch of its own: grid = clouds, 1024 threads
__global__ __launch_bounds__(LD_THREADS) void decoder_bwd_tail_kernel(DeviceAE A, int batch, int chunks, const float *partial,
                                                                     const float *d1, const float *d2, float *dz, JacApply ja) {
    decoder_bwd_tail_body(A, batch, chunks, partial, d1, d2, dz, ja, blockIdx.x, nullptr, 0u);
}

int launch_latent_decode(const DeviceAE &A, int b, const float *pmax, const int *parg, const int *pcnt, float *z,
                         int *crit, int *zcnt, int *dense, float *d1, float *d2, hipStream_t stream) {
    if (b <= 0) return GEOADV_OK;
    // several workgroups per cloud while the launch still fits the chip once (B <= 64: four, B <= 128: two): latent_decode_block
    const int tiles = encoder_tiles(A, b);
    if (d1 && 4 * b <= kCUs) latent_decode_kernel<4><<<4 * b, LD_THREADS, 0, stream>>>(A, tiles, pmax, parg, pcnt, z, crit, zcnt, dense, d1, d2);
    else if (d1 && 2 * b <= kCUs) latent_decode_kernel<2><<<2 * b, LD_THREADS, 0, stream>>>(A, tiles, pmax, parg, pcnt, z, crit, zcnt, dense, d1, d2);
    else latent_decode_kernel<1><<<b, LD_THREADS, 0, stream>>>(A, tiles, pmax, parg, pcnt, z, crit, zcnt, dense, d1, d2);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

// latent_decode + the paired grid search of (P, Q) -> (gd1, gi1, gd2, gi2) with give-up flags `need` (chamfer_grid.hip)
int launch_latent_decode_and_grid(const DeviceAE &A, int b, const float *pmax, const int *parg, const int *pcnt, float *z, int *crit,
                                  int *zcnt, int *dense, float *d1, float *d2, const float *P, const float *Q, float *gd1, int *gi1,
                                  float *gd2, int *gi2, int n, int *need, int call, const float *box, hipStream_t stream) {
    if (b <= 0) return GEOADV_OK;
    static DeviceOnce attr;
    if (int rc = attr.run([]() -> int {
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(latent_decode_and_grid_kernel<LD_THREADS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)chamfer_grid_lds_bytes(GR_MAX_N)));
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(latent_decode_and_grid_kernel<GR_THREADS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)chamfer_grid_lds_bytes(GR_MAX_N)));
            return GEOADV_OK;
        })) return rc;
    const GridArgs G{P, Q, gd1, gi1, gd2, gi2, n, need, nullptr, call, box};
    const int blocks = b + b * 2 * GR_QSPLIT;
    if (blocks <= kCUs)     // one workgroup per CU is enough: keep the latent blocks at their 16 waves
        latent_decode_and_grid_kernel<LD_THREADS><<<blocks, LD_THREADS, chamfer_grid_lds_bytes(n), stream>>>(
            A, encoder_tiles(A, b), pmax, parg, pcnt, z, crit, zcnt, dense, d1, d2, b, G);
    else
        latent_decode_and_grid_kernel<GR_THREADS><<<blocks, GR_THREADS, chamfer_grid_lds_bytes(n), stream>>>(
            A, encoder_tiles(A, b), pmax, parg, pcnt, z, crit, zcnt, dense, d1, d2, b, G);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

int launch_latent_fc(const DeviceAE &A, int b, const float *z, float *d2, hipStream_t stream) {
    if (b <= 0) return GEOADV_OK;
    latent_fc_kernel<<<b, LD_THREADS, 0, stream>>>(A, z, d2);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

int launch_decoder_fc2(const DeviceAE &A, int b, const float *d2, float *recon, hipStream_t stream, unsigned long long *fill, size_t fill_count) {
    if (b <= 0) return GEOADV_OK;
    decoder_fc2_kernel<<<dim3(A.dec2_fwd.N / 32, cdiv(b, 32)), 256, 0, stream>>>(A, b, d2, recon, fill, fill_count);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

int decoder_bwd_chunks(const DeviceAE &A) { return cdiv(A.dec2_bwd.K, DB_KC); }

// the first of the decoder backward's two launches alone (the second then comes from encoder.hip: launch_decoder_tail_dense)
int launch_decoder_fc2_bwd(const DeviceAE &A, int b, const float *g_recon, float *partial, hipStream_t stream) {
    if (b <= 0) return GEOADV_OK;
    decoder_fc2_bwd_kernel<<<dim3(decoder_bwd_chunks(A), cdiv(b, 32)), 512, 0, stream>>>(A, b, g_recon, partial);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

int launch_decoder_bwd(const DeviceAE &A, int b, const float *g_recon, const float *d1, const float *d2, float *partial,
                       float *dz, hipStream_t stream, const int *crit, const float *jac, const int *dense, float *g_enc) {
    if (b <= 0) return GEOADV_OK;
    const int chunks = decoder_bwd_chunks(A);
    decoder_fc2_bwd_kernel<<<dim3(chunks, cdiv(b, 32)), 512, 0, stream>>>(A, b, g_recon, partial);
    GA_LAUNCH_CHECK();
    decoder_bwd_tail_kernel<<<b, LD_THREADS, 0, stream>>>(A, b, chunks, partial, d1, d2, dz, JacApply{crit, jac, dense, g_enc, A.n_points});
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

}  // namespace geoadv
GA_STAMPS_GETTER(geoadv_debug_stamps_decoder)
