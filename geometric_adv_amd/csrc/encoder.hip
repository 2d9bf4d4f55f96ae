// PointNet-style encoder of the victim auto-encoder, forward and backward-to-input, for gfx950.
//
// Reference semantics: src/encoders_decoders.py:37-72 with the widths of src/ae_templates.py:22
// (3 -> 64 -> 128 -> 128 -> 256 -> 128): five per-point layers [conv1d k=1 (= x@W + b),
// batch-norm in inference mode, ReLU], then a max over the points of each cloud.  In the
// reference that is ~25 TF ops per forward; here it is ONE kernel: a workgroup owns 64 points,
// keeps their activations in LDS, chains the four wide layers on v_mfma_f32_32x32x2_f32 (exact
// fp32: the 1e-5 Chamfer tolerance rules out bf16/fp16 operands) and reduces the symmetric
// max-pool in registers.  Nothing but the points, the weights (L2 resident, pre-packed into MFMA
// fragment order) and 3*128 words per tile touches HBM -- the kernel is MFMA bound.
//
// Backward-to-input (weights are frozen, var_list = pert only, adv_ae.py:153): the max-pool
// passes gradient only to the <= 128 "critical" points of a cloud, so the backward kernel
// re-runs the forward for just those rows (keeping the ReLU masks as bytes in LDS) and chains
// the transposed layers.  Exact ties in the pool are handled like TF's _MinOrMaxGrad (equal
// split): a cloud with a tied positive maximum is flagged and processed densely instead.
#include "ae.h"
#include <limits.h>

namespace geoadv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ENC_THREADS = 512;
constexpr int ENC_WAVES = ENC_THREADS / kWave;   // 8
// LDS activation buffers (row stride = width + 4 floats keeps ds_read_b128 conflict-free)
constexpr int BUF_P_FLOATS = ENC_ROWS * (256 + 4);
constexpr int BUF_Q_FLOATS = ENC_ROWS * (128 + 4);
constexpr int MASK_BYTES = ENC_ROWS * (64 + 128 + 128 + 256);   // ReLU masks of h1..h4 (backward only)

// One k-group (8 k values = 4 MFMA k-steps) at a time; RM row blocks of 32 share each B fragment.
template <int RM>
__device__ __forceinline__ void gemm_tile(const float *in, int s_in, int row0, const PackedLayer &L, int cb,
                                          f32x16 (&acc)[RM]) {
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5, i = lane & 31;
    const int kg = L.K >> 3;
    const float4 *bp = reinterpret_cast<const float4 *>(L.w) + (size_t)cb * kg * 64 + lane;
    const float *ar[RM];
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) ar[rm] = in + (row0 + rm * 32 + i) * s_in + 4 * h;
    float4 bcur = bp[0];
    for (int t = 0; t < kg; ++t) {
        const float4 bnext = bp[(size_t)(t + 1 < kg ? t + 1 : t) * 64];
        float4 a[RM];
#pragma unroll
        for (int rm = 0; rm < RM; ++rm) a[rm] = *reinterpret_cast<const float4 *>(ar[rm] + 8 * t);
#pragma unroll
        for (int rm = 0; rm < RM; ++rm) {
            acc[rm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rm].x, bcur.x, acc[rm], 0, 0, 0);
            acc[rm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rm].y, bcur.y, acc[rm], 0, 0, 0);
            acc[rm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rm].z, bcur.z, acc[rm], 0, 0, 0);
            acc[rm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rm].w, bcur.w, acc[rm], 0, 0, 0);
        }
        bcur = bnext;
    }
}

// accumulator register -> row inside a 32-row block (C/D layout of the 32x32 MFMA shapes)
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// Hidden forward layer on a 64-row tile: out = relu(in @ W * scale + shift), optional ReLU mask.
// NOUT == 128: wave w -> column block w&3, row block w>>2.  NOUT == 256: wave w -> column block w,
// both row blocks.
template <int NOUT, bool SAVE_MASK>
__device__ __forceinline__ void fwd_layer(const float *in, int s_in, float *out, int s_out, const PackedLayer &L,
                                          const float *scale, const float *shift, unsigned char *mask) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = lane >> 5, i = lane & 31;
    if (NOUT == 256) {
        const int cb = wave;
        f32x16 acc[2] = {};
        gemm_tile<2>(in, s_in, 0, L, cb, acc);
        const int col = cb * 32 + i;
        const float sc = scale[col], sh = shift[col];
#pragma unroll
        for (int rm = 0; rm < 2; ++rm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rm * 32 + acc_row(r, h);
                const float v = fmaxf(fmaf(acc[rm][r], sc, sh), 0.f);
                out[row * s_out + col] = v;
                if (SAVE_MASK) mask[row * NOUT + col] = v > 0.f;
            }
    } else {
        constexpr int CB = NOUT / 32;             // 4 (128 wide) or 2 (64 wide)
        const int cb = wave % CB, rb = wave / CB;
        if (rb < 2) {
            f32x16 acc[1] = {};
            gemm_tile<1>(in, s_in, rb * 32, L, cb, acc);
            const int col = cb * 32 + i;
            const float sc = scale[col], sh = shift[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb * 32 + acc_row(r, h);
                const float v = fmaxf(fmaf(acc[0][r], sc, sh), 0.f);
                out[row * s_out + col] = v;
                if (SAVE_MASK) mask[row * NOUT + col] = v > 0.f;
            }
        }
    }
}

// Layer 0 (fan-in 3) on the VALU: 512 threads = 64 rows x 8 threads, 8 channels each.
template <bool SAVE_MASK>
__device__ __forceinline__ void fwd_layer0(const float *pts /*LDS [64][3]*/, float *out, int s_out, const DeviceAE &A,
                                           unsigned char *mask) {
    const int row = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 8;
    const float x = pts[row * 3], y = pts[row * 3 + 1], z = pts[row * 3 + 2];
    const int C1 = 64;
#pragma unroll
    for (int c = c0; c < c0 + 8; ++c) {
        float a = x * A.w0[c];
        a = fmaf(y, A.w0[C1 + c], a);
        a = fmaf(z, A.w0[2 * C1 + c], a);
        const float v = fmaxf(fmaf(a, A.scale[0][c], A.shift[0][c]), 0.f);
        out[row * s_out + c] = v;
        if (SAVE_MASK) mask[row * C1 + c] = v > 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// Forward kernel.  grid = (tiles per cloud, clouds).  Outputs per tile and channel: the maximum
// of h5 over the tile's valid rows, the first row attaining it, and how many rows attain it.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ENC_THREADS) void encoder_fwd_kernel(DeviceAE A, int n, const float *x,
                                                                  const float *pert, float *adv_out, float *pmax,
                                                                  int *parg, int *pcnt) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *bufP = lds;
    float *bufQ = lds + BUF_P_FLOATS;
    float *pts = bufQ + BUF_Q_FLOATS;                 // [64][3]
    float *redm = pts + ENC_ROWS * 3;                 // [2][128]
    int *reda = reinterpret_cast<int *>(redm + 256);  // [2][128]
    int *redc = reda + 256;                           // [2][128]

    const int tile = blockIdx.x, b = blockIdx.y, tiles = gridDim.x;
    const int n0 = tile * ENC_ROWS;
    if (threadIdx.x < ENC_ROWS * 3) {
        const int r = threadIdx.x / 3, a = threadIdx.x % 3;
        int p = n0 + r;
        const bool valid = p < n;
        p = valid ? p : n - 1;                        // padding rows repeat the last point (masked below)
        const size_t g = ((size_t)b * n + p) * 3 + a;
        float v = x[g];
        if (pert) v += pert[g];
        pts[threadIdx.x] = v;
        if (adv_out && valid) adv_out[g] = v;
    }
    __syncthreads();
    fwd_layer0<false>(pts, bufQ, 68, A, nullptr);
    __syncthreads();
    fwd_layer<128, false>(bufQ, 68, bufP, 132, A.enc_fwd[1], A.scale[1], A.shift[1], nullptr);
    __syncthreads();
    fwd_layer<128, false>(bufP, 132, bufQ, 132, A.enc_fwd[2], A.scale[2], A.shift[2], nullptr);
    __syncthreads();
    fwd_layer<256, false>(bufQ, 132, bufP, 260, A.enc_fwd[3], A.scale[3], A.shift[3], nullptr);
    __syncthreads();

    // layer 4 + symmetric max-pool straight from the accumulators
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = lane >> 5, i = lane & 31;
    const int cb = wave & 3, rb = wave >> 2;
    f32x16 acc[1] = {};
    gemm_tile<1>(bufP, 260, rb * 32, A.enc_fwd[4], cb, acc);
    const int col = cb * 32 + i;
    const float sc = A.scale[4][col], sh = A.shift[4][col];
    float mx = -1.f;
    int arg = INT_MAX, cnt = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = rb * 32 + acc_row(r, h);
        const float v = fmaxf(fmaf(acc[0][r], sc, sh), 0.f);
        if (n0 + row < n) {
            if (v > mx) { mx = v; arg = n0 + row; cnt = 1; }
            else if (v == mx) cnt++;
        }
    }
    {   // the two lane halves hold interleaved rows of the same column
        const float m2 = __shfl_xor(mx, 32);
        const int a2 = __shfl_xor(arg, 32), c2 = __shfl_xor(cnt, 32);
        if (m2 > mx) { mx = m2; arg = a2; cnt = c2; }
        else if (m2 == mx) { arg = a2 < arg ? a2 : arg; cnt += c2; }
    }
    if (h == 0) { redm[rb * 128 + col] = mx; reda[rb * 128 + col] = arg; redc[rb * 128 + col] = cnt; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int c = threadIdx.x;
        float m = redm[c];
        int a = reda[c], k = redc[c];
        const float m2 = redm[128 + c];
        if (m2 > m) { m = m2; a = reda[128 + c]; k = redc[128 + c]; }
        else if (m2 == m) { k += redc[128 + c]; }       // rows of block 1 are higher: arg stays
        const size_t o = ((size_t)b * tiles + tile) * 128 + c;
        pmax[o] = m; parg[o] = a; pcnt[o] = k;
    }
}

// ------------------------------------------------------------------------------------------
// Backward kernel over a list of rows.  grid = (row tiles, clouds).  rows: [b][rows_per_cloud]
// point indices (duplicates allowed: every listed row is written with the same value).  A cloud
// takes part only if dense_flag[b] == want_dense (the sparse list launch skips flagged clouds,
// the dense launch skips the others).  g_enc[b][row][3] = d loss / d adv through the encoder.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ENC_THREADS) void encoder_bwd_kernel(DeviceAE A, int n, const float *adv,
                                                                  const int *rows, int rows_per_cloud,
                                                                  const float *z, const int *zcnt, const float *dz,
                                                                  const int *dense_flag, int want_dense,
                                                                  float *g_enc) {
    const int b = blockIdx.y;
    if ((dense_flag[b] != 0) != (want_dense != 0)) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *bufP = lds;
    float *bufQ = lds + BUF_P_FLOATS;
    float *pts = bufQ + BUF_Q_FLOATS;                              // [64][3]
    int *rowid = reinterpret_cast<int *>(pts + ENC_ROWS * 3);      // [64]
    unsigned char *m1 = reinterpret_cast<unsigned char *>(rowid + ENC_ROWS);
    unsigned char *m2 = m1 + ENC_ROWS * 64;
    unsigned char *m3 = m2 + ENC_ROWS * 128;
    unsigned char *m4 = m3 + ENC_ROWS * 128;

    const int r0 = blockIdx.x * ENC_ROWS;
    if (threadIdx.x < ENC_ROWS) {
        int rr = r0 + threadIdx.x;
        int p;
        if (rows) { rr = rr < rows_per_cloud ? rr : rows_per_cloud - 1; p = rows[(size_t)b * rows_per_cloud + rr]; }
        else p = rr < n ? rr : n - 1;
        rowid[threadIdx.x] = p;
    }
    __syncthreads();
    if (threadIdx.x < ENC_ROWS * 3) {
        const int r = threadIdx.x / 3, a = threadIdx.x % 3;
        pts[threadIdx.x] = adv[((size_t)b * n + rowid[r]) * 3 + a];
    }
    __syncthreads();
    // forward recompute (bit-identical to the forward kernel: same instruction sequence per row)
    fwd_layer0<true>(pts, bufQ, 68, A, m1);
    __syncthreads();
    fwd_layer<128, true>(bufQ, 68, bufP, 132, A.enc_fwd[1], A.scale[1], A.shift[1], m2);
    __syncthreads();
    fwd_layer<128, true>(bufP, 132, bufQ, 132, A.enc_fwd[2], A.scale[2], A.shift[2], m3);
    __syncthreads();
    fwd_layer<256, true>(bufQ, 132, bufP, 260, A.enc_fwd[3], A.scale[3], A.shift[3], m4);
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = lane >> 5, i = lane & 31;
    {   // layer 4 forward -> da4 = dz/cnt * [h5 == z, z > 0] * scale4   into bufQ (128 wide)
        const int cb = wave & 3, rb = wave >> 2;
        f32x16 acc[1] = {};
        gemm_tile<1>(bufP, 260, rb * 32, A.enc_fwd[4], cb, acc);
        const int col = cb * 32 + i;
        const float sc = A.scale[4][col], sh = A.shift[4][col];
        const float zc = z[(size_t)b * 128 + col];
        const int kc = zcnt[(size_t)b * 128 + col];
        const float gz = (kc > 1 ? (1.0f / (float)kc) : 1.0f) * dz[(size_t)b * 128 + col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rb * 32 + acc_row(r, h);
            const float v = fmaxf(fmaf(acc[0][r], sc, sh), 0.f);
            bufQ[row * 132 + col] = (v == zc && v > 0.f) ? gz * sc : 0.f;
        }
    }
    __syncthreads();
    // dh4 = da4 @ W4^T (128 -> 256); da3 = dh4 * mask4 * scale3   into bufP (256 wide)
    {
        const int cb = wave;
        f32x16 acc[2] = {};
        gemm_tile<2>(bufQ, 132, 0, A.enc_bwd[4], cb, acc);
        const int col = cb * 32 + i;
        const float sc = A.scale[3][col];
#pragma unroll
        for (int rm = 0; rm < 2; ++rm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rm * 32 + acc_row(r, h);
                bufP[row * 260 + col] = m4[row * 256 + col] ? acc[rm][r] * sc : 0.f;
            }
    }
    __syncthreads();
    // dh3 = da3 @ W3^T (256 -> 128); da2 = dh3 * mask3 * scale2   into bufQ
    {
        const int cb = wave & 3, rb = wave >> 2;
        f32x16 acc[1] = {};
        gemm_tile<1>(bufP, 260, rb * 32, A.enc_bwd[3], cb, acc);
        const int col = cb * 32 + i;
        const float sc = A.scale[2][col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rb * 32 + acc_row(r, h);
            bufQ[row * 132 + col] = m3[row * 128 + col] ? acc[0][r] * sc : 0.f;
        }
    }
    __syncthreads();
    // dh2 = da2 @ W2^T (128 -> 128); da1 = dh2 * mask2 * scale1   into bufP (stride 132)
    {
        const int cb = wave & 3, rb = wave >> 2;
        f32x16 acc[1] = {};
        gemm_tile<1>(bufQ, 132, rb * 32, A.enc_bwd[2], cb, acc);
        const int col = cb * 32 + i;
        const float sc = A.scale[1][col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rb * 32 + acc_row(r, h);
            bufP[row * 132 + col] = m2[row * 128 + col] ? acc[0][r] * sc : 0.f;
        }
    }
    __syncthreads();
    // dh1 = da1 @ W1^T (128 -> 64); da0 = dh1 * mask1 * scale0   into bufQ (stride 68); 4 waves
    if (wave < 4) {
        const int cb = wave & 1, rb = wave >> 1;
        f32x16 acc[1] = {};
        gemm_tile<1>(bufP, 132, rb * 32, A.enc_bwd[1], cb, acc);
        const int col = cb * 32 + i;
        const float sc = A.scale[0][col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rb * 32 + acc_row(r, h);
            bufQ[row * 68 + col] = m1[row * 64 + col] ? acc[0][r] * sc : 0.f;
        }
    }
    __syncthreads();
    // dh0 = da0 @ W0^T (64 -> 3) on the VALU
    if (threadIdx.x < ENC_ROWS * 3) {
        const int r = threadIdx.x / 3, a = threadIdx.x % 3;
        float s = 0.f;
#pragma unroll 8
        for (int c = 0; c < 64; ++c) s = fmaf(bufQ[r * 68 + c], A.w0[a * 64 + c], s);
        const bool live = rows ? true : (r0 + r < n);
        if (live) g_enc[((size_t)b * n + rowid[r]) * 3 + a] = s;
    }
}

size_t encoder_fwd_lds_bytes() { return sizeof(float) * (BUF_P_FLOATS + BUF_Q_FLOATS + ENC_ROWS * 3 + 256) + sizeof(int) * 512; }
size_t encoder_bwd_lds_bytes() {
    return sizeof(float) * (BUF_P_FLOATS + BUF_Q_FLOATS + ENC_ROWS * 3) + sizeof(int) * ENC_ROWS + MASK_BYTES;
}

static int set_lds_attr_once() {
    static bool done = false;
    if (done) return GEOADV_OK;
    GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_fwd_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)encoder_fwd_lds_bytes()));
    GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_bwd_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)encoder_bwd_lds_bytes()));
    done = true;
    return GEOADV_OK;
}

int encoder_tiles(int n) { return cdiv(n, ENC_ROWS); }

// pmax/parg/pcnt: [b][tiles][128]
int launch_encoder_fwd(const DeviceAE &A, int b, const float *x, const float *pert, float *adv_out, float *pmax,
                       int *parg, int *pcnt, hipStream_t stream) {
    if (int st = set_lds_attr_once()) return st;
    if (b <= 0) return GEOADV_OK;
    encoder_fwd_kernel<<<dim3(encoder_tiles(A.n_points), b), ENC_THREADS, encoder_fwd_lds_bytes(), stream>>>(
        A, A.n_points, x, pert, adv_out, pmax, parg, pcnt);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

// Sparse pass over the 128 critical rows of every un-flagged cloud, then a dense pass that only
// does work for flagged clouds (its workgroups exit at once otherwise).  g_enc must be zeroed by
// the caller (rows that are not critical keep gradient 0).
int launch_encoder_bwd(const DeviceAE &A, int b, const float *adv, const int *crit_rows, const float *z,
                       const int *zcnt, const float *dz, const int *dense_flag, float *g_enc, hipStream_t stream) {
    if (int st = set_lds_attr_once()) return st;
    if (b <= 0) return GEOADV_OK;
    encoder_bwd_kernel<<<dim3(128 / ENC_ROWS, b), ENC_THREADS, encoder_bwd_lds_bytes(), stream>>>(
        A, A.n_points, adv, crit_rows, 128, z, zcnt, dz, dense_flag, 0, g_enc);
    GA_LAUNCH_CHECK();
    encoder_bwd_kernel<<<dim3(encoder_tiles(A.n_points), b), ENC_THREADS, encoder_bwd_lds_bytes(), stream>>>(
        A, A.n_points, adv, nullptr, 0, z, zcnt, dz, dense_flag, 1, g_enc);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

}  // namespace geoadv
