// One TRAINING step of the victim auto-encoder on gfx950.
//
// Reference: PointNetAutoEncoder (src/pointnet_ae.py:71-99: loss = reduce_mean(dist1) + reduce_mean(dist2) of
// nn_distance(x_reconstr, gt), AdamOptimizer(lr).minimize(loss)) driven by AutoEncoder.partial_fit
// (src/autoencoder.py:105-125, tflearn is_training(True)), architecture src/ae_templates.py:22-33, defaults
// default_train_params (:43-51: batch 50, lr 0.0005).  In training mode tflearn's batch_normalization uses the
// statistics of the batch (tf.nn.moments over all B*N rows) and differentiates through them, so -- unlike the
// attack path -- a layer cannot start before the previous one has finished on EVERY row.  The step is therefore
// layer-by-layer with the pre-BN activations a_i kept in HBM (288 MB at B = 50; the card has 288 GB), and every
// kernel fuses what the dependency structure allows:
//   forward layer i   : h_i = relu(a_{i-1} * s + t) formed while loading the tile, [64 x C_i] @ W_i on
//                       v_mfma_f32_32x32x2_f32, a_i stored, per-tile (sum a, sum a^2) for the batch statistics;
//   backward layer i  : da_i from (dy_i, a_i) on load; dW_i += h_i^T @ da_i accumulated in registers across the
//                       tiles of a persistent workgroup (MFMA, both operands from LDS); dy_{i-1} =
//                       (da_i @ W_i^T) * [h_i > 0] (MFMA, packed W_i^T) with the per-tile sums the BN backward
//                       of layer i-1 needs, all in ONE kernel per layer.
// All reductions run in a fixed order (per-tile partials, then a double-precision pass), so a step is
// deterministic.  Parity is checked by tests/test_gpu_train.py against a numpy fp64 model of the same step.
#include "chamfer_sym.h"
#include "ae.h"
#include "mfma_tile.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace geoadv {

// ---- from the other translation units (the Chamfer pieces of the attack loop) ----

constexpr int TR_THREADS = 512;
constexpr int TR_ROWS = 64;
constexpr float BN_EPS = 1e-5f;
__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
struct FwdArgs {
    const float *in;             // a_{i-1} [R][CIN]   (x [R][3] for layer 0)
    const float *pscale, *pshift;   // BN(batch) of the previous layer folded: h = relu(a * s + t)
    PackedLayer W;               // packed W_i (layers >= 1);  layer 0: W.w = canonical [3][64]
    const float *bias;           // b_i
    float *out;                  // a_i [R][COUT]
    float2 *psum;                // [tiles][COUT] (sum a, sum a^2) over the 64 rows of the tile
    int tiles;                   // 64-row tiles (the persistent forward kernel deals them round-robin)
    int *zero; int zero_count;   // layer 0 only: ints to clear on the way (the pool's maxima and tie counts: a memset launch less per step)
};

__global__ __launch_bounds__(TR_THREADS) void train_fwd0_kernel(FwdArgs A) {
    // layer 0 (fan-in 3) on the VALU: thread = (column, group of 8 rows)
    __shared__ float pts[TR_ROWS * 3];
    __shared__ float2 red[8][64];
    const size_t row0 = (size_t)blockIdx.x * TR_ROWS;
    for (int e = blockIdx.x * TR_THREADS + threadIdx.x; e < A.zero_count; e += gridDim.x * TR_THREADS) A.zero[e] = 0;
    if (threadIdx.x < TR_ROWS * 3) pts[threadIdx.x] = A.in[row0 * 3 + threadIdx.x];
    __syncthreads();
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const float w0 = A.W.w[c], w1 = A.W.w[64 + c], w2 = A.W.w[128 + c], b = A.bias[c];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int row = g * 8 + r;
        const float a = fmaf(pts[row * 3 + 2], w2, fmaf(pts[row * 3 + 1], w1, pts[row * 3] * w0)) + b;
        A.out[(row0 + row) * 64 + c] = a;
        s1 += a;
        s2 = fmaf(a, a, s2);
    }
    red[g][c] = make_float2(s1, s2);
    __syncthreads();
    if (threadIdx.x < 64) {
        float2 t = red[0][c];
#pragma unroll
        for (int k = 1; k < 8; ++k) { t.x += red[k][c].x; t.y += red[k][c].y; }
        A.psum[(size_t)blockIdx.x * 64 + c] = t;
    }
}

// h tile = relu(a * s + t) -> LDS [64][C + 4]
template <int C, int ROWS = TR_ROWS>
__device__ __forceinline__ void load_activated_tile(const float *a, size_t row0, const float *scale, const float *shift,
                                                    float *lds) {
    constexpr int Q = C / 4;                          // float4 per row; divides TR_THREADS
    const int c4 = threadIdx.x % Q, r0 = threadIdx.x / Q;
    const float4 s = reinterpret_cast<const float4 *>(scale)[c4], t = reinterpret_cast<const float4 *>(shift)[c4];
#pragma unroll
    for (int r = r0; r < ROWS; r += TR_THREADS / Q) {
        const float4 v = reinterpret_cast<const float4 *>(a + (row0 + r) * C)[c4];
        float4 h;
        h.x = fmaxf(fmaf(v.x, s.x, t.x), 0.f); h.y = fmaxf(fmaf(v.y, s.y, t.y), 0.f);
        h.z = fmaxf(fmaf(v.z, s.z, t.z), 0.f); h.w = fmaxf(fmaf(v.w, s.w, t.w), 0.f);
        *reinterpret_cast<float4 *>(lds + r * (C + 4) + 4 * c4) = h;
    }
}

// The loader waves' side of a tile: ROWS x C floats, global -> registers -> (activation) -> LDS [ROWS][C + 4], by the
// LOADER_THREADS threads [TR_THREADS, TR_THREADS + LOADER_THREADS) of the workgroup.
constexpr int LOADER_THREADS = 256;
template <int C, int ROWS> struct LoaderTile {
    static constexpr int Q = C / 4, STEP = LOADER_THREADS / Q, NP = ROWS / STEP;
    float4 v[NP];
    __device__ __forceinline__ void request(const float *a, size_t row0) {
        const int l = threadIdx.x - TR_THREADS, c4 = l % Q, r0 = l / Q;
#pragma unroll
        for (int j = 0; j < NP; ++j) v[j] = reinterpret_cast<const float4 *>(a + (row0 + r0 + j * STEP) * C)[c4];
    }
    // h = relu(a * s + t)   (s, t: this thread's four channels of the folded BN)
    __device__ __forceinline__ void store_activated(const float4 s, const float4 t, float *lds) const {
        const int l = threadIdx.x - TR_THREADS, c4 = l % Q, r0 = l / Q;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            float4 h;
            h.x = fmaxf(fmaf(v[j].x, s.x, t.x), 0.f); h.y = fmaxf(fmaf(v[j].y, s.y, t.y), 0.f);
            h.z = fmaxf(fmaf(v[j].z, s.z, t.z), 0.f); h.w = fmaxf(fmaf(v[j].w, s.w, t.w), 0.f);
            *reinterpret_cast<float4 *>(lds + (r0 + j * STEP) * (C + 4) + 4 * c4) = h;
        }
    }
};

// One forward layer, persistent and wave-specialised: 8 matrix waves + 4 loader waves per workgroup, two LDS tile buffers,
// and -- what makes it pay -- the weight ring carried from tile to tile and TWO tiles of loads in flight.
// What it replaces (one 64-row tile per workgroup, 1600 workgroups) ran at 0.49-0.56 of the MFMA peak: in-kernel stamps showed
// lockstep rounds (everyone loads, then everyone multiplies) and a near-empty third round, but compiling the loads and stores
// out left 65 of 84 us: a fixed ~2.3 us per tile and CU were weight-ring priming out of L2 at every tile start (one exposed L2
// round trip per chain, 11-22 % of these short chains), the epilogue and barrier skew.  A first persistent version without the
// carried ring gained nothing (84 us), which is how the ring was found.  Here a matrix wave's chain ends by refilling its ring
// with its own first four fragments (the next tile multiplies by the same weights): stamps then show the chain AT the pipe's
// rate (6.7 us for 7.0 us of MFMA issue per tile and CU at CIN = 256) -- and the matrix waves waiting ~3 us per tile for the
// loaders, whose requests take ~8 us to come back under this load.  So the loaders keep two register sets: tile j + 2 is
// requested while tile j is multiplied and staged (BN + ReLU -> the other LDS buffer) while tile j + 1 is.  One barrier per tile
// (two where the partial sums of two row blocks meet in LDS).  vmcnt retires in order, so the matrix waves themselves cannot
// prefetch tiles: their weight-fragment waits would wait for the tile's HBM round trip too.
// Tiles are dealt round-robin (tile = workgroup + j * workgroups): 1600 tiles over 256 workgroups leave the same 7-against-6.25
// imbalance per CU a dynamic queue would, and the loaders know their tiles two ahead.
constexpr int FW_THREADS = TR_THREADS + LOADER_THREADS;
template <int CIN, int COUT> struct FwdShape {
    static constexpr int TILE_FLOATS = TR_ROWS * (CIN + 4);
    static constexpr size_t lds_bytes = sizeof(float) * (2 * TILE_FLOATS + 4 * COUT);
    // ONE workgroup per CU for every shape.  Two fit for CIN <= 128 and were used until round 4, but they buy nothing: with global
    // loads, stores, sums and barriers compiled out (tools/debug/prof_train_fwd.sh on -D variants) the four launches still take
    // 18 / 30 / 56 / 56 us whether one or two workgroups share a CU -- the chains already run at the matrix pipe's rate
    // (tools/feed_probe.py: this loop shape sustains 0.96-0.98 of the fp32 peak), what is left is the launch's ramp and the 7-against-
    // 6.25 tiles per CU -- and the second workgroup's own weight ring and loads cost: 25.5 / 42.5 / 75.3 us with two, 24.9 / 40.3 /
    // 68.7 us with one (CIN = 256 never had room for two).
    static constexpr int WGS_PER_CU = 1;
    static constexpr int RM = COUT == 256 ? 2 : 1;          // row blocks per matrix wave (8 waves: 8 column blocks x 2, or 4 x 2 units)
};
template <int CIN, int COUT>
__global__ __launch_bounds__(FW_THREADS, (FwdShape<CIN, COUT>::WGS_PER_CU * 3)) void train_fwd_kernel(FwdArgs A) {
    extern __shared__ __align__(16) float lds[];
    using S = FwdShape<CIN, COUT>;
    static_assert(COUT == 128 || COUT == 256, "eight matrix waves cover 64 x 128 or 64 x 256 outputs");
    float2 *red = reinterpret_cast<float2 *>(lds + 2 * S::TILE_FLOATS);   // [2][COUT]
    [[maybe_unused]] constexpr int SK = (CIN == 128 && COUT == 256) ? 0 : (CIN == 256 ? 1 : (CIN == 128 ? 2 : 3));   // stamp slot (diagnostic builds)
    constexpr bool BOTH = S::RM == 2;                        // one matrix wave covers both row blocks
    GA_STAMP(SK, 0);
    const int G = gridDim.x, tiles = A.tiles;
    if ((int)blockIdx.x >= tiles) return;
    if (threadIdx.x >= TR_THREADS) {
        // ---------------- loader waves ----------------
        LoaderTile<CIN, TR_ROWS> P0, P1;                     // tile j lives in set j & 1
        const int c4 = (threadIdx.x - TR_THREADS) % (CIN / 4);
        const float4 ps = reinterpret_cast<const float4 *>(A.pscale)[c4], pt = reinterpret_cast<const float4 *>(A.pshift)[c4];
        int t = blockIdx.x;                                  // tile of iteration j
        P0.request(A.in, (size_t)t * TR_ROWS);
        if (t + G < tiles) P1.request(A.in, (size_t)(t + G) * TR_ROWS);
        P0.store_activated(ps, pt, lds);
        if (t + 2 * G < tiles) P0.request(A.in, (size_t)(t + 2 * G) * TR_ROWS);
        __syncthreads();                                     // P: tile 0 is in buffer 0
        for (;; t += 2 * G) {
            // even iteration j: tile j + 1 (set 1) -> buffer 1, then tile j + 3 -> set 1
            if (t + G < tiles) {
                P1.store_activated(ps, pt, lds + S::TILE_FLOATS);
                if (t + 3 * G < tiles) P1.request(A.in, (size_t)(t + 3 * G) * TR_ROWS);
            }
            if (!BOTH) __syncthreads();                      // X (the matrix waves' partial-sum exchange)
            __syncthreads();                                 // Y: tile j is consumed, tile j + 1 is staged
            if (t + G >= tiles) break;
            // odd iteration j + 1: tile j + 2 (set 0) -> buffer 0, then tile j + 4 -> set 0
            if (t + 2 * G < tiles) {
                P0.store_activated(ps, pt, lds);
                if (t + 4 * G < tiles) P0.request(A.in, (size_t)(t + 4 * G) * TR_ROWS);
            }
            if (!BOTH) __syncthreads();                      // X
            __syncthreads();                                 // Y
            if (t + 2 * G >= tiles) break;
        }
        return;
    }
    // ---------------- matrix waves ----------------
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, i = lane & 31;
    const int cb = BOTH ? wave : wave & 3, rb = BOTH ? 0 : wave >> 2;
    const int col = cb * 32 + i;
    const float b = A.bias[col];
    constexpr int kg = CIN >> 3;
    const FragSrc w = frag_src(A.W.w, cb * kg);
    const unsigned lb = (unsigned)lane * 16u;
    BRing ring;
    ring_fill(ring, w, lb);                                  // once: every chain refills it with its own first fragments
    __syncthreads();                                         // P
    GA_STAMP(SK, 1);
    for (int cur = blockIdx.x, it = 0; cur < tiles; cur += G, ++it) {
        const float *tile = lds + (it & 1) * S::TILE_FLOATS;
        const size_t row0 = (size_t)cur * TR_ROWS;
        const float *ar[S::RM];
#pragma unroll
        for (int rm = 0; rm < S::RM; ++rm) ar[rm] = tile + ((rb + rm) * 32 + i) * (CIN + 4) + 4 * h;
        f32x16 acc[S::RM] = {};
        if (it == 2) GA_STAMP(SK, 3);
        chain_ring_rm<kg, false, S::RM>(ar, 0, w, lb, ring, w, acc);
        if (it == 2) GA_STAMP(SK, 4);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int rm = 0; rm < S::RM; ++rm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a = acc[rm][r] + b;
                A.out[(row0 + (rb + rm) * 32 + acc_row(r, h)) * COUT + col] = a;
                s1 += a;
                s2 = fmaf(a, a, s2);
            }
        if (it == 0) GA_STAMP(SK, 2);
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (BOTH) {
            if (lane < 32) A.psum[(size_t)cur * COUT + col] = make_float2(s1, s2);
        } else {
            if (lane < 32) red[rb * COUT + col] = make_float2(s1, s2);
            if (it == 2) GA_STAMP(SK, 5);
            __syncthreads();                                 // X
            if (threadIdx.x < COUT) {
                const float2 p = red[threadIdx.x], q = red[COUT + threadIdx.x];
                A.psum[(size_t)cur * COUT + threadIdx.x] = make_float2(p.x + q.x, p.y + q.y);
            }
        }
        __syncthreads();                                     // Y
        if (it == 2) GA_STAMP(SK, 6);
    }
    GA_STAMP(SK, 7);
}

// Per-channel totals of per-tile partials (sum, sum of squares -- or the two BN-gradient sums) in a FIXED order, in double.
// A block = RED_CH channels x RED_GROUPS tile groups (1024 threads): thread (c, g) adds the tiles g, g + RED_GROUPS, ... in
// ascending order with all its loads in flight at once (<= 32 per thread at the step's sizes; more go in batches of 16), the
// groups are then folded 16 at a time and the eight folds added front to back.  8 channels per block instead of 32: four times
// the workgroups and a quarter of the tiles per thread -- these launches are a handful of workgroups wide and pure latency
// (9.7 -> ~5 us each, ten of them per step).  Returns true in the threads that hold a channel's totals (g == 0).
constexpr int RED_CH = 8, RED_GROUPS = 1024 / RED_CH;
__device__ __forceinline__ bool tile_partial_totals(const float2 *part, int tiles, int C, int block, double &s1, double &s2) {
    __shared__ double r1[RED_GROUPS][RED_CH], r2[RED_GROUPS][RED_CH];
    __shared__ double f1[8][RED_CH], f2[8][RED_CH];
    const int cl = threadIdx.x % RED_CH, g = threadIdx.x / RED_CH, c = block * RED_CH + cl;
    s1 = 0.0; s2 = 0.0;
    int t = g;
    for (; t + 15 * RED_GROUPS < tiles; t += 16 * RED_GROUPS) {
        float2 p[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) p[u] = part[(size_t)(t + RED_GROUPS * u) * C + c];
#pragma unroll
        for (int u = 0; u < 16; ++u) { s1 += p[u].x; s2 += p[u].y; }
    }
    {   // the rest (< 16 per thread): requested together, the ones beyond the end masked out
        float2 p[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int tt = t + RED_GROUPS * u;
            p[u] = part[(size_t)(tt < tiles ? tt : (tiles - 1)) * C + c];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (t + RED_GROUPS * u < tiles) { s1 += p[u].x; s2 += p[u].y; }
    }
    r1[g][cl] = s1; r2[g][cl] = s2;
    __syncthreads();
    if (g < 8) {
        double a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int k = 0; k < RED_GROUPS / 8; ++k) { a1 += r1[g * (RED_GROUPS / 8) + k][cl]; a2 += r2[g * (RED_GROUPS / 8) + k][cl]; }
        f1[g][cl] = a1; f2[g][cl] = a2;
    }
    __syncthreads();
    if (g != 0) return false;
    s1 = f1[0][cl]; s2 = f2[0][cl];
#pragma unroll
    for (int k = 1; k < 8; ++k) { s1 += f1[k][cl]; s2 += f2[k][cl]; }
    return true;
}

// Batch statistics of one layer from the per-tile partials (fixed order, double), the folded BN constants,
// and tflearn's moving-average update (assign_moving_average, zero_debias=False).
// mode 0: per-tile partials -> statistics in one go (single GPU).  Synchronised batch norm over several ranks splits it:
// mode 1 = partials -> the two per-channel totals (written to `totals`, which the host all-reduces, and to
// `local_totals`), mode 2 = all-reduced totals -> statistics, with inv_rows = 1 / (rows of ALL ranks).
struct BnArgs {
    int mode; double *totals, *local_totals;
    const float2 *psum; int tiles; int C; double inv_rows;
    const float *gamma, *beta;
    float *mean, *inv_std, *scale, *shift;    // batch mean, rsqrt(var + eps), gamma * inv_std, beta - mean * scale
    float *mov_mean, *mov_var; float one_minus_decay;
};

__global__ __launch_bounds__(1024) void bn_finalize_kernel(BnArgs A) {
    const int c = blockIdx.x * RED_CH + threadIdx.x % RED_CH;
    double s1 = 0.0, s2 = 0.0;
    bool mine = threadIdx.x < RED_CH;
    if (A.mode != 2) {
        mine = tile_partial_totals(A.psum, A.tiles, A.C, blockIdx.x, s1, s2);
        if (mine && A.mode == 1) { A.totals[c] = s1; A.totals[A.C + c] = s2; }
        if (A.mode == 1) return;
    } else if (mine) { s1 = A.totals[c]; s2 = A.totals[A.C + c]; }
    if (mine) {
        const double mean = s1 * A.inv_rows;
        double var = s2 * A.inv_rows - mean * mean;
        if (var < 0.0) var = 0.0;
        const float istd = (float)(1.0 / sqrt(var + (double)BN_EPS));
        const float sc = A.gamma[c] * istd;
        A.mean[c] = (float)mean; A.inv_std[c] = istd; A.scale[c] = sc;
        A.shift[c] = A.beta[c] - (float)mean * sc;
        A.mov_mean[c] -= (A.mov_mean[c] - (float)mean) * A.one_minus_decay;
        A.mov_var[c] -= (A.mov_var[c] - (float)var) * A.one_minus_decay;
    }
}

// Symmetric max-pool of h5 = relu(a4 * s + t) over the points of each cloud: values are >= 0, so the integer
// order of their bit patterns is their order and atomicMax (order independent) is exact.  Second pass: number of
// rows attaining the maximum (TF's _MinOrMaxGrad splits the gradient equally among them).
struct PoolArgs { const float *a4; const float *scale, *shift; int n_points; int *zbits; int *cnt; };

template <bool COUNT>
__device__ __forceinline__ void train_pool_block(const PoolArgs &A, const int tile) {
    __shared__ int red[8][128];
    const size_t row0 = (size_t)tile * TR_ROWS;
    const int cloud = (int)(row0 / A.n_points);
    const int c4 = threadIdx.x & 31, g = threadIdx.x >> 5;
    const float4 s = reinterpret_cast<const float4 *>(A.scale)[c4], t = reinterpret_cast<const float4 *>(A.shift)[c4];
    int4 z = make_int4(0, 0, 0, 0);
    if (COUNT) z = reinterpret_cast<const int4 *>(A.zbits + cloud * 128)[c4];
    int4 m = make_int4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float4 v = reinterpret_cast<const float4 *>(A.a4 + (row0 + g * 8 + r) * 128)[c4];
        const int hx = __float_as_int(fmaxf(fmaf(v.x, s.x, t.x), 0.f)), hy = __float_as_int(fmaxf(fmaf(v.y, s.y, t.y), 0.f));
        const int hz = __float_as_int(fmaxf(fmaf(v.z, s.z, t.z), 0.f)), hw = __float_as_int(fmaxf(fmaf(v.w, s.w, t.w), 0.f));
        if (COUNT) { m.x += hx == z.x; m.y += hy == z.y; m.z += hz == z.z; m.w += hw == z.w; }
        else { m.x = max(m.x, hx); m.y = max(m.y, hy); m.z = max(m.z, hz); m.w = max(m.w, hw); }
    }
    *reinterpret_cast<int4 *>(&red[g][4 * c4]) = m;
    __syncthreads();
    if (threadIdx.x < 128) {
        int v = red[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < 8; ++k) v = COUNT ? v + red[k][threadIdx.x] : max(v, red[k][threadIdx.x]);
        if (COUNT) { if (v) atomicAdd(A.cnt + cloud * 128 + threadIdx.x, v); }
        else atomicMax(A.zbits + cloud * 128 + threadIdx.x, v);
    }
}

template <bool COUNT>
__global__ __launch_bounds__(256) void train_pool_kernel(PoolArgs A) { train_pool_block<COUNT>(A, blockIdx.x); }

// ------------------------------------------------------------------------------------------------
// decoder (B rows only: VALU)
// ------------------------------------------------------------------------------------------------
// out[b][n] = act(in[b][:] @ W[:, n] + bias[n]); grid = B, block = NOUT (256)
template <int K, bool RELU>
__device__ __forceinline__ void fc_fwd_block(const float *in, const float *W, const float *bias, float *out, int nout, const int b) {
    __shared__ float x[K];
    if (threadIdx.x < K) x[threadIdx.x] = in[(size_t)b * K + threadIdx.x];
    __syncthreads();
    const int n = threadIdx.x;
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll 8
    for (int k = 0; k < K; k += 2) {
        acc0 = fmaf(x[k], W[(size_t)k * nout + n], acc0);
        acc1 = fmaf(x[k + 1], W[(size_t)(k + 1) * nout + n], acc1);
    }
    const float v = acc0 + acc1 + bias[n];
    out[(size_t)b * nout + n] = RELU ? fmaxf(v, 0.f) : v;
}

template <int K, bool RELU>
__global__ __launch_bounds__(256) void fc_fwd_kernel(const float *in, const float *W, const float *bias, float *out, int nout) {
    fc_fwd_block<K, RELU>(in, W, bias, out, nout, blockIdx.x);
}

// The middle decoder layer (256 -> 256, ReLU) on 1024 threads per cloud: thread = (output, K quarter), its 64 weights requested at
// once, the four partial sums added in a fixed order.  (One thread per output walking all 256 weights in batches of 16 loads was
// 16 dependent L2 round trips: 11.5 us for 64 K multiply-adds per cloud.)
__global__ __launch_bounds__(1024) void fc1_fwd_kernel(const float *in, const float *W, const float *bias, float *out) {
    __shared__ float x[256];
    __shared__ float part[4][256];
    const int t = threadIdx.x, o = t & 255, ks = t >> 8, b = blockIdx.x;
    float w[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) w[k] = W[(size_t)(ks * 64 + k) * 256 + o];
    const float bv = bias[o];
    if (t < 256) x[t] = in[(size_t)b * 256 + t];
    __syncthreads();
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < 64; k += 2) {
        s0 = fmaf(x[ks * 64 + k], w[k], s0);
        s1 = fmaf(x[ks * 64 + k + 1], w[k + 1], s1);
    }
    part[ks][o] = s0 + s1;
    __syncthreads();
    if (t < 256) out[(size_t)b * 256 + t] = fmaxf((((part[0][t] + part[1][t]) + part[2][t]) + part[3][t]) + bv, 0.f);
}

// First decoder layer (blocks [0, batch)) and the tie counts of the max-pool (the remaining blocks, one per tile): both
// need only the pooled maxima, neither needs the other.
__global__ __launch_bounds__(256) void fc0_and_pool_count_kernel(const float *z, const float *W, const float *bias, float *out, int batch,
                                                                 PoolArgs P) {
    if ((int)blockIdx.x < batch) fc_fwd_block<128, true>(z, W, bias, out, 256, blockIdx.x);
    else train_pool_block<true>(P, blockIdx.x - batch);
}

// recon[b][n] = d2[b][:] @ V2[:, n] + c2[n]; grid = (ceil(n3/64), ceil(B/16)), block 256 = 64 columns x 4 K slices
__global__ __launch_bounds__(256) void fc_out_fwd_kernel(const float *d2, const float *V2, const float *c2, float *out, float *out2, int batch,
                                                         int n3) {
    __shared__ __align__(16) float x[16][256];
    __shared__ float red[3][16][64];
    const int b0 = blockIdx.y * 16;
    const int col = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int n = min(blockIdx.x * 64 + col, n3 - 1);
    // Everything this thread needs from memory is requested before anything is waited for: the 64 weights of its column and
    // K slice, its 16 inputs of the row tile, the bias.  The launch is L2 round trips, not arithmetic: with eight weight
    // loads in flight per thread it took 22.7 us, with all of them 14.3, and with the row tile's request under them too 11.5.
    float wv[64], xin[16];
#pragma unroll
    for (int k = 0; k < 64; ++k) wv[k] = V2[(size_t)(ks * 64 + k) * n3 + n];
#pragma unroll
    for (int q = 0; q < 16; ++q) xin[q] = b0 + q < batch ? d2[(size_t)(b0 + q) * 256 + threadIdx.x] : 0.f;
    const float c = c2[n];
#pragma unroll
    for (int q = 0; q < 16; ++q) x[q][threadIdx.x] = xin[q];
    __syncthreads();
    float acc[16] = {};
#pragma unroll
    for (int k = 0; k < 64; k += 4) {
#pragma unroll
        for (int bb = 0; bb < 16; ++bb) {
            const float4 xv = *reinterpret_cast<const float4 *>(&x[bb][ks * 64 + k]);
            acc[bb] = fmaf(xv.w, wv[k + 3], fmaf(xv.z, wv[k + 2], fmaf(xv.y, wv[k + 1], fmaf(xv.x, wv[k], acc[bb]))));
        }
    }
    if (ks > 0) {
#pragma unroll
        for (int bb = 0; bb < 16; ++bb) red[ks - 1][bb][col] = acc[bb];
    }
    __syncthreads();
    if (ks == 0 && blockIdx.x * 64 + col < n3) {
#pragma unroll
        for (int bb = 0; bb < 16; ++bb)
            if (b0 + bb < batch) {
                const float r = (((acc[bb] + red[0][bb][col]) + red[1][bb][col]) + red[2][bb][col]) + c;
                out[(size_t)(b0 + bb) * n3 + n] = r;
                if (out2) out2[(size_t)(b0 + bb) * n3 + n] = r;         // the caller's copy (a 1.2 MB device-to-device copy launch less)
            }
    }
}

// loss 'emd': loss = inv * sum_b cost[b] (fixed order, fp64), g[e] *= inv -- block 0 also writes the loss
__global__ __launch_bounds__(256) void emd_loss_scale_kernel(const float *cost, int batch, double inv, float *loss, float *loss2, float *g,
                                                             size_t count) {
    const float s = (float)inv;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < count; e += (size_t)gridDim.x * 256) g[e] *= s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double acc = 0.0;
        for (int b = 0; b < batch; ++b) acc += (double)cost[b];
        *loss = (float)(acc * inv);
        if (loss2) *loss2 = (float)(acc * inv);
    }
}

// loss = (sum dist1 + sum dist2) / (B * N): tf.reduce_mean over all elements of each direction (pointnet_ae.py:77);
// one workgroup of 256 threads, fixed summation order
__device__ __forceinline__ void chamfer_loss_block(const float *d1, const float *d2, size_t count, double inv, float *loss, float *loss2) {
    __shared__ double red[256];
    const float4 *a = reinterpret_cast<const float4 *>(d1), *b = reinterpret_cast<const float4 *>(d2);
    const size_t q = count / 4;                               // count = B * n is a multiple of 64
    double s = 0.0;
    size_t e = threadIdx.x;
    for (; e + 768 < q; e += 1024) {
        float4 u[4], v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { u[k] = a[e + 256 * k]; v[k] = b[e + 256 * k]; }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            s += ((double)u[k].x + (double)u[k].y) + ((double)u[k].z + (double)u[k].w) + ((double)v[k].x + (double)v[k].y) +
                 ((double)v[k].z + (double)v[k].w);
    }
    for (; e < q; e += 256) {
        const float4 u = a[e], v = b[e];
        s += ((double)u.x + (double)u.y) + ((double)u.z + (double)u.w) + ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *loss = (float)(red[0] * inv);
        if (loss2) *loss2 = (float)(red[0] * inv);
    }
}

__global__ void fill_f32_kernel(float *p, float v, size_t count) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < count) p[e] = v;
}

// dV2[k][n] = sum_b d2[b][k] * g[b][n]; dc2[n] = sum_b g[b][n].  Block (bx, by) of (ceil(n3/64), 256/32), 256 threads = 64 columns x 4
// groups of 8 k (was: 128 columns x 32 k on HALF the block's threads -- under one wave per SIMD, 20 us of pure latency)
constexpr int DOW_COLS = 64;
__device__ __forceinline__ void fc_out_bwd_w_block(const float *d2, const float *g, float *dV2, float *dc2, int batch, int n3,
                                                   const int bx, const int by) {
    extern __shared__ __align__(16) float xs[];            // [batch][32]
    const int k0 = by * 32;
    for (int e = threadIdx.x; e < batch * 32; e += 256) xs[e] = d2[(size_t)(e >> 5) * 256 + k0 + (e & 31)];
    __syncthreads();
    const int col = threadIdx.x & 63, kq = threadIdx.x >> 6;     // this thread: column n, k in [k0 + 8 kq, k0 + 8 kq + 8)
    const int n = bx * DOW_COLS + col;
    if (n >= n3) return;
    float acc[8] = {};
    float gs = 0.f;
    constexpr int GB = 64;                                 // rows' gradients requested at once (ascending order kept): one round
    for (int b0 = 0; b0 < batch; b0 += GB) {               // trip for the default batch of 50
        float gvv[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) gvv[u] = g[(size_t)min(b0 + u, batch - 1) * n3 + n];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            if (b0 + u >= batch) break;
            const int b = b0 + u;
            const float gv = gvv[u];
            gs += gv;
            const float4 x0 = *reinterpret_cast<const float4 *>(&xs[b * 32 + 8 * kq]), x1 = *reinterpret_cast<const float4 *>(&xs[b * 32 + 8 * kq + 4]);
            acc[0] = fmaf(x0.x, gv, acc[0]); acc[1] = fmaf(x0.y, gv, acc[1]); acc[2] = fmaf(x0.z, gv, acc[2]); acc[3] = fmaf(x0.w, gv, acc[3]);
            acc[4] = fmaf(x1.x, gv, acc[4]); acc[5] = fmaf(x1.y, gv, acc[5]); acc[6] = fmaf(x1.z, gv, acc[6]); acc[7] = fmaf(x1.w, gv, acc[7]);
        }
    }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) dV2[(size_t)(k0 + 8 * kq + kk) * n3 + n] = acc[kk];
    if (by == 0 && kq == 0) dc2[n] = gs;
}

// dd2[b][k] = [d2[b][k] > 0] * sum_n g[b][n] * V2[k][n]; block (bx, by) of (256/DOX_K, ceil(B/DOX_B)), 256 threads striding n.
// A block streams DOX_K rows of V2 and DOX_B rows of g: 8 x 8 tiles move 88 MB through L2 in all, the earlier 4 x 8 tiles 131 MB
// (8 x 16: 75 MB but 128 accumulators per thread on 128 blocks: 40 us).
constexpr int DOX_K = 8, DOX_B = 8;
__device__ __forceinline__ void fc_out_bwd_x_block(const float *g, const float *V2, const float *d2, float *dd2, int batch, int n3,
                                                   const int bx, const int by) {
    __shared__ float red[4][DOX_K * DOX_B];
    const int k0 = bx * DOX_K, b0 = by * DOX_B;
    float acc[DOX_K][DOX_B] = {};
#pragma unroll 2
    for (int n = threadIdx.x; n < n3; n += 256) {
        float w[DOX_K], gv[DOX_B];
#pragma unroll
        for (int kk = 0; kk < DOX_K; ++kk) w[kk] = V2[(size_t)(k0 + kk) * n3 + n];
#pragma unroll
        for (int bb = 0; bb < DOX_B; ++bb) gv[bb] = g[(size_t)min(b0 + bb, batch - 1) * n3 + n];
#pragma unroll
        for (int kk = 0; kk < DOX_K; ++kk)
#pragma unroll
            for (int bb = 0; bb < DOX_B; ++bb) acc[kk][bb] = fmaf(w[kk], gv[bb], acc[kk][bb]);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int kk = 0; kk < DOX_K; ++kk)
#pragma unroll
        for (int bb = 0; bb < DOX_B; ++bb) {
            const float v = wave_sum(acc[kk][bb]);         // (DPP: six ds_bpermute round trips per value were ~5 us of this block)
            if (lane == 0) red[wave][kk * DOX_B + bb] = v;
        }
    __syncthreads();
    if (threadIdx.x < DOX_K * DOX_B) {
        const int kk = threadIdx.x / DOX_B, bb = threadIdx.x % DOX_B;
        if (b0 + bb < batch) {
            const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
            const size_t o = (size_t)(b0 + bb) * 256 + k0 + kk;
            dd2[o] = d2[o] > 0.f ? v : 0.f;
        }
    }
}

// The output layer's backward in ONE launch: its weight gradient, its data gradient and the scalar loss all depend
// only on the Chamfer results, not on each other.  blocks [0, nw): dV2 / dc2, [nw, nw + nx): dd2, last block: loss.
struct DecOutBwdArgs {
    const float *d2, *g, *V2; float *dV2, *dc2, *dd2; int batch, n3;
    const float *dist1, *dist2; size_t count; double inv; float *loss, *loss2;
};

__global__ __launch_bounds__(256) void dec_out_bwd_kernel(DecOutBwdArgs A) {
    const int wx = cdiv_dev(A.n3, DOW_COLS), nw = wx * 8, nxk = 256 / DOX_K, nx = nxk * cdiv_dev(A.batch, DOX_B);
    const int blk = blockIdx.x;
    if (blk < nw) fc_out_bwd_w_block(A.d2, A.g, A.dV2, A.dc2, A.batch, A.n3, blk % wx, blk / wx);
    else if (blk < nw + nx) fc_out_bwd_x_block(A.g, A.V2, A.d2, A.dd2, A.batch, A.n3, (blk - nw) % nxk, (blk - nw) / nxk);
    else chamfer_loss_block(A.dist1, A.dist2, A.count, A.inv, A.loss, A.loss2);
}

// Small dense layer backward: dW[k][n] = sum_b in[b][k] * dout[b][n] (grid.x = k), db[n] = sum_b dout[b][n]
__device__ __forceinline__ void fc_bwd_w_block(const float *in, const float *dout, float *dW, float *db, int batch, int K, const int k) {
    const int n = threadIdx.x;
    float acc = 0.f, s = 0.f;
#pragma unroll 8
    for (int b = 0; b < batch; ++b) {
        const float d = dout[(size_t)b * 256 + n];
        acc = fmaf(in[(size_t)b * K + k], d, acc);
        s += d;
    }
    dW[(size_t)k * 256 + n] = acc;
    if (k == 0) db[n] = s;
}

// din[b][k] = mask * sum_n dout[b][n] * W[k][n]; grid = B, block = K threads (one wave-strided row each)
template <bool MASK>
__device__ __forceinline__ void fc_bwd_x_block(const float *dout, const float *W, const float *act_in, float *din, int K, const int b) {
    __shared__ __align__(16) float d[256];
    d[threadIdx.x] = dout[(size_t)b * 256 + threadIdx.x];
    __syncthreads();
    const int k = threadIdx.x;
    if (k >= K) return;
    const float4 *w = reinterpret_cast<const float4 *>(W + (size_t)k * 256);
    float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
    for (int n4 = 0; n4 < 64; n4 += 2) {
        const float4 w0 = w[n4], w1 = w[n4 + 1];
        const float4 d0 = *reinterpret_cast<const float4 *>(&d[4 * n4]), d1 = *reinterpret_cast<const float4 *>(&d[4 * n4 + 4]);
        a0 = fmaf(w0.w, d0.w, fmaf(w0.z, d0.z, fmaf(w0.y, d0.y, fmaf(w0.x, d0.x, a0))));
        a1 = fmaf(w1.w, d1.w, fmaf(w1.z, d1.z, fmaf(w1.y, d1.y, fmaf(w1.x, d1.x, a1))));
    }
    const float v = a0 + a1;
    const size_t o = (size_t)b * K + k;
    din[o] = (!MASK || act_in[o] > 0.f) ? v : 0.f;
}

// weight gradient (blocks [0, K)) and data gradient (blocks [K, K + batch)) of a small dense layer in one launch
template <bool MASK>
__global__ __launch_bounds__(256) void fc_bwd_kernel(const float *in, const float *dout, const float *W, float *dW, float *db, float *din,
                                                     int batch, int K) {
    if ((int)blockIdx.x < K) fc_bwd_w_block(in, dout, dW, db, batch, K, blockIdx.x);
    else fc_bwd_x_block<MASK>(dout, W, in, din, K, blockIdx.x - K);
}

// ------------------------------------------------------------------------------------------------
// encoder backward
// ------------------------------------------------------------------------------------------------
// Gradient entering layer 4's BN output: dy4[r][c] = [h5 > 0 and h5 == z] * dz[b][c] / cnt[b][c]   (max-pool
// gradient with TF's equal split, then ReluGrad), stored densely, plus the per-tile sums of the BN backward.
struct PoolBwdArgs {
    const float *a4; const float *scale, *shift, *mean, *inv_std; int n_points;
    const int *zbits, *cnt; const float *dz;
    float *dy; float2 *qsum;
};

__global__ __launch_bounds__(256) void train_pool_bwd_kernel(PoolBwdArgs A) {
    __shared__ float2 red[8][128];
    const size_t row0 = (size_t)blockIdx.x * TR_ROWS;
    const int cloud = (int)(row0 / A.n_points);
    const int c4 = threadIdx.x & 31, g = threadIdx.x >> 5;
    const float4 s = reinterpret_cast<const float4 *>(A.scale)[c4], t = reinterpret_cast<const float4 *>(A.shift)[c4];
    const float4 mu = reinterpret_cast<const float4 *>(A.mean)[c4], is = reinterpret_cast<const float4 *>(A.inv_std)[c4];
    const int4 z = reinterpret_cast<const int4 *>(A.zbits + cloud * 128)[c4];
    const int4 cn = reinterpret_cast<const int4 *>(A.cnt + cloud * 128)[c4];
    const float4 dz = reinterpret_cast<const float4 *>(A.dz + cloud * 128)[c4];
    const float gx = z.x > 0 ? dz.x / (float)cn.x : 0.f, gy = z.y > 0 ? dz.y / (float)cn.y : 0.f;
    const float gz = z.z > 0 ? dz.z / (float)cn.z : 0.f, gw = z.w > 0 ? dz.w / (float)cn.w : 0.f;
    float q1[4] = {}, q2[4] = {};
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const size_t row = row0 + g * 8 + r;
        const float4 v = reinterpret_cast<const float4 *>(A.a4 + row * 128)[c4];
        float4 dy;
        dy.x = __float_as_int(fmaxf(fmaf(v.x, s.x, t.x), 0.f)) == z.x ? gx : 0.f;
        dy.y = __float_as_int(fmaxf(fmaf(v.y, s.y, t.y), 0.f)) == z.y ? gy : 0.f;
        dy.z = __float_as_int(fmaxf(fmaf(v.z, s.z, t.z), 0.f)) == z.z ? gz : 0.f;
        dy.w = __float_as_int(fmaxf(fmaf(v.w, s.w, t.w), 0.f)) == z.w ? gw : 0.f;
        reinterpret_cast<float4 *>(A.dy + row * 128)[c4] = dy;
        q1[0] += dy.x; q1[1] += dy.y; q1[2] += dy.z; q1[3] += dy.w;
        q2[0] = fmaf(dy.x, (v.x - mu.x) * is.x, q2[0]); q2[1] = fmaf(dy.y, (v.y - mu.y) * is.y, q2[1]);
        q2[2] = fmaf(dy.z, (v.z - mu.z) * is.z, q2[2]); q2[3] = fmaf(dy.w, (v.w - mu.w) * is.w, q2[3]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) red[g][4 * c4 + u] = make_float2(q1[u], q2[u]);
    __syncthreads();
    if (threadIdx.x < 128) {
        float2 v = red[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < 8; ++k) { v.x += red[k][threadIdx.x].x; v.y += red[k][threadIdx.x].y; }
        A.qsum[(size_t)blockIdx.x * 128 + threadIdx.x] = v;
    }
}

// d beta = sum dy, d gamma = sum dy * xhat (double, fixed order); m1 = d beta / R, m2 = d gamma / R
// modes as in BnArgs; d beta / d gamma always hold THIS rank's sums (the gradient all-reduce adds the ranks up), m1 / m2 the
// means over all ranks' rows
struct BnBwdArgs { int mode; double *totals, *local_totals; const float2 *qsum; int tiles; int C; double inv_rows; float *dbeta, *dgamma, *m1, *m2; };

__device__ __forceinline__ void bn_bwd_finalize_block(const BnBwdArgs &A, const int block) {
    const int c = block * RED_CH + threadIdx.x % RED_CH;
    double s1 = 0.0, s2 = 0.0, l1, l2;
    bool mine = threadIdx.x < RED_CH;
    if (A.mode != 2) {
        mine = tile_partial_totals(A.qsum, A.tiles, A.C, block, s1, s2);
        if (!mine) return;
        if (A.mode == 1) {
            A.totals[c] = s1; A.totals[A.C + c] = s2; A.local_totals[c] = s1; A.local_totals[A.C + c] = s2;
            return;
        }
        l1 = s1; l2 = s2;
    } else {
        if (!mine) return;
        s1 = A.totals[c]; s2 = A.totals[A.C + c]; l1 = A.local_totals[c]; l2 = A.local_totals[A.C + c];
    }
    A.dbeta[c] = (float)l1; A.dgamma[c] = (float)l2;
    A.m1[c] = (float)(s1 * A.inv_rows); A.m2[c] = (float)(s2 * A.inv_rows);
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(BnBwdArgs A) { bn_bwd_finalize_block(A, blockIdx.x); }

struct BwdArgs {
    int tiles;
    const float *dy;                 // [R][COUT]  gradient w.r.t. the BN output of this layer (ReLU mask applied)
    const float *a;                  // [R][COUT]  pre-BN activations of this layer
    const float *mean, *inv_std, *gamma, *m1, *m2;     // [COUT]
    const float *aprev;              // [R][CIN]   pre-BN activations of the previous layer (x for layer 0)
    const float *pscale, *pshift, *pmean, *pinv_std;   // [CIN] previous layer's BN (batch)
    PackedLayer WT;                  // packed W_i^T: K = COUT, N = CIN
    float *dy_out;                   // [R][CIN]
    float2 *qsum_out;                // [tiles][CIN]
    float *dw_partial;               // [grid][CIN][COUT]
    float *db_partial;               // [grid][COUT]
};

// One backward layer = one launch of persistent workgroups, one per CU, over 32-row tiles.  Inputs of a tile: dy_i, a_i (the
// layer's BN gradient is formed on load: da = gamma * inv_std * (dy - m1 - xhat * m2), xhat = (a - mean) * inv_std) and a_{i-1}.
// Until round 3 a layer was two KINDS of workgroups in one launch (weight gradient / data gradient), each staging its own copy
// of every tile and each leaving the matrix pipe idle while it waited for its own loads (in-kernel stamps: 12.5 us per 32-row
// tile for ~3.5 us of MFMAs in a weight-gradient workgroup; 6.0 us loading, 14.3 multiplying, 5.4 storing in a data-gradient
// one): 180 / 177 / 98 / 57 us for the four layers against 85 / 85 / 43 / 21 us of MFMA issue.  Now a workgroup stages a tile
// ONCE and has two kinds of waves:
//   "W" (8 waves): stage the tile -- da, h = relu(BN(a_prev)), raw a_prev -- and accumulate dW += h^T @ da in registers.
//                  Their MFMA loop reads LDS only, so the global loads of a later tile are issued before it and land in
//                  registers while everybody multiplies (vmcnt retires in order: the X waves could not do that, their
//                  weight-fragment waits would wait for the tile too);
//   "X" (4 or 8):  dy_prev = (da @ W^T) * [h > 0] with the BN sums of the layer below; their epilogue runs from registers.
// With two tile buffers (all shapes but CIN = 256) tile k + 1 is staged at the START of iteration k -- while the X waves' chains
// have the matrix pipe to themselves -- from registers requested during iteration k - 1, then tile k + 2 is requested, then the
// products of tile k: ONE barrier per tile and no phase in which nobody multiplies.  With one buffer: request, products,
// barrier B (all reads done), stage, barrier A.  Where the X waves split K, B also hands their partials over.
// Measured 128.5 / 128.2 / 73.2 / 47.1 us.  What is left: a tile's loads are a bandwidth-bound burst (every workgroup asks at
// once; ~8 us until the data is there) and one tile (80 KB per CU) in flight does not cover that -- a second register set does
// not fit (dedicated loader waves holding the next tile in 80 registers were measured: 141 us, they serialise on that latency;
// tools/experiments/train_bwd_fused_loader_waves.patch).  Skewing the workgroups' phases against each other (starts delayed by
// 0 / 1 / 2 / 3 x 3.4 us, every XCD holding all four phases) only adds the delay: 132 / 131 / 75 / 48 -> 140 / 135 / 83 / 56 us.
// Two staging register sets in the W waves (64 accumulators + 2 x 40) spill 256-288 registers at the 168 cap; touching one
// dword per line of the tile after next so that it waits in L2 made everything slower (164 / 150 / 90 / 58 us: the touched
// lines are fetched twice or push the weights out).
// Tiles are dealt round-robin (tile = workgroup + k * workgroups): a workgroup's dW partial sums over a fixed set of tiles in
// a fixed order, so the step stays deterministic.
// XW = number of X waves: 8 (16 waves, 128 VGPRs each) where the W waves' 64 accumulator + 40 staging registers leave room,
// 4 (12 waves, 168 VGPRs) for the two wide layers, where they do not (54 / 81 spilled registers made those launches slower
// than the split form); an X wave then owns a whole K chain (and two column blocks at CIN = 256).
constexpr int BWD_ROWS = 64;     // BwdArgs::tiles counts 64-row tiles
constexpr int BF_ROWS = 32, BF_WTHREADS = 512;
template <int CIN, int COUT> struct FusedShape {
    static constexpr int MB = CIN / 32, NB = COUT / 32;
    static constexpr int MBW = (MB * NB >= 32) ? 2 : 1, NBW = (MB * NB >= 16) ? 2 : 1, WCOLS = NB / NBW;
    static_assert((MB / MBW) * (NB / NBW) == 8, "dW blocks must map onto 8 waves");
    static constexpr int XW = (MB * NB >= 32) ? 4 : 8;                     // X waves
    static constexpr int THREADS = (XW + 8) * 64;
    static constexpr int UNITS = MB;                                       // X: column blocks of dy_prev
    static constexpr int CPW = UNITS >= XW ? UNITS / XW : 1, KC = UNITS >= XW ? 1 : XW / UNITS;   // column blocks per wave / K parts
    static_assert((COUT / 8 / KC) % 4 == 0, "bad K split");
    static constexpr int DA_FLOATS = BF_ROWS * (COUT + 4), H_FLOATS = BF_ROWS * (CIN + 4);
    static constexpr int TILE_FLOATS = DA_FLOATS + 2 * H_FLOATS;           // one staged tile: da, h = relu(BN(a_prev)), raw a_prev
    static constexpr int SCRATCH_FLOATS = (KC - 1) * UNITS * 16 * 64, CC_FLOATS = 5 * COUT;
    static_assert(DA_FLOATS >= (BF_WTHREADS / (COUT / 4)) * COUT, "bias-gradient reduction aliases the da tile");
    // two tile buffers where they fit (all but CIN = 256): the W waves then stage tile k + 1 while tile k is still being read
    // and a tile costs ONE barrier; with one buffer they stage between two barriers
    static constexpr bool DB = sizeof(float) * (2 * TILE_FLOATS + SCRATCH_FLOATS + CC_FLOATS) <= 160 * 1024;
    static constexpr size_t lds_bytes = sizeof(float) * ((DB ? 2 : 1) * TILE_FLOATS + SCRATCH_FLOATS + CC_FLOATS);
    // staging: float4s per W thread and tile
    static constexpr int QO = COUT / 4, SO = BF_WTHREADS / QO, NO = BF_ROWS / SO;      // dy, a
    static constexpr int QI = CIN / 4, SI = BF_WTHREADS / QI, NI = (BF_ROWS + SI - 1) / SI;   // a_prev (CIN = 64: one pass covers the 32 rows)
};

template <int CIN, int COUT>
__global__ __launch_bounds__((FusedShape<CIN, COUT>::THREADS)) void train_bwd_fused_kernel(BwdArgs A) {
    extern __shared__ __align__(16) float lds[];
    using S = FusedShape<CIN, COUT>;
    float *scratch = lds + (S::DB ? 2 : 1) * S::TILE_FLOATS, *cc = scratch + S::SCRATCH_FLOATS;   // tile buffer b at lds + b * TILE_FLOATS: da, h, raw a_prev
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int hh = lane >> 5, li = lane & 31;
    const int tiles32 = A.tiles * (64 / BF_ROWS), G = gridDim.x;
    for (int e = threadIdx.x; e < COUT; e += S::THREADS) {
        cc[e] = A.mean[e]; cc[COUT + e] = A.inv_std[e]; cc[2 * COUT + e] = A.gamma[e] * A.inv_std[e];
        cc[3 * COUT + e] = A.m1[e]; cc[4 * COUT + e] = A.m2[e];
    }
    __syncthreads();
    if (wave >= S::XW) {
        // ---------------- W waves: staging + weight gradient ----------------
        const int tw = threadIdx.x - S::XW * 64;
        const int oc4 = tw % S::QO, or0 = tw / S::QO, ic4 = tw % S::QI, ir0 = tw / S::QI;
        const int w8 = wave - S::XW, mb0 = (w8 / S::WCOLS) * S::MBW, nb0 = (w8 % S::WCOLS) * S::NBW;
        // h = relu(a_prev * s + t) once per element on the way into LDS, not once per use in the MFMA loop: VALU instructions in
        // that loop cost matrix-pipe time (measured: 7.9 -> 10.1 us for a tile's products with the activation on the operand)
        const float4 ps = reinterpret_cast<const float4 *>(A.pscale)[ic4], pt = reinterpret_cast<const float4 *>(A.pshift)[ic4];
        f32x16 dw[S::MBW][S::NBW] = {};
        float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 rdy[S::NO], ra[S::NO], rp[S::NI];
        auto request = [&](int tile) {
            const size_t row0 = (size_t)tile * BF_ROWS;
#pragma unroll
            for (int j = 0; j < S::NO; ++j) {
                rdy[j] = reinterpret_cast<const float4 *>(A.dy + (row0 + or0 + j * S::SO) * COUT)[oc4];
                ra[j] = reinterpret_cast<const float4 *>(A.a + (row0 + or0 + j * S::SO) * COUT)[oc4];
            }
#pragma unroll
            for (int j = 0; j < S::NI; ++j)
                if (ir0 + j * S::SI < BF_ROWS) rp[j] = reinterpret_cast<const float4 *>(A.aprev + (row0 + ir0 + j * S::SI) * CIN)[ic4];
        };
        auto stage = [&](float *da) {
            float *ht = da + S::DA_FLOATS, *araw = ht + S::H_FLOATS;
            const float4 mu = reinterpret_cast<const float4 *>(cc)[oc4], is = reinterpret_cast<const float4 *>(cc + COUT)[oc4];
            const float4 gi = reinterpret_cast<const float4 *>(cc + 2 * COUT)[oc4];
            const float4 m1 = reinterpret_cast<const float4 *>(cc + 3 * COUT)[oc4], m2 = reinterpret_cast<const float4 *>(cc + 4 * COUT)[oc4];
#pragma unroll
            for (int j = 0; j < S::NO; ++j) {
                const float4 d = rdy[j], a = ra[j];
                float4 o;
                o.x = gi.x * ((d.x - m1.x) - (a.x - mu.x) * is.x * m2.x); o.y = gi.y * ((d.y - m1.y) - (a.y - mu.y) * is.y * m2.y);
                o.z = gi.z * ((d.z - m1.z) - (a.z - mu.z) * is.z * m2.z); o.w = gi.w * ((d.w - m1.w) - (a.w - mu.w) * is.w * m2.w);
                *reinterpret_cast<float4 *>(da + (or0 + j * S::SO) * (COUT + 4) + 4 * oc4) = o;
                dbacc.x += o.x; dbacc.y += o.y; dbacc.z += o.z; dbacc.w += o.w;
            }
#pragma unroll
            for (int j = 0; j < S::NI; ++j) {
                if (ir0 + j * S::SI >= BF_ROWS) continue;
                const float4 v = rp[j];
                float4 h;
                h.x = fmaxf(fmaf(v.x, ps.x, pt.x), 0.f); h.y = fmaxf(fmaf(v.y, ps.y, pt.y), 0.f);
                h.z = fmaxf(fmaf(v.z, ps.z, pt.z), 0.f); h.w = fmaxf(fmaf(v.w, ps.w, pt.w), 0.f);
                *reinterpret_cast<float4 *>(ht + (ir0 + j * S::SI) * (CIN + 4) + 4 * ic4) = h;
                *reinterpret_cast<float4 *>(araw + (ir0 + j * S::SI) * (CIN + 4) + 4 * ic4) = v;
            }
        };
        // Two buffers (DB): tile k + 1 is staged at the START of iteration k -- while the X waves' chains have the matrix pipe --
        // from registers requested one iteration earlier, then tile k + 2 is requested, then the products of tile k: ONE barrier
        // per tile and no phase in which nobody multiplies.  One buffer: request, products, barrier B, stage, barrier A.
        int tile = blockIdx.x;
        if (tile < tiles32) { request(tile); stage(lds); }
        if (S::DB && tile + G < tiles32) request(tile + G);
        __syncthreads();                                   // A: the first tile is in buffer 0
        [[maybe_unused]] const bool st_on = CIN == 128 && COUT == 256;   // stamps (diagnostic builds): the second tile's phases
        for (int it = 0; tile < tiles32; tile += G, ++it) {
            const bool more = tile + G < tiles32;
            if (st_on && it == 1) GA_STAMP_T(6, 0, S::XW * 64);
            if (S::DB) {
                if (more) stage(lds + ((it + 1) & 1) * S::TILE_FLOATS);   // the OTHER buffer: its tile was consumed before the last barrier
                if (tile + 2 * G < tiles32) request(tile + 2 * G);
            } else if (more) request(tile + G);
            __builtin_amdgcn_sched_barrier(0);             // (the requests stay ahead of the products)
            if (st_on && it == 1) GA_STAMP_T(6, 1, S::XW * 64);
            const float *da = lds + (S::DB ? it & 1 : 0) * S::TILE_FLOATS, *ht = da + S::DA_FLOATS;
#pragma unroll 8
            for (int kk = 0; kk < BF_ROWS / 2; ++kk) {
                const int row = 2 * kk + hh;
                float av[S::MBW], bv[S::NBW];
#pragma unroll
                for (int m = 0; m < S::MBW; ++m) av[m] = ht[row * (CIN + 4) + (mb0 + m) * 32 + li];
#pragma unroll
                for (int n = 0; n < S::NBW; ++n) bv[n] = da[row * (COUT + 4) + (nb0 + n) * 32 + li];
#pragma unroll
                for (int m = 0; m < S::MBW; ++m)
#pragma unroll
                    for (int n = 0; n < S::NBW; ++n)
                        dw[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], dw[m][n], 0, 0, 0);
            }
            if (st_on && it == 1) GA_STAMP_T(6, 2, S::XW * 64);
            if (S::KC > 1 || !S::DB) __syncthreads();      // B: the X waves' K partials are in scratch / every read of the only buffer is done
            if (!S::DB && more) stage(lds);
            if (st_on && it == 1) GA_STAMP_T(6, 3, S::XW * 64);
            __syncthreads();                               // A: tile it is consumed, tile it + 1 is staged
            if (st_on && it == 1) GA_STAMP_T(6, 7, S::XW * 64);
        }
        float *dst = A.dw_partial + (size_t)blockIdx.x * CIN * COUT;
#pragma unroll
        for (int m = 0; m < S::MBW; ++m)
#pragma unroll
            for (int n = 0; n < S::NBW; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    dst[(size_t)((mb0 + m) * 32 + acc_row(r, hh)) * COUT + (nb0 + n) * 32 + li] = dw[m][n][r];
        // bias gradient: this workgroup's column sums, exchanged through the da tile (nobody reads it any more: the X waves
        // have left the loop behind the same last barrier A, and take part in the barrier below)
        float *dbred = lds;
        *reinterpret_cast<float4 *>(dbred + or0 * COUT + 4 * oc4) = dbacc;
        __syncthreads();
        if (tw < COUT) {
            float sacc = dbred[tw];
#pragma unroll
            for (int g = 1; g < S::SO; ++g) sacc += dbred[g * COUT + tw];
            A.db_partial[(size_t)blockIdx.x * COUT + tw] = sacc;
        }
        return;
    }
    // ---------------- X waves: data gradient ----------------
    const int unit = wave % S::UNITS, ks = S::KC > 1 ? wave / S::UNITS : 0;
    constexpr int kg = COUT >> 3;
    float pps[S::CPW], ppt[S::CPW], ppm[S::CPW], ppis[S::CPW];         // the previous layer's BN constants of this lane's column(s)
#pragma unroll
    for (int j = 0; j < S::CPW; ++j) {
        const int oc = (S::CPW > 1 ? wave + j * S::XW : unit) * 32 + li;
        pps[j] = A.pscale[oc]; ppt[j] = A.pshift[oc]; ppm[j] = A.pmean[oc]; ppis[j] = A.pinv_std[oc];
    }
    __syncthreads();                                       // A: the first tile is in LDS
    [[maybe_unused]] const bool st_on = CIN == 128 && COUT == 256;
    for (int tile = blockIdx.x, it = 0; tile < tiles32; tile += G, ++it) {
        const size_t row0 = (size_t)tile * BF_ROWS;
        if (st_on && it == 1) GA_STAMP(7, 0);
        const float *da = lds + (S::DB ? it & 1 : 0) * S::TILE_FLOATS, *araw = da + S::DA_FLOATS + S::H_FLOATS;
#pragma unroll
        for (int j = 0; j < S::CPW; ++j) {
            const int cb = S::CPW > 1 ? wave + j * S::XW : unit;
            const int ocol = cb * 32 + li;
            float apv[16];
            if (ks == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) apv[r] = araw[acc_row(r, hh) * (CIN + 4) + ocol];
            }
            f32x16 acc[1] = {};
            gemm_chain<1>(da, COUT + 4, 0, A.WT, cb, ks * kg / S::KC, (ks + 1) * kg / S::KC, acc);
            if (st_on && it == 1) GA_STAMP(7, 1);
            if (S::KC == 1 && !S::DB && j == S::CPW - 1) __syncthreads();   // B (one buffer: this wave's last read of the tile is done; the epilogue runs from registers)
            if (S::KC > 1) {
                if (ks > 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) scratch[(((ks - 1) * S::UNITS + unit) * 16 + r) * 64 + lane] = acc[0][r];
                }
                __syncthreads();                           // B (KC > 1: the K partials are in scratch)
                if (ks == 0) {
#pragma unroll
                    for (int p = 1; p < S::KC; ++p)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[0][r] += scratch[(((p - 1) * S::UNITS + unit) * 16 + r) * 64 + lane];
                }
            }
            if (ks == 0) {
                float q1 = 0.f, q2 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float dyv = fmaf(apv[r], pps[j], ppt[j]) > 0.f ? acc[0][r] : 0.f;
                    A.dy_out[(row0 + acc_row(r, hh)) * CIN + ocol] = dyv;
                    q1 += dyv;
                    q2 = fmaf(dyv, (apv[r] - ppm[j]) * ppis[j], q2);
                }
                q1 += __shfl_xor(q1, 32);
                q2 += __shfl_xor(q2, 32);
                if (hh == 0) A.qsum_out[(size_t)tile * CIN + ocol] = make_float2(q1, q2);
            }
        }
        if (st_on && it == 1) GA_STAMP(7, 2);
        __syncthreads();                                   // A
        if (st_on && it == 1) GA_STAMP(7, 7);
    }
    __syncthreads();                                       // (the W waves' bias-gradient exchange)
}

// layer 0: dW0[k][c] = sum_r x[r][k] * da0[r][c], db0[c] = sum_r da0[r][c]; persistent, thread = (column, 8-row group)
__global__ __launch_bounds__(TR_THREADS) void train_bwd0_kernel(BwdArgs A) {
    __shared__ float pts[TR_ROWS * 3];
    __shared__ float red[8][4][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const float mu = A.mean[c], is = A.inv_std[c], gi = A.gamma[c] * A.inv_std[c], m1 = A.m1[c], m2 = A.m2[c];
    float w0 = 0.f, w1 = 0.f, w2 = 0.f, sb = 0.f;
    for (int tile = blockIdx.x; tile < A.tiles; tile += gridDim.x) {
        const size_t row0 = (size_t)tile * TR_ROWS;
        __syncthreads();
        if (threadIdx.x < TR_ROWS * 3) pts[threadIdx.x] = A.aprev[row0 * 3 + threadIdx.x];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = g * 8 + r;
            const float d = A.dy[(row0 + row) * 64 + c], a = A.a[(row0 + row) * 64 + c];
            const float o = gi * ((d - m1) - (a - mu) * is * m2);
            w0 = fmaf(pts[row * 3], o, w0); w1 = fmaf(pts[row * 3 + 1], o, w1); w2 = fmaf(pts[row * 3 + 2], o, w2);
            sb += o;
        }
    }
    red[g][0][c] = w0; red[g][1][c] = w1; red[g][2][c] = w2; red[g][3][c] = sb;
    __syncthreads();
    if (threadIdx.x < 256) {
        const int k = threadIdx.x >> 6;
        float s = red[0][k][c];
#pragma unroll
        for (int j = 1; j < 8; ++j) s += red[j][k][c];
        if (k < 3) A.dw_partial[(size_t)blockIdx.x * 192 + k * 64 + c] = s;
        else A.db_partial[(size_t)blockIdx.x * 64 + c] = s;
    }
}

// out[e] = sum over workgroups of partial[w][e], in a fixed order: four contiguous runs of workgroups summed
// front to back by four threads, then run0 + run1 + run2 + run3.  COLS elements per block, 4 * COLS threads.
struct ReduceArgs { const float *partial; int parts; size_t count; float *out; };

template <int COLS>
__device__ __forceinline__ void partial_reduce_block(const ReduceArgs &A, const int block) {
    __shared__ float red[4][COLS];
    const int col = threadIdx.x % COLS, g = threadIdx.x / COLS;
    const size_t e = (size_t)block * COLS + col;
    const int per = (A.parts + 3) / 4, w0 = g * per, w1 = min(A.parts, w0 + per);
    float s = 0.f;
    if (e < A.count) {
        int w = w0;
        for (; w + 8 <= w1; w += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = A.partial[(size_t)(w + u) * A.count + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; w < w1; ++w) s += A.partial[(size_t)w * A.count + e];
    }
    red[g][col] = s;
    __syncthreads();
    if (g == 0 && e < A.count) A.out[e] = ((red[0][col] + red[1][col]) + red[2][col]) + red[3][col];
}

__global__ __launch_bounds__(256) void partial_reduce_kernel(const float *partial, int parts, size_t count, float *out) {
    partial_reduce_block<64>(ReduceArgs{partial, parts, count, out}, blockIdx.x);
}

// Everything that follows a backward layer kernel in ONE launch (three tiny dependent-free jobs, each of which would
// otherwise pay a launch of its own): the BN-gradient sums of the layer below (blocks [0, bn_blocks)), then the
// reductions of the layer's weight- and bias-gradient partials.
__global__ __launch_bounds__(1024) void post_layer_kernel(BnBwdArgs bn, int bn_blocks, ReduceArgs dw, int dw_blocks, ReduceArgs db) {
    const int blk = blockIdx.x;
    if (blk < bn_blocks) bn_bwd_finalize_block(bn, blk);
    else if (blk < bn_blocks + dw_blocks) partial_reduce_block<256>(dw, blk - bn_blocks);
    else partial_reduce_block<256>(db, blk - bn_blocks - dw_blocks);
}

// ------------------------------------------------------------------------------------------------
// optimizer + re-packing
// ------------------------------------------------------------------------------------------------
// TF 1.13 ApplyAdam over the flat parameter arena (see attack.hip adam_kernel for the restated form).  An element that
// belongs to one of the four wide encoder matrices is also written straight into its two MFMA fragment positions
// (W_i for the forward, W_i^T for the backward), so the step needs no separate re-packing launch.
struct AdamPack { size_t off[4]; int K[4], N[4]; float *fwd[4], *bwd[4]; };

__global__ __launch_bounds__(256) void train_adam_kernel(float *p, float *m, float *v, const float *g, size_t count, float gscale,
                                                         float lr, float b1p, float b2p, AdamPack pk) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= count) return;
    const float alpha = lr * sqrtf(1.f - b2p) / (1.f - b1p);
    const float gv = g[e] * gscale;
    const float mn = m[e] + (gv - m[e]) * (1.f - 0.9f);
    const float vn = v[e] + (gv * gv - v[e]) * (1.f - 0.999f);
    m[e] = mn; v[e] = vn;
    const float pn = p[e] - (mn * alpha) / (sqrtf(vn) + 1e-8f);
    p[e] = pn;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int K = pk.K[j], N = pk.N[j];
        if (e >= pk.off[j] && e < pk.off[j] + (size_t)K * N) {
            const int r = (int)(e - pk.off[j]), k = r / N, n = r % N;        // W[k][n]
            // forward fragments: B[k][n] = W[k][n]
            pk.fwd[j][((size_t)((n >> 5) * (K >> 3) + (k >> 3)) * 64 + (((k >> 2) & 1) << 5) + (n & 31)) * 4 + (k & 3)] = pn;
            // backward fragments: B'[k' = n][n' = k] = W[k][n], K' = N, N' = K
            pk.bwd[j][((size_t)((k >> 5) * (N >> 3) + (n >> 3)) * 64 + (((n >> 2) & 1) << 5) + (k & 31)) * 4 + (n & 3)] = pn;
        }
    }
}

// packed[((cb * K/8 + t) * 64 + lane) * 4 + u] = B[8t + 4*(lane>>5) + u][32cb + (lane&31)];
// B = W [K][N] (transpose == 0) or B[k][n] = W[n][k] with W stored [N][K] (transpose == 1).  One launch re-packs
// the forward and the transposed fragments of the four wide encoder layers.
struct RepackArgs { const float *W[8]; float *packed[8]; int K[8], N[8], transpose[8], first_block[9]; };

__global__ __launch_bounds__(256) void repack_kernel(RepackArgs A) {
    int j = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) j += (int)blockIdx.x >= A.first_block[k];
    const int K = A.K[j], N = A.N[j];
    const int e = ((int)blockIdx.x - A.first_block[j]) * 256 + threadIdx.x;
    if (e >= K * N) return;
    const int u = e & 3, lane = (e >> 2) & 63, rest = e >> 8;
    const int kg = K / 8, t = rest % kg, cb = rest / kg;
    const int k = 8 * t + 4 * (lane >> 5) + u, n = 32 * cb + (lane & 31);
    A.packed[j][e] = A.transpose[j] ? A.W[j][(size_t)n * K + k] : A.W[j][(size_t)k * N + n];
}

}  // namespace geoadv

using namespace geoadv;

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {

const int ENC[ENC_L + 1] = {3, 64, 128, 128, 256, 128};

struct Layout {                        // offsets (floats) into the flat parameter / gradient / Adam arenas
    size_t w[ENC_L], b[ENC_L], gamma[ENC_L], beta[ENC_L], v[3], c[3], count;
};

Layout make_layout(int n_points) {
    Layout L;
    size_t o = 0;
    auto take = [&](size_t cnt) { size_t r = o; o += (cnt + 3) / 4 * 4; return r; };
    for (int i = 0; i < ENC_L; ++i) {
        L.w[i] = take((size_t)ENC[i] * ENC[i + 1]); L.b[i] = take(ENC[i + 1]);
        L.gamma[i] = take(ENC[i + 1]); L.beta[i] = take(ENC[i + 1]);
    }
    const int dd[4] = {128, 256, 256, 3 * n_points};
    for (int k = 0; k < 3; ++k) { L.v[k] = take((size_t)dd[k] * dd[k + 1]); L.c[k] = take(dd[k + 1]); }
    L.count = o;
    return L;
}

}  // namespace

struct geoadv_trainer {
    int B, N, R, tiles, n3, grid_bwd, cus, max_wgs;
    float lr, one_minus_decay;
    float b1p, b2p;
    Layout L;
    char *arena; size_t arena_bytes;
    // carved from the arena
    float *params, *grads, *adam_m, *adam_v;
    float *mov_mean[ENC_L], *mov_var[ENC_L];
    float *packed_fwd[ENC_L], *packed_bwd[ENC_L];
    float *bn_mean[ENC_L], *bn_istd[ENC_L], *bn_scale[ENC_L], *bn_shift[ENC_L], *bn_m1[ENC_L], *bn_m2[ENC_L];
    float *act[ENC_L];                 // a_i [R][C_{i+1}]
    float *dybuf[2];                   // [R][256] ping-pong
    float2 *psum, *qsum;               // [tiles][256]
    int *zbits, *cnt;                  // [B][128]
    float *d1, *d2, *recon, *g_recon, *cham_ws, *gd, *dd2, *dd1, *dz;
    float *dist1, *dist2; int *idx1, *idx2;
    float *dw_partial, *db_partial;
    float *loss;
    float *user_loss, *user_recon;     // during geoadv_trainer_forward_backward: the caller's buffers, written by the producing kernels
    int loss_type;                     // GEOADV_TRAIN_LOSS_*
    float *emd_temp, *emd_cost;        // loss 'emd': scratch of geoadv_emd_cost_grad1, per-cloud match costs
    int world;                         // ranks sharing the batch statistics (synchronised BN); 1 = local
    double *xbuf, *lbuf;               // [10][512] per-phase totals: exchanged (all-reduced by the host) / local copy
};

static int trainer_repack(geoadv_trainer *t, hipStream_t st) {
    RepackArgs a;
    int blocks = 0;
    for (int i = 1; i < ENC_L; ++i)
        for (int tr = 0; tr < 2; ++tr) {
            const int j = 2 * (i - 1) + tr;
            a.W[j] = t->params + t->L.w[i];
            a.packed[j] = tr ? t->packed_bwd[i] : t->packed_fwd[i];
            a.K[j] = tr ? ENC[i + 1] : ENC[i]; a.N[j] = tr ? ENC[i] : ENC[i + 1]; a.transpose[j] = tr;
            a.first_block[j] = blocks;
            blocks += cdiv(ENC[i] * ENC[i + 1], 256);
        }
    a.first_block[8] = blocks;
    repack_kernel<<<blocks, 256, 0, st>>>(a);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

extern "C" int geoadv_trainer_create(geoadv_trainer **out, const geoadv_ae_weights *hw, const geoadv_train_config *cfg) {
    GA_REQUIRE(out && hw && cfg, "trainer_create: null argument");
    for (int i = 0; i <= ENC_L; ++i)
        GA_REQUIRE(hw->enc_dims[i] == ENC[i], "trainer_create: encoder widths must be 3,64,128,128,256,128 (src/ae_templates.py:22)");
    const int n = hw->n_points, B = cfg->batch;
    GA_REQUIRE(n >= 64 && n % 64 == 0 && n <= 32768, "trainer_create: n_points %d must be a multiple of 64 in [64, 32768]", n);
    GA_REQUIRE(cfg->max_workgroups >= 0, "trainer_create: max_workgroups %d must be >= 0", cfg->max_workgroups);
    GA_REQUIRE(B >= 1 && B <= 4096, "trainer_create: batch %d out of range [1, 4096]", B);
    GA_REQUIRE(hw->dec_dims[0] == 128 && hw->dec_dims[1] == 256 && hw->dec_dims[2] == 256 && hw->dec_dims[3] == 3 * n,
               "trainer_create: decoder widths must be 128,256,256,3*n_points (src/ae_templates.py:29)");
    GA_REQUIRE(cfg->learning_rate > 0.f && cfg->bn_decay >= 0.f && cfg->bn_decay <= 1.f, "trainer_create: bad learning rate / decay");
    GA_REQUIRE(cfg->loss == GEOADV_TRAIN_LOSS_CHAMFER || cfg->loss == GEOADV_TRAIN_LOSS_EMD, "trainer_create: unknown loss %d", cfg->loss);
    geoadv_trainer *t = new geoadv_trainer();
    t->B = B; t->N = n; t->R = B * n; t->tiles = t->R / TR_ROWS; t->n3 = 3 * n;
    t->lr = cfg->learning_rate; t->one_minus_decay = 1.f - cfg->bn_decay;
    t->b1p = 0.9f; t->b2p = 0.999f;
    t->L = make_layout(n);
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    t->max_wgs = cfg->max_workgroups;
    t->grid_bwd = cus * (64 / BWD_ROWS);                  // persistent backward workgroups (one per CU)
    if (t->max_wgs > 0 && t->grid_bwd > t->max_wgs) t->grid_bwd = t->max_wgs;
    t->cus = cus;
    // carve
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t r = off; off += (bytes + 255) / 256 * 256; return r; };
    const size_t P = t->L.count;
    const size_t o_params = take(4 * P), o_grads = take(4 * P), o_m = take(4 * P), o_v = take(4 * P);
    size_t o_mm[ENC_L], o_mv[ENC_L], o_pf[ENC_L], o_pb[ENC_L], o_bn[ENC_L][6], o_act[ENC_L];
    for (int i = 0; i < ENC_L; ++i) {
        o_mm[i] = take(4 * ENC[i + 1]); o_mv[i] = take(4 * ENC[i + 1]);
        o_pf[i] = take(4 * (size_t)ENC[i] * ENC[i + 1]); o_pb[i] = take(4 * (size_t)ENC[i] * ENC[i + 1]);
        for (int k = 0; k < 6; ++k) o_bn[i][k] = take(4 * ENC[i + 1]);
        o_act[i] = take(4 * (size_t)t->R * ENC[i + 1]);
    }
    const size_t o_dy0 = take(4 * (size_t)t->R * 256), o_dy1 = take(4 * (size_t)t->R * 256);
    const size_t o_ps = take(8 * (size_t)t->tiles * 256), o_qs = take(8 * (size_t)(t->R / BF_ROWS) * 256);
    const size_t o_z = take(4 * (size_t)B * 128), o_cnt = take(4 * (size_t)B * 128);   // adjacent: B * 512 B is a multiple of 256
    const size_t o_d1 = take(4 * (size_t)B * 256), o_d2 = take(4 * (size_t)B * 256);
    const size_t o_rec = take(4 * (size_t)B * t->n3), o_gr = take(4 * (size_t)B * t->n3);
    const size_t o_cws = take(4 * chamfer_sym_workspace_floats(1, B, n, n));
    const size_t o_gd = take(4 * (size_t)B), o_dd2 = take(4 * (size_t)B * 256), o_dd1 = take(4 * (size_t)B * 256);
    const size_t o_dz = take(4 * (size_t)B * 128);
    const size_t o_di1 = take(4 * (size_t)B * n), o_di2 = take(4 * (size_t)B * n), o_i1 = take(4 * (size_t)B * n), o_i2 = take(4 * (size_t)B * n);
    const size_t o_dwp = take(4 * (size_t)t->grid_bwd * 256 * 128), o_dbp = take(4 * (size_t)t->grid_bwd * 256);
    const size_t o_loss = take(256);
    const bool emd = cfg->loss == GEOADV_TRAIN_LOSS_EMD;
    const size_t o_et = take(emd ? 4 * geoadv_emd_cost_grad1_temp_floats(B, n, n) + 8 : 0), o_ec = take(emd ? 4 * (size_t)B : 0);
    const size_t o_xb = take(8 * 10 * 512), o_lb = take(8 * 10 * 512);
    t->arena_bytes = off;
    if (hipMalloc(reinterpret_cast<void **>(&t->arena), off) != hipSuccess) {
        set_error("trainer_create: hipMalloc of %zu bytes failed", off);
        delete t;
        return GEOADV_ENOMEM;
    }
    auto F = [&](size_t o) { return reinterpret_cast<float *>(t->arena + o); };
    t->params = F(o_params); t->grads = F(o_grads); t->adam_m = F(o_m); t->adam_v = F(o_v);
    for (int i = 0; i < ENC_L; ++i) {
        t->mov_mean[i] = F(o_mm[i]); t->mov_var[i] = F(o_mv[i]); t->packed_fwd[i] = F(o_pf[i]); t->packed_bwd[i] = F(o_pb[i]);
        t->bn_mean[i] = F(o_bn[i][0]); t->bn_istd[i] = F(o_bn[i][1]); t->bn_scale[i] = F(o_bn[i][2]);
        t->bn_shift[i] = F(o_bn[i][3]); t->bn_m1[i] = F(o_bn[i][4]); t->bn_m2[i] = F(o_bn[i][5]);
        t->act[i] = F(o_act[i]);
    }
    t->dybuf[0] = F(o_dy0); t->dybuf[1] = F(o_dy1);
    t->psum = reinterpret_cast<float2 *>(F(o_ps)); t->qsum = reinterpret_cast<float2 *>(F(o_qs));
    t->zbits = reinterpret_cast<int *>(F(o_z)); t->cnt = reinterpret_cast<int *>(F(o_cnt));
    t->d1 = F(o_d1); t->d2 = F(o_d2); t->recon = F(o_rec); t->g_recon = F(o_gr); t->cham_ws = F(o_cws); t->gd = F(o_gd);
    t->dd2 = F(o_dd2); t->dd1 = F(o_dd1); t->dz = F(o_dz);
    t->dist1 = F(o_di1); t->dist2 = F(o_di2); t->idx1 = reinterpret_cast<int *>(F(o_i1)); t->idx2 = reinterpret_cast<int *>(F(o_i2));
    t->dw_partial = F(o_dwp); t->db_partial = F(o_dbp); t->loss = F(o_loss);
    t->user_loss = nullptr; t->user_recon = nullptr;
    t->loss_type = cfg->loss; t->emd_temp = emd ? F(o_et) : nullptr; t->emd_cost = emd ? F(o_ec) : nullptr;
    t->world = 1;
    t->xbuf = reinterpret_cast<double *>(t->arena + o_xb); t->lbuf = reinterpret_cast<double *>(t->arena + o_lb);
    // upload parameters
    std::vector<float> host(P, 0.f);
    const int dd[4] = {128, 256, 256, 3 * n};
    for (int i = 0; i < ENC_L; ++i) {
        memcpy(&host[t->L.w[i]], hw->enc_w[i], sizeof(float) * ENC[i] * ENC[i + 1]);
        memcpy(&host[t->L.b[i]], hw->enc_b[i], sizeof(float) * ENC[i + 1]);
        memcpy(&host[t->L.gamma[i]], hw->bn_gamma[i], sizeof(float) * ENC[i + 1]);
        memcpy(&host[t->L.beta[i]], hw->bn_beta[i], sizeof(float) * ENC[i + 1]);
    }
    for (int k = 0; k < 3; ++k) {
        memcpy(&host[t->L.v[k]], hw->dec_w[k], sizeof(float) * (size_t)dd[k] * dd[k + 1]);
        memcpy(&host[t->L.c[k]], hw->dec_b[k], sizeof(float) * dd[k + 1]);
    }
    hipError_t e = hipMemcpy(t->params, host.data(), 4 * P, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(t->adam_m, 0, 4 * P);
    if (e == hipSuccess) e = hipMemset(t->adam_v, 0, 4 * P);
    if (e == hipSuccess) e = hipMemset(t->grads, 0, 4 * P);
    for (int i = 0; i < ENC_L && e == hipSuccess; ++i) {
        e = hipMemcpy(t->mov_mean[i], hw->bn_mean[i], 4 * ENC[i + 1], hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(t->mov_var[i], hw->bn_var[i], 4 * ENC[i + 1], hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) {
        std::vector<float> gdv((size_t)B, 1.0f / (float)B);     // d reduce_mean over [B, n] / d dist = (1/B) * (1/n)
        e = hipMemcpy(t->gd, gdv.data(), 4 * gdv.size(), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        set_error("trainer_create: upload failed: %s", hipGetErrorString(e));
        (void)hipFree(t->arena);
        delete t;
        return GEOADV_EHIP;
    }
    if (int rc = trainer_repack(t, nullptr)) { (void)hipFree(t->arena); delete t; return rc; }
    GA_HIP(hipDeviceSynchronize());
    *out = t;
    return GEOADV_OK;
}

extern "C" void geoadv_trainer_destroy(geoadv_trainer *t) {
    if (!t) return;
    (void)hipFree(t->arena);
    delete t;
}

template <int CIN, int COUT>
static int launch_fwd(geoadv_trainer *t, int i, hipStream_t st) {
    FwdArgs a;
    a.zero = nullptr; a.zero_count = 0;
    a.in = t->act[i - 1]; a.pscale = t->bn_scale[i - 1]; a.pshift = t->bn_shift[i - 1];
    a.W = PackedLayer{t->packed_fwd[i], CIN, COUT};
    a.bias = t->params + t->L.b[i]; a.out = t->act[i]; a.psum = t->psum;
    using S = FwdShape<CIN, COUT>;
    static DeviceOnce attr;
    if (int rc = attr.run([]() -> int {
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(train_fwd_kernel<CIN, COUT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::lds_bytes));
            return GEOADV_OK;
        })) return rc;
    // persistent workgroups (as many as stay resident), tiles dealt round-robin
    int grid = t->tiles < S::WGS_PER_CU * t->cus ? t->tiles : S::WGS_PER_CU * t->cus;
    if (t->max_wgs > 0 && grid > t->max_wgs) grid = t->max_wgs;
    a.tiles = t->tiles;
    train_fwd_kernel<CIN, COUT><<<grid, FW_THREADS, S::lds_bytes, st>>>(a);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

// mode: 0 = local statistics in one launch, 1 = this rank's totals into exchange slot `slot`, 2 = statistics from the
// (all-reduced) totals of that slot
static int launch_bn(geoadv_trainer *t, int i, int mode, int slot, hipStream_t st) {
    BnArgs a;
    a.mode = mode; a.totals = t->xbuf + 512 * slot; a.local_totals = t->lbuf + 512 * slot;
    a.psum = t->psum; a.tiles = t->tiles; a.C = ENC[i + 1]; a.inv_rows = 1.0 / ((double)t->R * t->world);
    a.gamma = t->params + t->L.gamma[i]; a.beta = t->params + t->L.beta[i];
    a.mean = t->bn_mean[i]; a.inv_std = t->bn_istd[i]; a.scale = t->bn_scale[i]; a.shift = t->bn_shift[i];
    a.mov_mean = t->mov_mean[i]; a.mov_var = t->mov_var[i]; a.one_minus_decay = t->one_minus_decay;
    bn_finalize_kernel<<<ENC[i + 1] / RED_CH, 1024, 0, st>>>(a);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

// persistent backward workgroups (one per CU) = dW / db partials per layer; rows per BN-gradient partial of the layer below
static int bwd_workgroups(const geoadv_trainer *t) { return (t->R / BF_ROWS) < t->grid_bwd ? (t->R / BF_ROWS) : t->grid_bwd; }
constexpr int BWD_QROWS = BF_ROWS;

template <int CIN, int COUT>
static int launch_bwd(geoadv_trainer *t, int i, const float *dy, float *dy_out, hipStream_t st) {
    BwdArgs a;
    a.tiles = t->R / BWD_ROWS; a.dy = dy; a.a = t->act[i];
    a.mean = t->bn_mean[i]; a.inv_std = t->bn_istd[i]; a.gamma = t->params + t->L.gamma[i]; a.m1 = t->bn_m1[i]; a.m2 = t->bn_m2[i];
    a.aprev = t->act[i - 1];
    a.pscale = t->bn_scale[i - 1]; a.pshift = t->bn_shift[i - 1]; a.pmean = t->bn_mean[i - 1]; a.pinv_std = t->bn_istd[i - 1];
    a.WT = PackedLayer{t->packed_bwd[i], COUT, CIN};
    a.dy_out = dy_out; a.qsum_out = t->qsum; a.dw_partial = t->dw_partial; a.db_partial = t->db_partial;
    using S = FusedShape<CIN, COUT>;
    static DeviceOnce attr;
    if (int rc = attr.run([]() -> int {
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(train_bwd_fused_kernel<CIN, COUT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::lds_bytes));
            return GEOADV_OK;
        })) return rc;
    train_bwd_fused_kernel<CIN, COUT><<<bwd_workgroups(t), S::THREADS, S::lds_bytes, st>>>(a);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;                                   // its partials are reduced by post_layer (next launch)
}

static BnBwdArgs bn_bwd_args(geoadv_trainer *t, int i, int partial_rows, int mode, int slot) {
    BnBwdArgs a;
    a.mode = mode; a.totals = t->xbuf + 512 * slot; a.local_totals = t->lbuf + 512 * slot;
    a.qsum = t->qsum; a.tiles = t->R / partial_rows; a.C = ENC[i + 1]; a.inv_rows = 1.0 / ((double)t->R * t->world);
    a.dbeta = t->grads + t->L.beta[i]; a.dgamma = t->grads + t->L.gamma[i]; a.m1 = t->bn_m1[i]; a.m2 = t->bn_m2[i];
    return a;
}

static int launch_bn_bwd(geoadv_trainer *t, int i, int partial_rows, int mode, int slot, hipStream_t st) {
    bn_bwd_finalize_kernel<<<ENC[i + 1] / RED_CH, 1024, 0, st>>>(bn_bwd_args(t, i, partial_rows, mode, slot));
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

// after the backward kernel of layer i (>= 1): BN-gradient sums of layer i - 1 (mode as in BnBwdArgs) + dW_i + db_i
static int launch_post_layer(geoadv_trainer *t, int i, int mode, int slot, hipStream_t st) {
    const int parts = bwd_workgroups(t);
    const size_t cnt = (size_t)ENC[i] * ENC[i + 1];
    const ReduceArgs dw{t->dw_partial, parts, cnt, t->grads + t->L.w[i]};
    const ReduceArgs db{t->db_partial, parts, (size_t)ENC[i + 1], t->grads + t->L.b[i]};
    const int bn_blocks = ENC[i] / RED_CH, dw_blocks = (int)((cnt + 255) / 256), db_blocks = cdiv(ENC[i + 1], 256);
    post_layer_kernel<<<bn_blocks + dw_blocks + db_blocks, 1024, 0, st>>>(bn_bwd_args(t, i - 1, BWD_QROWS, mode, slot), bn_blocks, dw,
                                                                         dw_blocks, db);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

// One training step = TRAIN_PHASES phases.  Between two phases sits the only cross-row coupling of the encoder: the
// per-channel batch-norm sums.  With one rank a phase ends by turning its per-tile partials into statistics itself
// (sync == false); with synchronised BN over several ranks it ends by leaving this rank's totals in exchange slot
// `phase` (geoadv_trainer_exchange), the host all-reduces them, and the next phase starts from the global totals.
//   phase 0..4 : forward layer i (+ statistics of layer i)
//   phase 5    : pool, decoder, Chamfer, decoder backward, pool backward (+ BN-gradient sums of layer 4)
//   phase 6..9 : backward layer 4..1 (+ BN-gradient sums of the layer below);   phase 10: backward layer 0
constexpr int TRAIN_PHASES = 11;

static int run_phase(geoadv_trainer *t, int phase, const float *x, const float *gt, bool sync, hipStream_t st) {
    const int B = t->B, n = t->N, n3 = t->n3;
    const int end_mode = sync ? 1 : 0;
    if (phase <= 4) {
        const int i = phase;
        if (sync && i > 0)
            if (int rc = launch_bn(t, i - 1, 2, phase - 1, st)) return rc;
        int rc = GEOADV_OK;
        if (i == 0) {
            FwdArgs a;
            a.in = x; a.pscale = a.pshift = nullptr;
            a.W = PackedLayer{t->params + t->L.w[0], 3, 64};
            a.bias = t->params + t->L.b[0]; a.out = t->act[0]; a.psum = t->psum;
            a.zero = t->zbits; a.zero_count = 2 * B * 128;           // zbits and cnt are adjacent; nothing reads them before phase 5
            train_fwd0_kernel<<<t->tiles, TR_THREADS, 0, st>>>(a);
            GA_LAUNCH_CHECK();
        } else if (i == 1) rc = launch_fwd<64, 128>(t, 1, st);
        else if (i == 2) rc = launch_fwd<128, 128>(t, 2, st);
        else if (i == 3) rc = launch_fwd<128, 256>(t, 3, st);
        else rc = launch_fwd<256, 128>(t, 4, st);
        if (rc) return rc;
        return launch_bn(t, i, end_mode, phase, st);
    }
    if (phase == 5) {
        if (sync)
            if (int rc = launch_bn(t, 4, 2, 4, st)) return rc;
        // ---- symmetric max-pool ----
        PoolArgs pa{t->act[4], t->bn_scale[4], t->bn_shift[4], n, t->zbits, t->cnt};
        train_pool_kernel<false><<<t->tiles, 256, 0, st>>>(pa);
        GA_LAUNCH_CHECK();
        const float *z = reinterpret_cast<const float *>(t->zbits);
        // ---- decoder forward ----
        const float *V0 = t->params + t->L.v[0], *V1 = t->params + t->L.v[1], *V2 = t->params + t->L.v[2];
        fc0_and_pool_count_kernel<<<B + t->tiles, 256, 0, st>>>(z, V0, t->params + t->L.c[0], t->d1, B, pa);
        fc1_fwd_kernel<<<B, 1024, 0, st>>>(t->d1, V1, t->params + t->L.c[1], t->d2);
        fc_out_fwd_kernel<<<dim3(cdiv(n3, 64), cdiv(B, 16)), 256, 0, st>>>(t->d2, V2, t->params + t->L.c[2], t->recon, t->user_recon, B, n3);
        GA_LAUNCH_CHECK();
        DecOutBwdArgs da;
        da.inv = 1.0 / ((double)B * t->world * n);   // reduce_mean over the batch of ALL ranks (the host adds the ranks up)
        da.loss = t->loss; da.loss2 = t->user_loss; da.count = (size_t)B * n;
        if (t->loss_type == GEOADV_TRAIN_LOSS_EMD) {
            // ---- approx-EMD loss (pointnet_ae.py:77-79): reduce_mean over the clouds of match_cost(recon, gt, match), and
            // d loss / d recon = match_cost_grad's grad1 / clouds with the match held constant (tf_approxmatch.py:19, 44-50).
            // The plan is never stored (geoadv_emd_cost_grad1: levels + cost + gradient fused). ----
            if (int rc = geoadv_emd_cost_grad1_mode(GEOADV_EMD_FAST, B, n, n, t->recon, gt, t->emd_cost, t->g_recon, t->emd_temp, st)) return rc;
            emd_loss_scale_kernel<<<256, 256, 0, st>>>(t->emd_cost, B, 1.0 / ((double)B * t->world), t->loss, t->user_loss, t->g_recon, (size_t)B * n3);
            GA_LAUNCH_CHECK();
            da.count = 0; da.loss = t->loss + 1; da.loss2 = nullptr;     // (the loss block of the launch below then writes a zero beside the real loss)
        } else {
            // ---- Chamfer loss and its gradient w.r.t. the reconstruction ----
            // one distance evaluation per pair serves both directions (chamfer_sym.hip); dist/idx bit-identical to nn_distance
            const ChamferPair cp{t->recon, gt, t->dist1, t->idx1, t->dist2, t->idx2};
            if (int rc = launch_chamfer_sym(&cp, 1, B, n, n, t->cham_ws, st)) return rc;
            // NnDistanceGrad w.r.t. the reconstruction only (order-independent fixed-point accumulation, attack.hip)
            const CGradProblem gp{t->recon, gt, t->idx1, t->idx2, t->g_recon, t->gd, nullptr, 0.f};
            if (int rc = launch_chamfer_grad(&gp, 1, B, n, st)) return rc;
        }
        // ---- decoder backward: three launches, each = weight gradient + data gradient of one layer side by side ----
        da.d2 = t->d2; da.g = t->g_recon; da.V2 = V2; da.dV2 = t->grads + t->L.v[2]; da.dc2 = t->grads + t->L.c[2]; da.dd2 = t->dd2;
        da.batch = B; da.n3 = n3; da.dist1 = t->dist1; da.dist2 = t->dist2;
        dec_out_bwd_kernel<<<cdiv(n3, DOW_COLS) * 8 + (256 / DOX_K) * cdiv(B, DOX_B) + 1, 256, sizeof(float) * B * 32, st>>>(da);
        fc_bwd_kernel<true><<<256 + B, 256, 0, st>>>(t->d1, t->dd2, V1, t->grads + t->L.v[1], t->grads + t->L.c[1], t->dd1, B, 256);
        fc_bwd_kernel<false><<<128 + B, 256, 0, st>>>(z, t->dd1, V0, t->grads + t->L.v[0], t->grads + t->L.c[0], t->dz, B, 128);
        GA_LAUNCH_CHECK();
        // ---- encoder backward starts: pool gradient + ReLU of layer 4 ----
        PoolBwdArgs pb{t->act[4], t->bn_scale[4], t->bn_shift[4], t->bn_mean[4], t->bn_istd[4], n, t->zbits, t->cnt, t->dz,
                       t->dybuf[0], t->qsum};
        train_pool_bwd_kernel<<<t->tiles, 256, 0, st>>>(pb);
        GA_LAUNCH_CHECK();
        return launch_bn_bwd(t, 4, TR_ROWS, end_mode, phase, st);
    }
    if (phase <= 9) {
        const int i = 10 - phase;                                  // layer whose backward runs: 4, 3, 2, 1
        if (sync)
            if (int rc = launch_bn_bwd(t, i, i == 4 ? TR_ROWS : BWD_QROWS, 2, phase - 1, st)) return rc;
        int rc;
        if (i == 4) rc = launch_bwd<256, 128>(t, 4, t->dybuf[0], t->dybuf[1], st);
        else if (i == 3) rc = launch_bwd<128, 256>(t, 3, t->dybuf[1], t->dybuf[0], st);
        else if (i == 2) rc = launch_bwd<128, 128>(t, 2, t->dybuf[0], t->dybuf[1], st);
        else rc = launch_bwd<64, 128>(t, 1, t->dybuf[1], t->dybuf[0], st);
        if (rc) return rc;
        return launch_post_layer(t, i, end_mode, phase, st);
    }
    if (sync)
        if (int rc = launch_bn_bwd(t, 0, BWD_QROWS, 2, 9, st)) return rc;
    BwdArgs a = {};
    a.tiles = t->tiles; a.dy = t->dybuf[0]; a.a = t->act[0];
    a.mean = t->bn_mean[0]; a.inv_std = t->bn_istd[0]; a.gamma = t->params + t->L.gamma[0]; a.m1 = t->bn_m1[0]; a.m2 = t->bn_m2[0];
    a.aprev = x; a.dw_partial = t->dw_partial; a.db_partial = t->db_partial;
    const int grid0 = t->tiles < t->grid_bwd ? t->tiles : t->grid_bwd;
    train_bwd0_kernel<<<grid0, TR_THREADS, 0, st>>>(a);
    post_layer_kernel<<<2, 1024, 0, st>>>(BnBwdArgs{}, 0, ReduceArgs{t->dw_partial, grid0, 192, t->grads + t->L.w[0]}, 1,
                                          ReduceArgs{t->db_partial, grid0, 64, t->grads + t->L.b[0]});
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

extern "C" int geoadv_trainer_forward_backward(geoadv_trainer *t, const float *x, const float *gt, float *loss, float *recon,
                                               void *stream) {
    GA_REQUIRE(t && x, "trainer_forward_backward: null argument");
    GA_REQUIRE(t->world == 1, "trainer_forward_backward: synchronised batch norm is on (world %d): drive the phases", t->world);
    hipStream_t st = as_stream(stream);
    if (!gt) gt = x;
    // the caller's loss / reconstruction are written by the kernels that produce them (two copy launches less at the end of a step)
    t->user_loss = loss; t->user_recon = recon;
    int rc = GEOADV_OK;
    for (int p = 0; p < TRAIN_PHASES && rc == GEOADV_OK; ++p) rc = run_phase(t, p, x, gt, false, st);
    t->user_loss = nullptr; t->user_recon = nullptr;
    return rc;
}

// ---- synchronised batch norm over `world` ranks: the caller runs the phases and all-reduces the exchange slots ----
extern "C" int geoadv_trainer_set_world(geoadv_trainer *t, int world) {
    GA_REQUIRE(t && world >= 1 && world <= 4096, "trainer_set_world: bad arguments");
    t->world = world;
    std::vector<float> gdv((size_t)t->B, 1.0f / ((float)t->B * (float)world));      // d reduce_mean over the GLOBAL batch
    GA_HIP(hipMemcpy(t->gd, gdv.data(), 4 * gdv.size(), hipMemcpyHostToDevice));
    return GEOADV_OK;
}

extern "C" int geoadv_trainer_num_phases(void) { return TRAIN_PHASES; }

extern "C" int geoadv_trainer_run_phase(geoadv_trainer *t, int phase, const float *x, const float *gt, void *stream) {
    GA_REQUIRE(t && x && phase >= 0 && phase < TRAIN_PHASES, "trainer_run_phase: bad arguments");
    return run_phase(t, phase, x, gt ? gt : x, true, as_stream(stream));
}

// What must be summed over the ranks after `phase` before the next one starts: *count doubles at *buf (device), or 0.
extern "C" int geoadv_trainer_exchange(geoadv_trainer *t, int phase, double **buf, size_t *count) {
    GA_REQUIRE(t && buf && count && phase >= 0 && phase < TRAIN_PHASES, "trainer_exchange: bad arguments");
    *buf = t->xbuf + 512 * (phase < 10 ? phase : 0);
    if (phase <= 4) *count = 2 * (size_t)ENC[phase + 1];
    else if (phase <= 9) *count = 2 * (size_t)ENC[10 - phase];            // sums of layer 4, 3, 2, 1, 0
    else *count = 0;
    return GEOADV_OK;
}

// loss (this rank's share of the global mean when world > 1) and reconstruction of the last step
extern "C" int geoadv_trainer_fetch(geoadv_trainer *t, float *loss, float *recon, void *stream) {
    GA_REQUIRE(t, "trainer_fetch: null trainer");
    hipStream_t st = as_stream(stream);
    if (loss) GA_HIP(hipMemcpyAsync(loss, t->loss, sizeof(float), hipMemcpyDeviceToDevice, st));
    if (recon) GA_HIP(hipMemcpyAsync(recon, t->recon, sizeof(float) * (size_t)t->B * t->n3, hipMemcpyDeviceToDevice, st));
    return GEOADV_OK;
}

extern "C" int geoadv_trainer_apply(geoadv_trainer *t, float grad_scale, void *stream) {
    GA_REQUIRE(t, "trainer_apply: null trainer");
    hipStream_t st = as_stream(stream);
    const size_t P = t->L.count;
    AdamPack pk;
    for (int i = 1; i < ENC_L; ++i) {
        pk.off[i - 1] = t->L.w[i]; pk.K[i - 1] = ENC[i]; pk.N[i - 1] = ENC[i + 1];
        pk.fwd[i - 1] = t->packed_fwd[i]; pk.bwd[i - 1] = t->packed_bwd[i];
    }
    train_adam_kernel<<<(unsigned)((P + 255) / 256), 256, 0, st>>>(t->params, t->adam_m, t->adam_v, t->grads, P, grad_scale, t->lr,
                                                                   t->b1p, t->b2p, pk);
    GA_LAUNCH_CHECK();
    t->b1p *= 0.9f; t->b2p *= 0.999f;
    return GEOADV_OK;
}

extern "C" int geoadv_trainer_step(geoadv_trainer *t, const float *x, const float *gt, float *loss, float *recon, void *stream) {
    if (int rc = geoadv_trainer_forward_backward(t, x, gt, loss, recon, stream)) return rc;
    return geoadv_trainer_apply(t, 1.0f, stream);
}

extern "C" int geoadv_trainer_buffers(geoadv_trainer *t, float **params, float **grads, size_t *count) {
    GA_REQUIRE(t, "trainer_buffers: null trainer");
    if (params) *params = t->params;
    if (grads) *grads = t->grads;
    if (count) *count = t->L.count;
    return GEOADV_OK;
}

extern "C" int geoadv_trainer_export(geoadv_trainer *t, const geoadv_ae_weights *dst, void *stream) {
    GA_REQUIRE(t && dst, "trainer_export: null argument");
    GA_HIP(hipStreamSynchronize(as_stream(stream)));
    std::vector<float> host(t->L.count);
    GA_HIP(hipMemcpy(host.data(), t->params, 4 * host.size(), hipMemcpyDeviceToHost));
    const int dd[4] = {128, 256, 256, t->n3};
    for (int i = 0; i < ENC_L; ++i) {
        GA_REQUIRE(dst->enc_w[i] && dst->enc_b[i] && dst->bn_gamma[i] && dst->bn_beta[i] && dst->bn_mean[i] && dst->bn_var[i],
                   "trainer_export: null destination at encoder layer %d", i);
        memcpy(const_cast<float *>(dst->enc_w[i]), &host[t->L.w[i]], sizeof(float) * ENC[i] * ENC[i + 1]);
        memcpy(const_cast<float *>(dst->enc_b[i]), &host[t->L.b[i]], sizeof(float) * ENC[i + 1]);
        memcpy(const_cast<float *>(dst->bn_gamma[i]), &host[t->L.gamma[i]], sizeof(float) * ENC[i + 1]);
        memcpy(const_cast<float *>(dst->bn_beta[i]), &host[t->L.beta[i]], sizeof(float) * ENC[i + 1]);
        GA_HIP(hipMemcpy(const_cast<float *>(dst->bn_mean[i]), t->mov_mean[i], 4 * ENC[i + 1], hipMemcpyDeviceToHost));
        GA_HIP(hipMemcpy(const_cast<float *>(dst->bn_var[i]), t->mov_var[i], 4 * ENC[i + 1], hipMemcpyDeviceToHost));
    }
    for (int k = 0; k < 3; ++k) {
        GA_REQUIRE(dst->dec_w[k] && dst->dec_b[k], "trainer_export: null destination at decoder layer %d", k);
        memcpy(const_cast<float *>(dst->dec_w[k]), &host[t->L.v[k]], sizeof(float) * (size_t)dd[k] * dd[k + 1]);
        memcpy(const_cast<float *>(dst->dec_b[k]), &host[t->L.c[k]], sizeof(float) * dd[k + 1]);
    }
    return GEOADV_OK;
}

// Offsets (in floats) of every variable inside the flat parameter / gradient buffers, in the order
// enc_w[5], enc_b[5], bn_gamma[5], bn_beta[5], dec_w[3], dec_b[3]  (26 entries)
extern "C" int geoadv_trainer_layout(const geoadv_trainer *t, size_t *offsets26) {
    GA_REQUIRE(t && offsets26, "trainer_layout: null argument");
    for (int i = 0; i < ENC_L; ++i) {
        offsets26[i] = t->L.w[i]; offsets26[5 + i] = t->L.b[i]; offsets26[10 + i] = t->L.gamma[i]; offsets26[15 + i] = t->L.beta[i];
    }
    for (int k = 0; k < 3; ++k) { offsets26[20 + k] = t->L.v[k]; offsets26[23 + k] = t->L.c[k]; }
    return GEOADV_OK;
}
GA_STAMPS_GETTER(geoadv_debug_stamps_train)
