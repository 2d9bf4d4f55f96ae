// The encoder's fp32 products on the bf16 matrix pipe ("x3" arithmetic).
//
// v_mfma_f32_32x32x2_f32 runs at the VALU's rate on gfx950 (1/16 of the bf16 forms), so the fp32 encoder is bounded by
// 157 TFLOP/s.  Here every operand is written as three bf16 pieces, x = x0 + x1 + x2 with
//     x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16((x - x0) - x1)      (round to nearest even; both subtractions are exact)
// -- 8 + 8 + 8 significant bits, i.e. all 24 of an fp32 -- and a product a * w as the six piece products of weight >= 2^-16,
//     a2 w0, a1 w1, a0 w2, a1 w0, a0 w1, a0 w0                          (smallest first; a1 w2, a2 w1, a2 w2 <= 2^-24 dropped)
// each of which is EXACT in the matrix pipe's fp32 accumulator (8 x 8 bits), on v_mfma_f32_32x32x16_bf16.  Measured against
// float64 (tools/bf16x3_probe.py, profiles/r05_bf16x3_probe.jsonl): rms error 0.48-0.52 units of 2^-24 |a|.|w| for K = 64 ...
// 256 against 0.43-0.46 for the fp32 MFMA chain and 0.45-0.48 for a host fp32 multiply-add loop -- the error of an fp32
// accumulation in another order, not that of a reduced-precision product (two pieces per operand: 10-20 units).
//
// What makes results reproducible (the recomputing backward compares its h5 with the forward's z for equality): every
// output element is ONE chain of MFMAs -- sixteen-k blocks in ascending order, the six piece products in the order above --
// with a fixed assignment of input channels to the instruction's k slots (x3_in_channel) and fixed operand roles per layer
// (layers 1-3: A = weights, B = activations; layer 4: A = activations, B = weights).  A point's result does not depend on
// which other points share its tile.
//
// The f16x2 arithmetic (round 6) is the same construction with TWO fp16 pieces per operand -- 11 + 11 significant bits, the
// remainder below 2^-23 |x| -- and the three piece products of weight >= 2^-11 (a1 w0, a0 w1, a0 w0; a1 w1 <= 2^-22 dropped), each
// exact in the fp32 accumulator (11 x 11 bits), on v_mfma_f32_32x32x16_f16: HALF the matrix instructions.  fp16 has five exponent
// bits, so the operands are scaled by powers of two (exact): layer j's activations by s_j = 2^6 / (the power of two nearest to
// the rms of sqrt(gamma^2 + beta^2) over the layer's batch-norm channels -- what relu(gamma z + beta) puts out; 2^6 exactly for
// gamma = 1, beta = 0: ae.hip) -- the second piece of every scaled activation >= 2^-3 is a normal fp16 number, smaller ones
// keep an ABSOLUTE error <= 2^-25 / s_j, the pipe honours fp16 subnormals --, a layer's weights so that the largest magnitude
// lies in [2^13, 2^14).  The scales ride in the constants of the epilogue (scale' = scale s_L / (s_{L-1} S_w), shift' = s_L
// shift: the next layer's scaled activation comes out of the same fma, bit for bit s_L times the unscaled one).  Measured against float64 (tools/bf16x3_probe.py modes 5-7, profiles/r06_f16x2_probe.jsonl): rms error
// 0.44-0.52 units of 2^-24 |a|.|w| -- the fp32 chain's.  RANGE: a scaled activation above 65504 (an activation >= 1023.5 x its
// layer's batch-norm magnitude) would
// round to an fp16 infinity; every epilogue keeps a running maximum (one v_max3_f32 per two values) and a workgroup that sees
// one poisons its pool partial with +inf and raises DeviceAE::range_flag (geoadv_ae_status: GEOADV_ERANGE) -- never silent.
#pragma once
#include "ae.h"
#include "mfma_tile.h"
#include <type_traits>

namespace geoadv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int X3_STEPS = 44;                  // sixteen-k steps of one 32-point unit: 4 (layer 1) + 8 (layer 2) + 2 x (8 + 8) (layers 3 + 4 by halves)
constexpr int X3_FRAG_WORDS = 64 * 4;         // one fragment: 64 lanes x 16 bytes (8 bf16 k-slots per lane)
// NP = pieces per operand: 3 (bf16x3) or 2 (f16x2)
__host__ __device__ constexpr int xp_step_words(int np) { return 4 * np * X3_FRAG_WORDS; }   // 4 output-channel blocks x NP pieces: 12 / 8 KiB
__host__ __device__ constexpr size_t xp_image_words(int np) { return (size_t)X3_STEPS * xp_step_words(np); }
constexpr int X3_STEP_WORDS = xp_step_words(3);
constexpr size_t X3_IMAGE_WORDS = xp_image_words(3), H2_IMAGE_WORDS = xp_image_words(2);
constexpr float H2_ACT_LIMIT = 65504.f;       // largest finite fp16: a scaled activation above it trips the range guard
// the constants the forward keeps in LDS, as one block (DeviceAE::enc_x3_consts): W0 [3][64], scale0 [64], shift0 [64], then
// [scale | shift] of layers 1 (128 + 128), 2 (128 + 128), 3 (256 + 256), 4 (128 + 128)
constexpr int X3_CONST_FLOATS = 320 + 2 * (128 + 128 + 256 + 128);
constexpr int X3_SC1 = 320, X3_SC2 = X3_SC1 + 256, X3_SC3 = X3_SC2 + 256, X3_SC4 = X3_SC3 + 512;

// Step of the weight image that holds layer L's fragments of output-channel block ob (32 channels), sixteen-k block kb; the
// fragment's position inside the step is (ob & 3).  Order = the order the forward consumes them in.
__host__ __device__ constexpr int x3_step_of(int L, int ob, int kb) {
    return L == 1 ? kb : L == 2 ? 4 + kb : L == 3 ? (ob < 4 ? 12 : 28) + kb : (kb < 8 ? 20 + kb : 36 + (kb - 8));
}
// Input channel held in k slot j (0..7) of lane half h (0, 1) of sixteen-k block kb.  Layer 1 reads layer 0's output, which the
// VALU writes in natural order; layers 2-4 read an MFMA result straight from the accumulator registers, whose lane (point,
// h) holds channels 32 c + 8 g + 4 h + u (g, u = 0..3) of channel block c: sixteen-k block kb = 2 c + (g >> 1) takes them as
// slot j = 4 (g & 1) + u -- no lane exchange between layers.
__host__ __device__ constexpr int x3_in_channel(int L, int kb, int h, int j) {
    return L == 1 ? 16 * kb + 8 * h + j : 32 * (kb >> 1) + 8 * (2 * (kb & 1) + (j >> 2)) + 4 * h + (j & 3);
}

template <int NP> struct XP { u32x4 p[NP]; }; // the pieces of a lane's 8 k slots (bf16 / fp16 pairs in 32-bit words)
using X3 = XP<3>;
using H2 = XP<2>;
__device__ __forceinline__ const unsigned *xp_image(const DeviceAE &A, std::integral_constant<int, 3>) { return A.enc_x3; }
__device__ __forceinline__ const unsigned *xp_image(const DeviceAE &A, std::integral_constant<int, 2>) { return A.enc_h2; }
__device__ __forceinline__ const float *xp_consts(const DeviceAE &A, std::integral_constant<int, 3>) { return A.enc_x3_consts; }
__device__ __forceinline__ const float *xp_consts(const DeviceAE &A, std::integral_constant<int, 2>) { return A.enc_h2_consts; }

__device__ __forceinline__ unsigned x3_cvt_pk(float lo, float hi) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f2){lo, hi}, bf16x2));      // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
}
// pieces of two values -> word `w` of each piece
__device__ __forceinline__ void x3_split_pair(float a, float b, unsigned &w0, unsigned &w1, unsigned &w2) {
    w0 = x3_cvt_pk(a, b);
    const float ra = a - __uint_as_float(w0 << 16), rb = b - __uint_as_float(w0 & 0xffff0000u);
    w1 = x3_cvt_pk(ra, rb);
    const float sa = ra - __uint_as_float(w1 << 16), sb = rb - __uint_as_float(w1 & 0xffff0000u);
    w2 = x3_cvt_pk(sa, sb);
}
__device__ __forceinline__ unsigned h2_cvt_pk(float lo, float hi) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f2){lo, hi}, f16x2));        // v_cvt_pk_f16_f32: RNE
}
// the same for the two fp16 pieces of two (already scaled) values; gmax: the running maximum of the range guard
__device__ __forceinline__ void h2_split_pair(float a, float b, unsigned &w0, unsigned &w1, float &gmax) {
    gmax = fmaxf(gmax, fmaxf(a, b));                                                          // v_max3_f32
    w0 = h2_cvt_pk(a, b);
    const f16x2 h = __builtin_bit_cast(f16x2, w0);
    w1 = h2_cvt_pk(a - (float)h[0], b - (float)h[1]);
}
// word `w` of every piece of a lane's fragment from two values
__device__ __forceinline__ void xp_split_pair(float a, float b, X3 &out, int w, float &) {
    unsigned p0, p1, p2;
    x3_split_pair(a, b, p0, p1, p2);
    out.p[0][w] = p0; out.p[1][w] = p1; out.p[2][w] = p2;
}
__device__ __forceinline__ void xp_split_pair(float a, float b, H2 &out, int w, float &gmax) {
    unsigned p0, p1;
    h2_split_pair(a, b, p0, p1, gmax);
    out.p[0][w] = p0; out.p[1][w] = p1;
}
template <int NP>
__device__ __forceinline__ void xp_split8(const float (&v)[8], XP<NP> &out, float &gmax) {
#pragma unroll
    for (int w = 0; w < 4; ++w) xp_split_pair(v[2 * w], v[2 * w + 1], out, w, gmax);
}

// acc += a . w over the 16 k slots: the six piece products, smallest first.  ACT_IS_A: the activations are the A operand
// (rows = points; layer 4, whose result is pooled over points in registers), else the weights are (rows = channels).
template <bool ACT_IS_A>
__device__ __forceinline__ void xp_mfma(const X3 &w, const X3 &a, f32x16 &acc) {
    constexpr int WQ[6] = {0, 1, 2, 0, 1, 0}, AQ[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const bf16x8 wf = __builtin_bit_cast(bf16x8, w.p[WQ[t]]), af = __builtin_bit_cast(bf16x8, a.p[AQ[t]]);
        acc = ACT_IS_A ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wf, acc, 0, 0, 0)
                       : __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, af, acc, 0, 0, 0);
    }
}

// the three piece products of the f16x2 arithmetic, smallest first
template <bool ACT_IS_A>
__device__ __forceinline__ void xp_mfma(const H2 &w, const H2 &a, f32x16 &acc) {
    constexpr int WQ[3] = {0, 1, 0}, AQ[3] = {1, 0, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const f16x8 wf = __builtin_bit_cast(f16x8, w.p[WQ[t]]), af = __builtin_bit_cast(f16x8, a.p[AQ[t]]);
        acc = ACT_IS_A ? __builtin_amdgcn_mfma_f32_32x32x16_f16(af, wf, acc, 0, 0, 0)
                       : __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, af, acc, 0, 0, 0);
    }
}

// A lane's fragment of (step, block c, piece q) in the global weight image.
template <int NP>
__device__ __forceinline__ const u32x4 *xp_frag_ptr(const unsigned *img, int step, int c, int q, int lane) {
    return reinterpret_cast<const u32x4 *>(img + ((size_t)(step * 4 + c) * NP + q) * X3_FRAG_WORDS) + lane;
}

// ------------------------------------------------------------------------------------------
// The same layers for a tile of fp32 activations in LDS (the recomputing backward, encoder.hip: rare -- clouds with a
// tied pool maximum -- and the recompute_backward A/B of the tests): the bits of the forward kernel, at no particular
// speed.  out[row][c] for ROWS rows (a multiple of 32); units (channel block, row block) dealt to the 8 waves; weights
// straight from the global image.  epi(row, channel, value).  Must be called by every wave; no barrier inside.
// ------------------------------------------------------------------------------------------
// NP = 2: the rows are scaled by `in_scale` = s_{L-1} on the way into the split and the result by `unscale` = 1 / (s_{L-1} S_w(L))
// on the way out (powers of two: what the forward's folded constants do, bit for bit); NP = 3: both 1.
template <int NP, int L, int ROWS, class Epi>
__device__ __forceinline__ void xp_layer_lds(const float *in, int s_in, const unsigned *img, float in_scale, float unscale, Epi epi) {
    constexpr int K = L == 1 ? 64 : L == 4 ? 256 : 128, NOUT = L == 3 ? 256 : 128;
    constexpr int OB = NOUT / 32, RB = ROWS / 32, UNITS = OB * RB;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
    for (int unit = wave; unit < UNITS; unit += ENC_THREADS / 64) {
        const int ob = unit % OB, rb = unit / OB;
        const float *row = in + (rb * 32 + p) * s_in;
        f32x16 acc = {};
        for (int kb = 0; kb < K / 16; ++kb) {
            float v[8];
            if (L == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = row[16 * kb + 8 * h + j];
            } else {
                // layer 4's second K half reads channels 128 .. 255: block kb - 8 of the second 128
                const int base = (L == 4 && kb >= 8 ? 128 : 0), k7 = kb & 7;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = row[base + x3_in_channel(L, k7, h, j)];
            }
            if (NP == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= in_scale;
            }
            XP<NP> a;
            float gmax = 0.f;                  // (the forward that made these activations already checked their range)
            xp_split8(v, a, gmax);
            XP<NP> w;
            const int step = x3_step_of(L, ob, kb);
#pragma unroll
            for (int q = 0; q < NP; ++q) w.p[q] = *xp_frag_ptr<NP>(img, step, ob & 3, q, lane);
            xp_mfma<L == 4>(w, a, acc);
        }
        if (NP == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] *= unscale;
        }
        if (L == 4) {   // rows = points, column = this lane's channel
#pragma unroll
            for (int r = 0; r < 16; ++r) epi(rb * 32 + acc_row(r, h), ob * 32 + p, acc[r]);
        } else {        // rows = channels, column = this lane's point
#pragma unroll
            for (int r = 0; r < 16; ++r) epi(rb * 32 + p, ob * 32 + acc_row(r, h), acc[r]);
        }
    }
}

int encoder_x3_points(int b, int n);          // points per workgroup (= per pool partial) of the x3 / f16x2 forward for b clouds of n points
int launch_encoder_fwd_x3(const DeviceAE &A, int b, const float *x, const float *pert, float *adv_out, float *pmax, int *parg,
                          int *pcnt, unsigned *masks, hipStream_t stream, hipEvent_t start, hipEvent_t stop, const FusedAdam &fa);

}  // namespace geoadv
