// Device-side description of the victim auto-encoder (internal to libgeoadv.so).
#pragma once
#include <atomic>
#include "common.h"

namespace geoadv {

constexpr int ENC_L = GEOADV_ENC_LAYERS;   // 5 per-point layers

// Packed weight fragments for v_mfma_f32_32x32x2_f32.  For a layer computing out[r][n] =
// sum_k in[r][k] * W[k][n] (K = fan-in, a multiple of 8; N padded to a multiple of 32):
//   packed[((cb * K/8 + t) * 64 + lane) * 4 + u] = W[8t + 4*(lane>>5) + u][32cb + (lane&31)]
// so one wave fetches the B operands of four consecutive MFMA k-steps for column block cb with a
// single coalesced 1 KiB global_load_dwordx4.
struct PackedLayer {
    const float *w;   // packed fragments (device)
    int K, N;         // fan-in, (padded) fan-out
};
// The same for v_mfma_f32_16x16x4_f32 (16-row tiles: the masked encoder backward): K a multiple of 16, N of 16,
//   packed16[((cb * K/16 + t) * 64 + lane) * 4 + u] = W[16t + 4*(lane>>4) + u][16cb + (lane&15)]
// -- again one coalesced 1 KiB load per wave for the B operands of four consecutive MFMA k-steps of column block cb.

struct DeviceAE {
    int n_points, bneck;
    int enc_dims[ENC_L + 1];
    int dec_dims[GEOADV_DEC_LAYERS + 1];
    // encoder: layer 0 (fan-in 3) stays un-packed, [3][C1] row-major, plus its transpose use
    const float *w0;               // [3][C1]
    PackedLayer enc_fwd[ENC_L];    // [1..4] used: in[r][C_i] -> [C_{i+1}]
    PackedLayer enc_bwd[ENC_L];    // [1..4] used: W_i^T : [C_{i+1}] -> [C_i]
    PackedLayer enc_bwd16[ENC_L];  // the same products packed for the 16x16x4 shape
    const unsigned *enc_x3;        // encoder layers 1-4 as bf16 piece fragments in step order (encoder_x3.h)
    const float *enc_x3_consts;    // the x3 forward's LDS constants as one block (encoder_x3.h: X3_CONST_FLOATS)
    const unsigned *enc_h2;        // the same as fp16 piece fragments of the SCALED weights (encoder_x3.h, f16x2); null if the model's
    const float *enc_h2_consts;    // weights / BN constants do not scale exactly (then the arithmetic is refused); its LDS constants
    float h2_act_scale[ENC_L];     // [0..3]: s_j, the power of two layer j's activations are carried times (2^6 x the layer's batch-norm magnitude)
    float h2_unscale[ENC_L];       // [1..4]: 1 / (s_{L-1} S_w(L)), the power of two that takes an f16x2 accumulator of layer L back
    int *range_flag;               // device int, sticky: an f16x2 forward saw an activation beyond the fp16 range (geoadv_ae_status)
    int enc_arith;                 // GEOADV_ENC_ARITH_*: which forward (and recompute) arithmetic the encoder kernels use
    const float *scale[ENC_L];     // BN folded: h = max(a*scale + shift, 0), a = x@W (no bias)
    const float *shift[ENC_L];     // shift = b*scale + (beta - mean*scale)
    // decoder
    const float *v0, *c0;          // [bneck][256], [256]
    const float *v1, *c1;          // [256][256], [256]
    const float *v0t, *v1t;        // transposes for the backward pass: [256][bneck], [256][256]
    PackedLayer dec2_fwd;          // V2: K = 256 -> N = 3*n_points (padded to 32)
    PackedLayer dec2_bwd;          // V2^T: K = 3*n_points (padded to 8) -> N = 256
    const float *c2;               // [3*n_points]
};

// The attack's Adam step on `pert` folded into the NEXT encoder forward (attack.hip): the forward's point loaders update
// their own coordinate first.  m == nullptr: off.  Same arithmetic as adam_kernel, element by element.
struct FusedAdam {
    float *pert, *m, *v, *g_enc;     // g_enc is read and zeroed (the next backward scatters into it)
    const float *g_dist;
    float *grad_out;                 // optional copy of the total gradient (tests)
    float alpha, one_minus_b1, one_minus_b2, eps;
};

}  // namespace geoadv

namespace geoadv { int ae_range_check(const geoadv_ae *ae, hipStream_t stream, const char *who); }

struct geoadv_ae {
    geoadv::DeviceAE d;
    mutable std::atomic<int> attack_refs{0};   // live attack handles on this model (they cache a forward in d.enc_arith's layout)
    void *arena;       // one device allocation holding everything above
    size_t arena_bytes;
};
