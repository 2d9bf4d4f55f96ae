// Pool Jacobian of the encoder, for the attack's sparse backward (src/adv_ae.py:153 differentiates the loss through the
// encoder; src/encoders_decoders.py:37-72 is what is differentiated).
//
// The max-pool sends gradient to ONE point per latent channel (tie-free clouds), and with frozen weights and the forward's
// ReLU masks the way from a latent channel back to its point is a fixed linear map.  So instead of back-propagating dz
// when it finally arrives -- five dependent layers, 11-13 us, in the middle of the iteration's critical path -- the map
// itself is evaluated as soon as the forward knows the critical points:
//     J[b][c][:] = d z[b][c] / d adv[b][crit[b][c]]   (3 numbers per channel; 0 where z[b][c] <= 0: ReLU at the pool input)
// which needs nothing from the decoder, the losses or their backward and therefore runs BESIDE them (as extra workgroups of
// the Chamfer scan's launch, or as a short launch of its own).  The backward then is g_enc[b][p] = sum over the channels c
// with crit[b][c] == p of dz[b][c] * J[b][c] -- 128 x 3 multiply-adds per cloud, done by the decoder backward's last
// kernel (decoder.hip).  Same products as the masked backward (encoder.hip: encoder_bwd_masked_body), summed per channel
// first instead of per point first: equal to rounding.  Clouds with a tied maximum (TF's _MinOrMaxGrad splits the
// gradient equally) do not use J: they keep the dense recomputing backward.
//
// A workgroup (8 waves) owns 16 channels = 16 rows of the 16x16x4 MFMA shape.  The first layer of the way back has a
// one-hot input, so it is a row gather from the packed weights, not a product: dh4[s][:] = scale4[c] * W4^T[c][:].
#pragma once
#include "mfma_tile.h"

namespace geoadv {

// ReLU masks of h1..h4 the forward keeps per point (encoder.hip writes them): MASK_WORDS 32-bit words,
// word 0-1: h1 (64 channels), MASK_OFF2..: h2 (128), MASK_OFF3..: h3 (128), MASK_OFF4..: h4 (256)
constexpr int MASK_WORDS = 18;
constexpr int MASK_OFF2 = 2, MASK_OFF3 = 6, MASK_OFF4 = 10;

constexpr int JAC_ROWS = 16;
constexpr int JAC_SCALES = 64 + 128 + 128 + 256;       // scale0 .. scale3, staged once
constexpr size_t JAC_LDS_BYTES = sizeof(float) * (JAC_ROWS * 260 + JAC_ROWS * 132 + JAC_SCALES + 192 + JAC_ROWS) +
                                 sizeof(int) * JAC_ROWS + sizeof(unsigned) * JAC_ROWS * MASK_WORDS;

struct JacArgs {
    int n;                      // points per cloud
    const unsigned *masks;      // [b][n][MASK_WORDS]
    const int *crit;            // [b][128] lowest point attaining the pool maximum
    const float *z;             // [b][128]
    const int *dense_flag;      // [b] clouds with a tied maximum: skipped
    float *jac;                 // [b][128][3]
};

// channels [16 bx, 16 bx + 16) of cloud b; lds: JAC_LDS_BYTES, 16-byte aligned; all 8 waves (512 threads) take part.
// AHEAD: request every layer's weights one layer ahead of their use (a launch of its own: 145 VGPRs) or just in time (as a
// rider of the Chamfer scan, whose 128-register budget it must respect -- 14 spilled registers made that whole launch use
// scratch and twice as long; the extra L2 round trips are hidden there anyway).
template <bool AHEAD>
__device__ __forceinline__ void encoder_jac_block(const DeviceAE &A, const JacArgs &a, float *lds, const int bx, const int b) {
    constexpr int ROWS = JAC_ROWS;
    float *bufP = lds;                                  // [16][260]
    float *bufQ = bufP + ROWS * 260;                    // [16][132]
    float *sc0 = bufQ + ROWS * 132;                     // BN scales of layers 0..3
    float *sc1 = sc0 + 64, *sc2 = sc1 + 128, *sc3 = sc2 + 128;
    float *w0s = sc0 + JAC_SCALES;                      // [3][64] first layer's weights
    float *s4 = w0s + 192;                              // [16] scale4 of the slot's channel, 0 where z <= 0
    int *rowid = reinterpret_cast<int *>(s4 + ROWS);    // [16]
    unsigned *mw = reinterpret_cast<unsigned *>(rowid + ROWS);   // [16][MASK_WORDS]
    const int c0 = bx * ROWS, n = a.n, t = threadIdx.x;
    // every request of the prologue goes out before anything is waited for: the flag, the 16 channels' points and pool
    // values, the scales, W0 -- and the weights of the first product (layer 3)
    Frag16<256, 128> f3;
    Frag16<128, 128> f2;
    Frag16<128, 64> f1;
    frag16_load(f3, A.enc_bwd16[3]);
    const int dense = a.dense_flag[b];
    int my_row = 0;
    float my_s4 = 0.f;
    if (t < ROWS) {
        my_row = a.crit[(size_t)b * 128 + c0 + t];
        my_s4 = a.z[(size_t)b * 128 + c0 + t] > 0.f ? A.scale[4][c0 + t] : 0.f;
    }
    float v0 = 0.f, v1 = 0.f;
    if (t < 128) { v0 = A.scale[1][t]; v1 = A.scale[2][t]; }
    else if (t < 384) v0 = A.scale[3][t - 128];
    else if (t < 448) v0 = A.scale[0][t - 384];
    const float wv = t < 192 ? A.w0[t] : 0.f;
    if (dense != 0) return;                             // (uniform)
    if (t < ROWS) { rowid[t] = my_row; s4[t] = my_s4; }
    if (t < 128) { sc1[t] = v0; sc2[t] = v1; }
    else if (t < 384) sc3[t - 128] = v0;
    else if (t < 448) sc0[t - 384] = v0;
    if (t < 192) w0s[t] = wv;
    __syncthreads();
    for (int e = t; e < ROWS * MASK_WORDS; e += ENC_THREADS)
        mw[e] = a.masks[((size_t)b * n + rowid[e / MASK_WORDS]) * MASK_WORDS + e % MASK_WORDS];
    // dh4[s][:] = scale4 * W4^T[channel][:] -- a row of the packed 16x16x4 fragments (ae.h):
    //   W^T[k][nn] = packed16[((nn / 16 * K/16 + k / 16) * 64 + (k % 16) / 4 * 16 + nn % 16) * 4 + k % 4],  K = 128
    float g4[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {                       // 16 slots x 256 columns over 512 threads
        const int e = t + q * ENC_THREADS, s = e >> 8, nn = e & 255, k = c0 + s;
        g4[q] = A.enc_bwd16[4].w[(((nn >> 4) * 8 + (k >> 4)) * 64 + ((k & 15) >> 2) * 16 + (nn & 15)) * 4 + (k & 3)];
    }
    __syncthreads();                                    // masks, rowid, scales are in LDS
    auto bit = [&](int row, int off, int c) { return (mw[row * MASK_WORDS + off + (c >> 5)] >> (c & 31)) & 1u; };
    if (AHEAD) frag16_load(f2, A.enc_bwd16[2]);
#pragma unroll
    for (int q = 0; q < 8; ++q) {                       // da3 = dh4 * mask4 * scale3   into bufP
        const int e = t + q * ENC_THREADS, s = e >> 8, nn = e & 255;
        bufP[s * 260 + nn] = bit(s, MASK_OFF4, nn) ? (g4[q] * s4[s]) * sc3[nn] : 0.f;
    }
    __syncthreads();
    // dh3 = da3 @ W3^T (256 -> 128); da2 = dh3 * mask3 * scale2   into bufQ
    if (AHEAD) frag16_load(f1, A.enc_bwd16[1]);
    layer_gemm16(bufP, 260, f3, [&](int row, int c, float v) { bufQ[row * 132 + c] = bit(row, MASK_OFF3, c) ? v * sc2[c] : 0.f; });
    if (!AHEAD) frag16_load(f2, A.enc_bwd16[2]);
    __syncthreads();
    // dh2 = da2 @ W2^T (128 -> 128); da1 = dh2 * mask2 * scale1   into bufP (stride 132)
    layer_gemm16(bufQ, 132, f2, [&](int row, int c, float v) { bufP[row * 132 + c] = bit(row, MASK_OFF2, c) ? v * sc1[c] : 0.f; });
    if (!AHEAD) frag16_load(f1, A.enc_bwd16[1]);
    __syncthreads();
    // dh1 = da1 @ W1^T (128 -> 64); da0 = dh1 * mask1 * scale0   into bufQ (stride 68)
    layer_gemm16(bufP, 132, f1, [&](int row, int c, float v) { bufQ[row * 68 + c] = bit(row, 0, c) ? v * sc0[c] : 0.f; });
    __syncthreads();
    if (t < ROWS * 3) {   // dh0 = da0 @ W0^T (64 -> 3) on the VALU
        const int r = t / 3, x = t % 3;
        float s = 0.f;
#pragma unroll 8
        for (int c = 0; c < 64; ++c) s = fmaf(bufQ[r * 68 + c], w0s[x * 64 + c], s);
        a.jac[((size_t)b * 128 + c0 + r) * 3 + x] = s;
    }
}

// The block as extra workgroups of another kernel's launch (1-D grid, 512 threads): blocks [first_block, first_block + blocks).
// raise_prio: the rider's waves above the host kernel's.  Small batches: the Jacobian's 8 * b workgroups (16.4 us each beside the
// scan's waves, 12 alone) end the scan launch ~3 us after the scan itself (profiles/r05_timeline_b4.jsonl); at priority 1 they
// take what they need from the few CUs they share (B = 4: 0.0673 -> 0.0666 ms, B = 8: 0.0791 -> 0.0780).  At B = 32 there is a
// rider on EVERY CU and the scan is the long pole: raised priority costs 0.1706 -> 0.173.
struct JacRider { JacArgs j; DeviceAE A; int first_block, blocks; int raise_prio; };
// AHEAD: the form with the weights requested a layer ahead (145 registers: only a host kernel with that budget -- the screened scan)
template <bool AHEAD = false>
__device__ __forceinline__ bool jac_rider_block(const JacRider &r, float *lds) {
    if (r.blocks == 0 || (int)blockIdx.x < r.first_block || (int)blockIdx.x >= r.first_block + r.blocks) return false;
    const int g = blockIdx.x - r.first_block;
    if (r.raise_prio) __builtin_amdgcn_s_setprio(1);      // few workgroups beside a launch they would otherwise END (small batches)
    GA_STAMP(3, 0);
    encoder_jac_block<AHEAD>(r.A, r.j, lds, g % (128 / JAC_ROWS), g / (128 / JAC_ROWS));
    GA_STAMP(3, 7);
    return true;
}

}  // namespace geoadv
