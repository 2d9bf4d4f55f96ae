// geoadv_ae: device-resident, MFMA-packed weights of the victim auto-encoder + plain forward
// (AdversaryAutoEncoder.restore_ae_model / reconstruct, src/adversary_autoencoder.py:42-51,75-91).
#include "ae.h"
#include <cmath>
#include "encoder_x3.h"
#include <math.h>
#include <string.h>
#include <vector>
#include <atomic>

namespace geoadv {

int launch_encoder_fwd(const DeviceAE &A, int b, const float *x, const float *pert, float *adv_out, float *pmax,
                       int *parg, int *pcnt, unsigned *masks, hipStream_t stream, hipEvent_t start = nullptr, hipEvent_t stop = nullptr,
                       const FusedAdam *fused = nullptr);
int launch_latent_decode(const DeviceAE &A, int b, const float *pmax, const int *parg, const int *pcnt, float *z,
                         int *crit, int *zcnt, int *dense, float *d1, float *d2, hipStream_t stream);
int launch_decoder_fc2(const DeviceAE &A, int b, const float *d2, float *recon, hipStream_t stream, unsigned long long *fill = nullptr,
                       size_t fill_count = 0);
int launch_latent_fc(const DeviceAE &A, int b, const float *z, float *d2, hipStream_t stream);
int encoder_tiles_max(int n);

static inline size_t rup(size_t v, size_t a) { return (v + a - 1) / a * a; }

// B[k][n] for k < K, n < N taken from src through `at(k, n)`; zero padded to (Kp, Np).
template <class F>
static void pack_fragments(std::vector<float> &dst, size_t off, int K, int N, int Kp, int Np, F at) {
    const int kg = Kp / 8;
    for (int cb = 0; cb < Np / 32; ++cb)
        for (int t = 0; t < kg; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int u = 0; u < 4; ++u) {
                    const int k = 8 * t + 4 * (lane >> 5) + u, n = 32 * cb + (lane & 31);
                    dst[off + (((size_t)cb * kg + t) * 64 + lane) * 4 + u] = (k < K && n < N) ? at(k, n) : 0.f;
                }
}

// the 16x16x4 packing (ae.h)
template <class F>
static void pack_fragments16(std::vector<float> &dst, size_t off, int K, int N, int Kp, int Np, F at) {
    const int kg = Kp / 16;
    for (int cb = 0; cb < Np / 16; ++cb)
        for (int t = 0; t < kg; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int u = 0; u < 4; ++u) {
                    const int k = 16 * t + 4 * (lane >> 4) + u, n = 16 * cb + (lane & 15);
                    dst[off + (((size_t)cb * kg + t) * 64 + lane) * 4 + u] = (k < K && n < N) ? at(k, n) : 0.f;
                }
}

// bf16 pieces of an fp32 (encoder_x3.h): round to nearest even, the remainders exact
static inline uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);     // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_value(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline void x3_pieces(float x, uint16_t (&p)[3]) {
    p[0] = bf16_rne(x);
    if (std::isfinite(x) && !std::isfinite(bf16_value(p[0]))) {   // |x| within half a bf16 ulp of FLT_MAX: rounding up would be an
        uint32_t u;                                               // infinity and the remainders NaN -- truncate the first piece
        memcpy(&u, &x, 4);                                        // instead (the remainders stay exact, the three still carry x)
        p[0] = (uint16_t)(u >> 16);
    }
    const float r1 = x - bf16_value(p[0]);
    p[1] = bf16_rne(r1);
    const float r2 = r1 - bf16_value(p[1]);
    p[2] = bf16_rne(r2);
}
// The x3 weight image: for every sixteen-k step of the forward, 4 output-channel blocks x 3 pieces x 64 lanes x 8 k slots;
// lane (i, h) of block ob holds W_L[x3 input channel of slot j][32 ob + i] (the same values whichever operand they become).
static void pack_x3(uint32_t *img, const float *const W[ENC_L], const int *C) {
    for (int L = 1; L <= 4; ++L) {
        const int K = C[L], N = C[L + 1];
        for (int ob = 0; ob < N / 32; ++ob)
            for (int kb = 0; kb < K / 16; ++kb) {
                const int step = x3_step_of(L, ob, kb);
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, h = lane >> 5;
                    uint16_t pc[8][3];
                    for (int j = 0; j < 8; ++j) {
                        const int k = L == 4 ? 128 * (kb >> 3) + x3_in_channel(4, kb & 7, h, j) : x3_in_channel(L, kb, h, j);
                        x3_pieces(W[L][(size_t)k * N + 32 * ob + i], pc[j]);
                    }
                    for (int q = 0; q < 3; ++q) {
                        uint32_t *dst = img + ((size_t)(step * 4 + (ob & 3)) * 3 + q) * X3_FRAG_WORDS + lane * 4;
                        for (int w = 0; w < 4; ++w) dst[w] = (uint32_t)pc[2 * w][q] | ((uint32_t)pc[2 * w + 1][q] << 16);
                    }
                }
            }
    }
}

// fp16 pieces of an (already scaled) fp32 (encoder_x3.h, f16x2): round to nearest even, the remainder exact
static inline void h2_pieces(float x, uint16_t (&p)[2]) {
    const _Float16 h0 = (_Float16)x;
    const _Float16 h1 = (_Float16)(x - (float)h0);
    memcpy(&p[0], &h0, 2);
    memcpy(&p[1], &h1, 2);
}
// The power of two that brings a layer's largest |weight| into [2^13, 2^14); 0 if a weight is not finite.
static float h2_weight_scale(const float *W, size_t count) {
    float mx = 0.f;
    for (size_t i = 0; i < count; ++i) {
        if (!std::isfinite(W[i])) return 0.f;
        mx = std::max(mx, std::fabs(W[i]));
    }
    if (mx == 0.f) return 1.f;
    int e;
    (void)frexpf(mx, &e);                       // mx in [2^(e-1), 2^e)
    return ldexpf(1.f, 14 - e);
}
// The f16x2 weight image: pack_x3's layout with two fp16 pieces of S_w(L) W_L per fragment position.
static void pack_h2(uint32_t *img, const float *const W[ENC_L], const int *C, const float *sw) {
    for (int L = 1; L <= 4; ++L) {
        const int K = C[L], N = C[L + 1];
        for (int ob = 0; ob < N / 32; ++ob)
            for (int kb = 0; kb < K / 16; ++kb) {
                const int step = x3_step_of(L, ob, kb);
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, h = lane >> 5;
                    uint16_t pc[8][2];
                    for (int j = 0; j < 8; ++j) {
                        const int k = L == 4 ? 128 * (kb >> 3) + x3_in_channel(4, kb & 7, h, j) : x3_in_channel(L, kb, h, j);
                        h2_pieces(W[L][(size_t)k * N + 32 * ob + i] * sw[L], pc[j]);
                    }
                    for (int q = 0; q < 2; ++q) {
                        uint32_t *dst = img + ((size_t)(step * 4 + (ob & 3)) * 2 + q) * X3_FRAG_WORDS + lane * 4;
                        for (int w = 0; w < 4; ++w) dst[w] = (uint32_t)pc[2 * w][q] | ((uint32_t)pc[2 * w + 1][q] << 16);
                    }
                }
            }
    }
}

struct ForwardScratch {
    float *pmax; int *parg; int *pcnt;   // [b][tiles][128]
    float *z; int *crit; int *zcnt;      // [b][128]
    int *dense;                          // [b]
    float *d1, *d2;                      // [b][256]
    size_t bytes;
};

ForwardScratch carve_forward_scratch(void *base, int b, int n_points) {
    ForwardScratch s;
    const size_t tiles = encoder_tiles_max(n_points);     // (the forward picks its tile height per launch: encoder.hip)
    char *p = static_cast<char *>(base);
    auto take = [&](size_t bytes) { char *q = p; p += rup(bytes, 256); return q; };
    s.pmax = reinterpret_cast<float *>(take(sizeof(float) * b * tiles * 128));
    s.parg = reinterpret_cast<int *>(take(sizeof(int) * b * tiles * 128));
    s.pcnt = reinterpret_cast<int *>(take(sizeof(int) * b * tiles * 128));
    s.z = reinterpret_cast<float *>(take(sizeof(float) * b * 128));
    s.crit = reinterpret_cast<int *>(take(sizeof(int) * b * 128));
    s.zcnt = reinterpret_cast<int *>(take(sizeof(int) * b * 128));
    s.dense = reinterpret_cast<int *>(take(sizeof(int) * b));
    s.d1 = reinterpret_cast<float *>(take(sizeof(float) * b * 256));
    s.d2 = reinterpret_cast<float *>(take(sizeof(float) * b * 256));
    s.bytes = (size_t)(p - static_cast<char *>(base));
    return s;
}

int run_forward(const DeviceAE &A, int b, const float *x, const float *pert, float *adv_out, const ForwardScratch &s,
                float *recon, hipStream_t stream) {
    if (int st = launch_encoder_fwd(A, b, x, pert, adv_out, s.pmax, s.parg, s.pcnt, nullptr, stream)) return st;
    if (int st = launch_latent_decode(A, b, s.pmax, s.parg, s.pcnt, s.z, s.crit, s.zcnt, s.dense,
                                      recon ? s.d1 : nullptr, s.d2, stream)) return st;
    if (recon)
        if (int st = launch_decoder_fc2(A, b, s.d2, recon, stream)) return st;
    return GEOADV_OK;
}

}  // namespace geoadv

using namespace geoadv;

// -1: f16x2 where the model's constants scale exactly (every sane model), bf16x3 otherwise
static std::atomic<int> g_default_enc_arith{-1};
static bool known_arith(int a) { return a == GEOADV_ENC_ARITH_F32 || a == GEOADV_ENC_ARITH_BF16X3 || a == GEOADV_ENC_ARITH_F16X2; }

extern "C" int geoadv_ae_create(geoadv_ae **out, const geoadv_ae_weights *hw) {
    GA_REQUIRE(out && hw, "ae_create: null argument");
    static const int want_enc[ENC_L + 1] = {3, 64, 128, 128, 256, 128};
    for (int i = 0; i <= ENC_L; ++i)
        GA_REQUIRE(hw->enc_dims[i] == want_enc[i],
                   "ae_create: encoder widths must be 3,64,128,128,256,128 (src/ae_templates.py:22); got %d at %d",
                   hw->enc_dims[i], i);
    const int n = hw->n_points;
    GA_REQUIRE(n >= 1 && n <= 32768, "ae_create: n_points %d out of range [1, 32768]", n);
    GA_REQUIRE(hw->dec_dims[0] == 128 && hw->dec_dims[1] == 256 && hw->dec_dims[2] == 256 && hw->dec_dims[3] == 3 * n,
               "ae_create: decoder widths must be 128,256,256,3*n_points (src/ae_templates.py:29)");
    for (int i = 0; i < ENC_L; ++i)
        GA_REQUIRE(hw->enc_w[i] && hw->enc_b[i] && hw->bn_gamma[i] && hw->bn_beta[i] && hw->bn_mean[i] && hw->bn_var[i],
                   "ae_create: null encoder weight pointer at layer %d", i);
    for (int k = 0; k < GEOADV_DEC_LAYERS; ++k)
        GA_REQUIRE(hw->dec_w[k] && hw->dec_b[k], "ae_create: null decoder weight pointer at layer %d", k);

    const int *C = hw->enc_dims;
    const int n3 = 3 * n, n3p32 = (int)rup(n3, 32), n3p8 = (int)rup(n3, 8);
    // arena layout (floats)
    std::vector<float> host;
    auto reserve = [&](size_t count) { size_t off = rup(host.size(), 64); host.resize(off + count, 0.f); return off; };
    size_t o_w0 = reserve(3 * C[1]);
    size_t o_fwd[ENC_L] = {0}, o_bwd[ENC_L] = {0}, o_bwd16[ENC_L] = {0}, o_scale[ENC_L], o_shift[ENC_L];
    for (int i = 1; i < ENC_L; ++i) {
        o_fwd[i] = reserve((size_t)C[i] * C[i + 1]);
        o_bwd[i] = reserve((size_t)C[i] * C[i + 1]);
        o_bwd16[i] = reserve((size_t)C[i] * C[i + 1]);
    }
    for (int i = 0; i < ENC_L; ++i) { o_scale[i] = reserve(C[i + 1]); o_shift[i] = reserve(C[i + 1]); }
    size_t o_v0 = reserve(128 * 256), o_c0 = reserve(256), o_v1 = reserve(256 * 256), o_c1 = reserve(256);
    size_t o_v0t = reserve(256 * 128), o_v1t = reserve(256 * 256);
    size_t o_d2f = reserve((size_t)256 * n3p32), o_d2b = reserve((size_t)n3p8 * 256), o_c2 = reserve(n3);
    size_t o_x3 = reserve(X3_IMAGE_WORDS), o_x3c = reserve(X3_CONST_FLOATS);
    size_t o_h2 = reserve(H2_IMAGE_WORDS), o_h2c = reserve(X3_CONST_FLOATS), o_flag = reserve(1);

    memcpy(&host[o_w0], hw->enc_w[0], sizeof(float) * 3 * C[1]);
    for (int i = 1; i < ENC_L; ++i) {
        const float *W = hw->enc_w[i];
        const int K = C[i], N = C[i + 1];
        pack_fragments(host, o_fwd[i], K, N, K, N, [&](int k, int nn) { return W[(size_t)k * N + nn]; });
        // transposed product: B[k][nn] = W[nn][k], K' = N, N' = K
        pack_fragments(host, o_bwd[i], N, K, N, (int)rup(K, 32), [&](int k, int nn) { return W[(size_t)nn * N + k]; });
        pack_fragments16(host, o_bwd16[i], N, K, N, K, [&](int k, int nn) { return W[(size_t)nn * N + k]; });   // (widths are multiples of 64)
    }
    for (int i = 0; i < ENC_L; ++i)
        for (int c = 0; c < C[i + 1]; ++c) {
            // tf.nn.batch_normalization, inference branch: inv = gamma * rsqrt(var + eps);
            // y = a * inv + (beta - mean * inv) with a = x@W + b  ==>  y = (x@W) * inv + (b*inv + beta - mean*inv)
            const float inv = hw->bn_gamma[i][c] * (1.0f / sqrtf(hw->bn_var[i][c] + 1e-5f));
            host[o_scale[i] + c] = inv;
            host[o_shift[i] + c] = hw->enc_b[i][c] * inv + (hw->bn_beta[i][c] - hw->bn_mean[i][c] * inv);
        }
    {   // the x3 forward's constant block (encoder_x3.h)
        float *c = &host[o_x3c];
        memcpy(c, hw->enc_w[0], sizeof(float) * 192);
        memcpy(c + 192, &host[o_scale[0]], sizeof(float) * 64);
        memcpy(c + 256, &host[o_shift[0]], sizeof(float) * 64);
        const int off[5] = {0, X3_SC1, X3_SC2, X3_SC3, X3_SC4};
        for (int i = 1; i < ENC_L; ++i) {
            memcpy(c + off[i], &host[o_scale[i]], sizeof(float) * C[i + 1]);
            memcpy(c + off[i] + C[i + 1], &host[o_shift[i]], sizeof(float) * C[i + 1]);
        }
    }
    // the f16x2 forward's image and constants (encoder_x3.h): weights scaled by S_w(L), the activations a layer hands on by s_j
    // (below), all of it folded into the epilogue's constants.  Every folded constant must be the exact power-of-two multiple
    // (no under- / overflow): else no f16x2 for this model.
    float sw[ENC_L] = {1.f, 1.f, 1.f, 1.f, 1.f}, h2_unscale[ENC_L] = {1.f, 1.f, 1.f, 1.f, 1.f}, act_scale[ENC_L] = {1.f, 1.f, 1.f, 1.f, 1.f};
    bool h2_ok = true;
    {
        // s_j, the power of two layer j's activations are carried times (j = 0..3): 2^6 for a layer whose batch norm has gamma^2 +
        // beta^2 = 1 on average -- relu(gamma z + beta) with z ~ N(0, 1) is what a layer trained on such data puts out --, moved with
        // that magnitude otherwise, so that the fp16 window [2^-9, 1023.5] / 2^6 x t_j sits where the model's own constants say its
        // activations are (a layer with gamma = 2^-12 would otherwise work on second pieces that are all fp16 subnormals:
        // tools/debug/h2_low_side.py).  gamma = 1, beta = 0 gives exactly 2^6.
        for (int j = 0; j < ENC_L - 1; ++j) {
            double q = 0.0;
            for (int c = 0; c < C[j + 1]; ++c) q += (double)hw->bn_gamma[j][c] * hw->bn_gamma[j][c] + (double)hw->bn_beta[j][c] * hw->bn_beta[j][c];
            const double t = std::sqrt(q / C[j + 1]);
            const int e = (std::isfinite(t) && t > 0.0) ? std::max(-60, std::min(60, (int)std::lround(std::log2(t)))) : 0;   // t ~ 2^e
            act_scale[j] = ldexpf(1.f, 6 - e);
        }
        for (int L = 1; L < ENC_L; ++L) {
            sw[L] = h2_weight_scale(hw->enc_w[L], (size_t)C[L] * C[L + 1]);
            if (sw[L] == 0.f) { h2_ok = false; sw[L] = 1.f; }
            h2_unscale[L] = 1.f / (act_scale[L - 1] * sw[L]);
        }
        float *c = &host[o_h2c];
        const float *x3c = &host[o_x3c];
        auto fold = [&](float v, float factor) {            // v * factor, factor a power of two: exact unless it leaves the normal range
            const float r = v * factor;
            if (!std::isfinite(r) || r / factor != v) h2_ok = false;
            return r;
        };
        memcpy(c, x3c, sizeof(float) * 192);
        for (int k = 0; k < 64; ++k) { c[192 + k] = fold(x3c[192 + k], act_scale[0]); c[256 + k] = fold(x3c[256 + k], act_scale[0]); }
        const int off[5] = {0, X3_SC1, X3_SC2, X3_SC3, X3_SC4};
        for (int L = 1; L < ENC_L; ++L)
            for (int k = 0; k < C[L + 1]; ++k) {
                // layers 1-3 hand a scaled activation on (s_L a); layer 4's result is the (unscaled) latent candidate
                c[off[L] + k] = fold(fold(x3c[off[L] + k], h2_unscale[L]), L < 4 ? act_scale[L] : 1.f);
                c[off[L] + C[L + 1] + k] = L < 4 ? fold(x3c[off[L] + C[L + 1] + k], act_scale[L]) : x3c[off[L] + C[L + 1] + k];
            }
        std::vector<uint32_t> img(H2_IMAGE_WORDS, 0u);
        if (h2_ok) pack_h2(img.data(), hw->enc_w, C, sw);
        memcpy(&host[o_h2], img.data(), sizeof(uint32_t) * H2_IMAGE_WORDS);
    }
    memcpy(&host[o_v0], hw->dec_w[0], sizeof(float) * 128 * 256);
    memcpy(&host[o_c0], hw->dec_b[0], sizeof(float) * 256);
    memcpy(&host[o_v1], hw->dec_w[1], sizeof(float) * 256 * 256);
    memcpy(&host[o_c1], hw->dec_b[1], sizeof(float) * 256);
    for (int k = 0; k < 128; ++k)
        for (int t = 0; t < 256; ++t) host[o_v0t + (size_t)t * 128 + k] = hw->dec_w[0][(size_t)k * 256 + t];
    for (int k = 0; k < 256; ++k)
        for (int t = 0; t < 256; ++t) host[o_v1t + (size_t)t * 256 + k] = hw->dec_w[1][(size_t)k * 256 + t];
    {
        const float *V2 = hw->dec_w[2];
        pack_fragments(host, o_d2f, 256, n3, 256, n3p32, [&](int k, int nn) { return V2[(size_t)k * n3 + nn]; });
        pack_fragments(host, o_d2b, n3, 256, n3p8, 256, [&](int k, int nn) { return V2[(size_t)nn * n3 + k]; });
    }
    memcpy(&host[o_c2], hw->dec_b[2], sizeof(float) * n3);
    {
        std::vector<uint32_t> img(X3_IMAGE_WORDS, 0u);
        pack_x3(img.data(), hw->enc_w, C);
        memcpy(&host[o_x3], img.data(), sizeof(uint32_t) * X3_IMAGE_WORDS);      // (bit patterns: the arena is only a byte container here)
    }

    geoadv_ae *ae = new geoadv_ae();
    ae->arena_bytes = sizeof(float) * host.size();
    if (hipMalloc(&ae->arena, ae->arena_bytes) != hipSuccess) {
        delete ae;
        set_error("ae_create: hipMalloc of %zu bytes failed", sizeof(float) * host.size());
        return GEOADV_ENOMEM;
    }
    hipError_t e = hipMemcpy(ae->arena, host.data(), ae->arena_bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(ae->arena);
        delete ae;
        set_error("ae_create: upload failed: %s", hipGetErrorString(e));
        return GEOADV_EHIP;
    }
    const float *base = static_cast<const float *>(ae->arena);
    DeviceAE &d = ae->d;
    d.n_points = n;
    d.bneck = 128;
    memcpy(d.enc_dims, hw->enc_dims, sizeof(d.enc_dims));
    memcpy(d.dec_dims, hw->dec_dims, sizeof(d.dec_dims));
    d.w0 = base + o_w0;
    for (int i = 0; i < ENC_L; ++i) {
        d.enc_fwd[i] = PackedLayer{i ? base + o_fwd[i] : nullptr, C[i], C[i + 1]};
        d.enc_bwd[i] = PackedLayer{i ? base + o_bwd[i] : nullptr, C[i + 1], (int)rup(C[i], 32)};
        d.enc_bwd16[i] = PackedLayer{i ? base + o_bwd16[i] : nullptr, C[i + 1], C[i]};
        d.scale[i] = base + o_scale[i];
        d.shift[i] = base + o_shift[i];
    }
    d.v0 = base + o_v0; d.c0 = base + o_c0; d.v1 = base + o_v1; d.c1 = base + o_c1;
    d.v0t = base + o_v0t; d.v1t = base + o_v1t;
    d.dec2_fwd = PackedLayer{base + o_d2f, 256, n3p32};
    d.dec2_bwd = PackedLayer{base + o_d2b, n3p8, 256};
    d.c2 = base + o_c2;
    d.enc_x3 = reinterpret_cast<const unsigned *>(base + o_x3);
    d.enc_x3_consts = base + o_x3c;
    d.enc_h2 = h2_ok ? reinterpret_cast<const unsigned *>(base + o_h2) : nullptr;
    d.enc_h2_consts = h2_ok ? base + o_h2c : nullptr;
    memcpy(d.h2_unscale, h2_unscale, sizeof(h2_unscale));
    memcpy(d.h2_act_scale, act_scale, sizeof(act_scale));
    d.range_flag = reinterpret_cast<int *>(static_cast<float *>(ae->arena) + o_flag);     // (uploaded as 0)
    {
        const int want = g_default_enc_arith.load();
        d.enc_arith = want < 0 ? (h2_ok ? GEOADV_ENC_ARITH_F16X2 : GEOADV_ENC_ARITH_BF16X3)
                               : (want == GEOADV_ENC_ARITH_F16X2 && !h2_ok ? GEOADV_ENC_ARITH_BF16X3 : want);
    }
    *out = ae;
    return GEOADV_OK;
}

extern "C" int geoadv_ae_set_encoder_arith(geoadv_ae *ae, int arith) {
    GA_REQUIRE(ae, "ae_set_encoder_arith: null handle");
    GA_REQUIRE(known_arith(arith), "ae_set_encoder_arith: unknown arithmetic %d", arith);
    GA_REQUIRE(arith != GEOADV_ENC_ARITH_F16X2 || ae->d.enc_h2,
               "ae_set_encoder_arith: this model's weights / batch-norm constants do not scale into the fp16 range exactly "
               "(non-finite, or a constant that would leave the normal fp32 range): f16x2 is not available for it");
    // an attack handle caches a forward (pool partials, masks) in the tile layout of the arithmetic it ran under and recomputes in
    // it: switching under a live handle would make the two disagree -- refused, not silently accepted
    GA_REQUIRE(arith == ae->d.enc_arith || ae->attack_refs.load() == 0,
               "ae_set_encoder_arith: %d attack handle(s) hold a forward of this model in its present arithmetic; destroy them first "
               "(or choose the arithmetic when the model is created)", ae->attack_refs.load());
    ae->d.enc_arith = arith;
    return GEOADV_OK;
}
extern "C" int geoadv_ae_encoder_arith(const geoadv_ae *ae) { return ae ? ae->d.enc_arith : -1; }

// The f16x2 range guard's verdict (encoder_x3.h): synchronises, reads and clears the model's flag.
int geoadv::ae_range_check(const geoadv_ae *ae, hipStream_t stream, const char *who) {
    GA_HIP(hipStreamSynchronize(stream));
    int flag = 0;
    GA_HIP(hipMemcpy(&flag, ae->d.range_flag, sizeof(int), hipMemcpyDeviceToHost));
    if (!flag) return GEOADV_OK;
    GA_HIP(hipMemset(ae->d.range_flag, 0, sizeof(int)));
    set_error("%s: an encoder forward in the f16x2 arithmetic met an activation outside the fp16 range of its scaled operands (1023.5 x "
              "the magnitude its layer's batch-norm constants announce; layer scales 2^%d 2^%d 2^%d 2^%d): the latents of the clouds "
              "concerned were set to +inf.  Use GEOADV_ENC_ARITH_BF16X3 for this model", who, ilogbf(ae->d.h2_act_scale[0]),
              ilogbf(ae->d.h2_act_scale[1]), ilogbf(ae->d.h2_act_scale[2]), ilogbf(ae->d.h2_act_scale[3]));
    return GEOADV_ERANGE;
}
extern "C" int geoadv_ae_status(const geoadv_ae *ae, void *stream) {
    GA_REQUIRE(ae, "ae_status: null handle");
    return ae_range_check(ae, static_cast<hipStream_t>(stream), "ae_status");
}
extern "C" int geoadv_set_default_encoder_arith(int arith) {
    GA_REQUIRE(arith == GEOADV_ENC_ARITH_AUTO || known_arith(arith), "set_default_encoder_arith: unknown arithmetic %d", arith);
    g_default_enc_arith.store(arith);
    return GEOADV_OK;
}

extern "C" void geoadv_ae_destroy(geoadv_ae *ae) {
    if (!ae) return;
    (void)hipFree(ae->arena);
    delete ae;
}

extern "C" size_t geoadv_ae_workspace_bytes(const geoadv_ae *ae, int b) {
    if (!ae || b <= 0) return 256;
    return carve_forward_scratch(nullptr, b, ae->d.n_points).bytes + 256;
}

extern "C" int geoadv_ae_forward(const geoadv_ae *ae, int b, const float *pc, float *latent, float *recon,
                                 void *workspace, void *stream) {
    GA_REQUIRE(ae && b >= 0, "ae_forward: bad arguments");
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(pc && workspace, "ae_forward: null pointer");
    GA_REQUIRE(b <= 65535, "ae_forward: batch %d exceeds 65535", b);
    hipStream_t st = as_stream(stream);
    void *aligned = reinterpret_cast<void *>(rup(reinterpret_cast<size_t>(workspace), 256));
    ForwardScratch s = carve_forward_scratch(aligned, b, ae->d.n_points);
    if (int rc = run_forward(ae->d, b, pc, nullptr, nullptr, s, recon, st)) return rc;
    if (latent) GA_HIP(hipMemcpyAsync(latent, s.z, sizeof(float) * (size_t)b * 128, hipMemcpyDeviceToDevice, st));
    return GEOADV_OK;
}

extern "C" int geoadv_ae_critical(const geoadv_ae *ae, int b, const float *pc, float *latent, int *arg_idx,
                                  void *workspace, void *stream) {
    GA_REQUIRE(ae && b >= 0, "ae_critical: bad arguments");
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(pc && workspace, "ae_critical: null pointer");
    GA_REQUIRE(b <= 65535, "ae_critical: batch %d exceeds 65535", b);
    hipStream_t st = as_stream(stream);
    void *aligned = reinterpret_cast<void *>(rup(reinterpret_cast<size_t>(workspace), 256));
    ForwardScratch s = carve_forward_scratch(aligned, b, ae->d.n_points);
    if (int rc = run_forward(ae->d, b, pc, nullptr, nullptr, s, nullptr, st)) return rc;
    if (latent) GA_HIP(hipMemcpyAsync(latent, s.z, sizeof(float) * (size_t)b * 128, hipMemcpyDeviceToDevice, st));
    if (arg_idx) GA_HIP(hipMemcpyAsync(arg_idx, s.crit, sizeof(int) * (size_t)b * 128, hipMemcpyDeviceToDevice, st));
    return GEOADV_OK;
}

extern "C" int geoadv_ae_decode(const geoadv_ae *ae, int b, const float *latent, float *recon, void *workspace, void *stream) {
    GA_REQUIRE(ae && b >= 0, "ae_decode: bad arguments");
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(latent && recon && workspace, "ae_decode: null pointer");
    GA_REQUIRE(b <= 65535, "ae_decode: batch %d exceeds 65535", b);
    hipStream_t st = as_stream(stream);
    void *aligned = reinterpret_cast<void *>(rup(reinterpret_cast<size_t>(workspace), 256));
    ForwardScratch s = carve_forward_scratch(aligned, b, ae->d.n_points);
    if (int rc = launch_latent_fc(ae->d, b, latent, s.d2, st)) return rc;
    return launch_decoder_fc2(ae->d, b, s.d2, recon, st);
}
