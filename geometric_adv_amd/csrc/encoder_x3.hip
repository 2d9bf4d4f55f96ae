// Encoder forward in the x3 arithmetic (encoder_x3.h): a WAVE owns 32 points and carries them through all five layers.
//
// Reference semantics: src/encoders_decoders.py:37-72 with the widths of src/ae_templates.py:22 -- what encoder.hip's fp32
// kernel computes, same outputs (per-tile pool maximum / first arg-max / tie count, ReLU masks, adv = x + pert with the fused
// Adam step), products formed as six bf16 piece products instead of one fp32 product.
//
// Shape.  On the 32x32x16 MFMA with the WEIGHTS as the A operand, a lane holds, for ITS point, four runs of four consecutive
// output channels per accumulator -- exactly the eight k slots (two runs) that lane feeds to the next layer's MFMA as the B
// operand.  So activations never leave the registers between layers: no LDS tile, no workgroup barrier at a layer boundary,
// no bank conflicts, and the BN + ReLU + split epilogue of one channel block runs under the next layer's first MFMAs.  Layer 4
// takes the activations as the A operand instead (same registers), so its result has the channel on the lane and the points in
// the registers: the max-pool is reduced in registers like the fp32 kernel's.
// Two waves per SIMD, each a 256-register wave (h3 = 96 registers of pieces, two accumulator sets of 64, the epilogue's staging);
// the four waves of a workgroup share the only operand that streams, the weights: 12 KiB per sixteen-k step, LDS-DMA'd into a
// four-slot ring three steps ahead (each wave fetches the fragments of "its" channel block; one raw s_barrier per step orders
// them) and read back as 12 ds_read_b128 per 24 MFMAs; two workgroups share a CU, one's stalls under the other's MFMAs.
#include "encoder_x3.h"
#include "encoder_jac.h"
#include <hip/hip_ext.h>
#include <limits.h>
#include <type_traits>

namespace geoadv {

constexpr int X3_THREADS = 256;               // 4 waves, one per SIMD
constexpr int X3_POINTS = 128;                // points per workgroup
constexpr int X3_RING = 4, X3_AHEAD = 3;      // ring slots and slots in flight ahead of the consumer; a slot = X3_SLOT_STEPS steps
// TWO workgroups per CU (two waves per SIMD): one's prologue, barriers and the epilogue instructions that found no shadow run
// under the other's MFMAs -- 0.1432 -> 0.135 ms per iteration at B = 32.  What it takes: 256 registers per wave (the compiler gets
// there from 288 with 4 spilled) and 80 KB of LDS per workgroup, i.e. ring slots of ONE step (with two steps per slot and one
// workgroup per CU: 0.1432; one step per slot, one workgroup: 0.142 -- the barrier per step costs nothing since the ring's
// LDS-DMA comes from inline assembly).
constexpr int X3_SLOT_STEPS = 1;              // steps per ring slot = per barrier
constexpr int X3_BIG_WG_PER_CU = 2;
constexpr int X3_SLOTS = X3_STEPS / X3_SLOT_STEPS;
__host__ __device__ constexpr int xp_slot_words(int np) { return X3_SLOT_STEPS * xp_step_words(np); }
static_assert(X3_STEPS % X3_SLOT_STEPS == 0, "whole slots");
#ifndef X3_SPLIT_HALVES
#define X3_SPLIT_HALVES 1                     // the split form while 128-point workgroups would cover at most this many halves of the CUs
#endif                                        // (measured, iteration ms at B = 8 / 16: split 0.0766 / 0.108, wave-private 0.0825 / 0.0935)
constexpr size_t x3_lds_bytes(int np, bool masks) {
    return (size_t)X3_RING * xp_slot_words(np) * 4 + sizeof(float) * X3_CONST_FLOATS + (sizeof(float) + 2 * sizeof(int)) * 4 * 128 +
           (masks ? sizeof(unsigned) * 2 * X3_POINTS * MASK_WORDS : 0);
}
static_assert(X3_BIG_WG_PER_CU * x3_lds_bytes(3, true) <= 160 * 1024, "the workgroups of a CU share its 160 KB of LDS");

// LDS-DMA of 16 bytes per lane: lane l's bytes land at lds_dst + 16 l (lds_dst wave-uniform).  Inline assembly, not
// __builtin_amdgcn_global_load_lds: with the builtin the compiler knows the LDS is being written and puts an s_waitcnt vmcnt(0)
// in front of the next ds_read that might alias -- every LDS read of this kernel -- which drains the ring at every slot; here the
// counted vmcnt + barrier of slot_sync is the only ordering, as intended.  (M0 is compiler-reserved: saved and restored in the
// same statement.)
__device__ __forceinline__ void x3_glds16(const unsigned *gsrc, unsigned *lds_dst) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(lds_dst));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// (v > 0) of a non-negative-or-minus-zero float shifted into `m` from the right: bits + 0x7fffffff carries into bit 31 exactly
// for the bit patterns 1 .. 0x7fffffff (two instructions per value; -0 and +0 give 0)
__device__ __forceinline__ unsigned x3_push_positive(unsigned m, float v) {
    return __builtin_amdgcn_alignbit(m, __float_as_uint(v) + 0x7fffffffu, 31);
}

// The lane's point (both lanes of a point load it) with the pending Adam step applied on the way (attack.hip adam_kernel, the same
// operations in the same order).  The stores are a step of their own (x3_store_point): where several waves load the same
// points (the split form) they must all have loaded before one of them stores.
struct X3Point { float v[3], g[3], m[3], vv[3], pnew[3]; size_t pg; bool valid; };
__device__ __forceinline__ void x3_load_point(int n, int b, int pt_raw, const float *x, const float *pert, const FusedAdam &fa, X3Point &o) {
    o.valid = pt_raw < n;
    const int pt = o.valid ? pt_raw : n - 1;
    o.pg = ((size_t)b * n + pt) * 3;
    float xv[3], pp[3] = {0.f, 0.f, 0.f}, ag[3], agd[3], am[3], av[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) xv[a] = x[o.pg + a];
    if (pert) {
#pragma unroll
        for (int a = 0; a < 3; ++a) pp[a] = pert[o.pg + a];
    }
    if (fa.m) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { ag[a] = fa.g_enc[o.pg + a]; agd[a] = fa.g_dist[o.pg + a]; am[a] = fa.m[o.pg + a]; av[a] = fa.v[o.pg + a]; }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float v = pert ? xv[a] + pp[a] : xv[a];
        if (fa.m) {
            float g = ag[a];
            g += agd[a];
            float m = am[a], vv = av[a];
            m += (g - m) * fa.one_minus_b1;
            vv += (g * g - vv) * fa.one_minus_b2;
            const float pnew = pp[a] - (m * fa.alpha) / (sqrtf(vv) + fa.eps);
            v = xv[a] + pnew;
            o.g[a] = g; o.m[a] = m; o.vv[a] = vv; o.pnew[a] = pnew;
        }
        o.v[a] = v;
    }
}
__device__ __forceinline__ void x3_store_point(const X3Point &o, float *adv_out, const FusedAdam &fa) {
    if (!o.valid) return;                              // (padding lanes repeat the cloud's last point and must not store)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (fa.m) {
            fa.g_enc[o.pg + a] = 0.f;
            if (fa.grad_out) fa.grad_out[o.pg + a] = o.g[a];
            fa.m[o.pg + a] = o.m[a]; fa.v[o.pg + a] = o.vv[a]; fa.pert[o.pg + a] = o.pnew[a];
        }
        if (adv_out) adv_out[o.pg + a] = o.v[a];
    }
}

// layer 0's constants and the (scale, shift) pairs of layers 1-4 -> LDS: a straight copy of the block ae.hip packs in this
// order (X3_CONST_FLOATS floats; two 16-byte loads per thread, all in flight at once -- gathering them from the seven arrays
// cost seven dependent round trips, 1.5 us of every workgroup's prologue)
__device__ __forceinline__ void x3_stage_constants(const float *consts, float *cst, int threads) {
    const float4 *src = reinterpret_cast<const float4 *>(consts);
    float4 *dst = reinterpret_cast<float4 *>(cst);
    for (int e = threadIdx.x; e < X3_CONST_FLOATS / 4; e += threads) dst[e] = src[e];
}

// Layer 0 (fan-in 3) on the VALU: fwd_layer0's arithmetic (encoder.hip), 32 channels per lane = the k slots it feeds to layer 1.
// m01: this lane's bits of mask words 0 and 1 (channel 16 kb + 8 h + j = bit (kb & 1) * 16 + 8 h + j of word kb / 2).
// (f16x2: scale0 / shift0 arrive multiplied by s_0, v is the scaled activation; gmax: the range guard's running maximum)
template <int NP, bool MASKS>
__device__ __forceinline__ void x3_layer0(const float *cst, const float (&pc)[3], int h, XP<NP> (&act1)[4], unsigned (&m01)[2], float &gmax) {
    const float4 *c4 = reinterpret_cast<const float4 *>(cst);
    m01[0] = m01[1] = 0u;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        float v[8];
        unsigned m8 = 0;
        const int c0 = 16 * kb + 8 * h;
        float wx[8], wy[8], wz[8], sc[8], sh[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 a = c4[(c0 >> 2) + q], bb = c4[((64 + c0) >> 2) + q], c = c4[((128 + c0) >> 2) + q];
            const float4 d = c4[((192 + c0) >> 2) + q], e = c4[((256 + c0) >> 2) + q];
            wx[4 * q] = a.x; wx[4 * q + 1] = a.y; wx[4 * q + 2] = a.z; wx[4 * q + 3] = a.w;
            wy[4 * q] = bb.x; wy[4 * q + 1] = bb.y; wy[4 * q + 2] = bb.z; wy[4 * q + 3] = bb.w;
            wz[4 * q] = c.x; wz[4 * q + 1] = c.y; wz[4 * q + 2] = c.z; wz[4 * q + 3] = c.w;
            sc[4 * q] = d.x; sc[4 * q + 1] = d.y; sc[4 * q + 2] = d.z; sc[4 * q + 3] = d.w;
            sh[4 * q] = e.x; sh[4 * q + 1] = e.y; sh[4 * q + 2] = e.z; sh[4 * q + 3] = e.w;
        }
#pragma unroll
        for (int j = 7; j >= 0; --j) {
            float a = pc[0] * wx[j];
            a = fmaf(pc[1], wy[j], a);
            a = fmaf(pc[2], wz[j], a);
            v[j] = fmaxf(fmaf(a, sc[j], sh[j]), 0.f);
            if (MASKS) m8 = x3_push_positive(m8, v[j]);          // j descending: bit j = [v[j] > 0]
        }
        xp_split8(v, act1[kb], gmax);
        if (MASKS) m01[kb >> 1] |= m8 << ((kb & 1) * 16 + 8 * h);
    }
}

// BN + ReLU + split of HALF a channel block of a layer result (acc: lane = point, registers = channels 32 cb + 8 g + 4 h + u;
// gh = 0: g = 0, 1; gh = 1: g = 2, 3): the pieces of the next layer's sixteen-k block 2 cb + gh; m16 collects the mask bits.
// sc, sh: the layer's scale / shift at channel 32 cb + 4 h (LDS).
template <int NP, bool MASKS>
__device__ __forceinline__ void x3_epilogue_half(const f32x16 &acc, const float *sc_p, const float *sh_p, int gh, XP<NP> &dst, unsigned &m16, float &gmax) {
    const float4 *s4 = reinterpret_cast<const float4 *>(sc_p), *t4 = reinterpret_cast<const float4 *>(sh_p);
#pragma unroll
    for (int gg = 1; gg >= 0; --gg) {
        const int g = 2 * gh + gg;
        const float4 sc = s4[2 * g], sh = t4[2 * g];
        float v[4];
        v[3] = fmaxf(fmaf(acc[4 * g + 3], sc.w, sh.w), 0.f);
        v[2] = fmaxf(fmaf(acc[4 * g + 2], sc.z, sh.z), 0.f);
        v[1] = fmaxf(fmaf(acc[4 * g + 1], sc.y, sh.y), 0.f);
        v[0] = fmaxf(fmaf(acc[4 * g + 0], sc.x, sh.x), 0.f);
        if (MASKS) {
#pragma unroll
            for (int u = 3; u >= 0; --u) m16 = x3_push_positive(m16, v[u]);   // within a half: bit 4 gg + u
        }
#pragma unroll
        for (int w = 0; w < 2; ++w) xp_split_pair(v[2 * w], v[2 * w + 1], dst, 2 * gg + w, gmax);
    }
}
// this lane's bits of a channel block's mask word from the bits of the two halves (bit 4 gg + u each): channel 8 g + 4 h + u
__device__ __forceinline__ unsigned x3_mask_bits(unsigned m_lo, unsigned m_hi, int h) {
    const unsigned m16 = (m_lo & 0xffu) | ((m_hi & 0xffu) << 8);
    const unsigned spread = (m16 & 0xfu) | ((m16 & 0xf0u) << 4) | ((m16 & 0xf00u) << 8) | ((m16 & 0xf000u) << 12);
    return spread << (4 * h);
}

template <int NP, bool MASKS>
__global__ __launch_bounds__(X3_THREADS, X3_BIG_WG_PER_CU) void encoder_fwd3_kernel(DeviceAE A, int n, const float *x, const float *pert, float *adv_out,
                                                                    float *pmax, int *parg, int *pcnt, unsigned *masks, FusedAdam fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds_w[];
    using XPN = XP<NP>;
    using NPc = std::integral_constant<int, NP>;
    constexpr int STEP_WORDS = xp_step_words(NP), SLOT_WORDS = xp_slot_words(NP);
    unsigned *ring = lds_w;                                                  // [X3_RING][X3_SLOT_STEPS][4 NP fragments][64 lanes][4 words]
    float *cst = reinterpret_cast<float *>(ring + X3_RING * SLOT_WORDS);      // X3_CONST_FLOATS
    float *redm = cst + X3_CONST_FLOATS;                                      // [4][128]
    int *reda = reinterpret_cast<int *>(redm + 4 * 128), *redc = reda + 4 * 128;
    unsigned *mtile = reinterpret_cast<unsigned *>(redc + 4 * 128);          // [2][X3_POINTS][MASK_WORDS] (MASKS)

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x, b = blockIdx.y, tiles = gridDim.x;
    const int n0 = tile * X3_POINTS + wave * 32;                              // this wave's first point
    const unsigned *img = xp_image(A, NPc{});
    float gmax = 0.f;                                                         // f16x2: largest scaled activation this lane split
    GA_STAMP(0, 0);

    // ---- the wave's points ----
    float pc[3];
    {
        X3Point pt;
        x3_load_point(n, b, n0 + p, x, pert, fa, pt);
        if (h == 0) x3_store_point(pt, adv_out, fa);      // (the two lanes of a point belong to one wave: both have loaded)
        pc[0] = pt.v[0]; pc[1] = pt.v[1]; pc[2] = pt.v[2];
    }
    x3_stage_constants(xp_consts(A, NPc{}), cst, X3_THREADS);
    __syncthreads();                                   // constants visible; nothing of the ring is in flight yet
    GA_STAMP(0, 1);

    // ---- the weight ring ----
    // fetch(t): this wave's fragments (the NP pieces of channel block `wave`, 1 KiB each, lane-linear) of the steps of slot t
    auto fetch = [&](int t) {
#pragma unroll
        for (int u = 0; u < X3_SLOT_STEPS; ++u) {
            const int s = t * X3_SLOT_STEPS + u;
            const unsigned *src = img + (size_t)s * STEP_WORDS + wave * NP * X3_FRAG_WORDS + lane * 4;
            unsigned *dst = ring + (t % X3_RING) * SLOT_WORDS + u * STEP_WORDS + wave * NP * X3_FRAG_WORDS;
#pragma unroll
            for (int q = 0; q < NP; ++q) x3_glds16(src + q * X3_FRAG_WORDS, dst + q * X3_FRAG_WORDS);
        }
    };
#pragma unroll
    for (int t = 0; t < X3_AHEAD; ++t) fetch(t);

    // ReLU mask words of this lane's channels, [h][point of the workgroup][MASK_WORDS] (the two lanes of a point hold
    // complementary bits of every word: OR-ed on the way out)
    unsigned *mrow = mtile + ((size_t)h * X3_POINTS + wave * 32 + p) * MASK_WORDS;

    // ---- layer 0 ----
    XPN act1[4];
    {
        unsigned m01[2];
        x3_layer0<NP, MASKS>(cst, pc, h, act1, m01, gmax);
        if (MASKS) { mrow[0] = m01[0]; mrow[1] = m01[1]; }
    }

    GA_STAMP(0, 2);
    // ---- the step machinery ----
    // Before the first fragment read of slot t (steps 2 t, 2 t + 1): this wave's fetches up to slot t + 1 have landed (counted
    // vmcnt: at most the 2 NP of slot t + 2 stay in flight), then the barrier -- after it EVERY wave's share of slots <= t + 1 is in
    // LDS and every wave is done with slot t - 1, which the fetch of slot t + 3 now overwrites.
    XPN wcur, wnext;                                   // fragments of (step, channel block) in use / requested
    auto frag_read = [&](XPN &d, int s, int cb) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(ring + ((s / X3_SLOT_STEPS) % X3_RING) * SLOT_WORDS + (s % X3_SLOT_STEPS) * STEP_WORDS +
                                                           cb * NP * X3_FRAG_WORDS) + lane;
#pragma unroll
        for (int q = 0; q < NP; ++q) d.p[q] = src[q * 64];
    };
    auto slot_sync = [&](int t) {
        if (t + X3_AHEAD - 1 < X3_SLOTS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP * X3_SLOT_STEPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + X3_AHEAD < X3_SLOTS) fetch(t + X3_AHEAD);
    };
    // one sixteen-k step: acc[cb] += w(s, cb) . a for the four channel blocks.  `side`: VALU work that does not depend on this
    // step (the epilogue of the channel block the NEXT steps consume), placed inside the step so that it can issue between the
    // MFMAs; `after0`: work on acc[0] once its chain is complete (a layer's last step: the first half-epilogues of the boundary
    // ride under the other three blocks' MFMAs instead of standing alone)
    auto step = [&](auto act_is_a, int s, const XPN &a, f32x16 (&acc)[4], auto side, auto after0) {
        constexpr bool ACT_IS_A = decltype(act_is_a)::value;
        if (s % X3_SLOT_STEPS == 0) slot_sync(s / X3_SLOT_STEPS);
        if (s == 0) frag_read(wcur, 0, 0);
        side();
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            if (cb < 3) frag_read(wnext, s, cb + 1);
            else if (s + 1 < X3_STEPS) frag_read(wnext, s + 1, 0);   // (legal: the slot of step s + 1 landed before this slot's barrier)
            xp_mfma<ACT_IS_A>(wcur, a, acc[cb]);
            if (cb == 0) after0();
            if (cb < 3 || s + 1 < X3_STEPS) wcur = wnext;
        }
    };
    auto nothing = [] {};
    auto epilogue_half = [&](const f32x16 &acc, int cb, int gh, const float *scsh /* [scale[C] | shift[C]] */, int C, int coff, XPN &dst,
                             unsigned &m16) {
        x3_epilogue_half<NP, MASKS>(acc, scsh + coff + 32 * cb + 4 * h, scsh + C + coff + 32 * cb + 4 * h, gh, dst, m16, gmax);
    };
    auto mask_store = [&](int word, unsigned m_lo, unsigned m_hi) { mrow[word] = x3_mask_bits(m_lo, m_hi, h); };
    using ActB = std::integral_constant<bool, false>;
    using ActA = std::integral_constant<bool, true>;

    // The pieces of channel block 0 of a finished layer, made under that layer's last step (step's `after0`).
    XPN first[2];
    auto first_epilogue = [&](f32x16 (&accs)[4], const float *scsh, int C, int coff, int mask_word) {
        return [&, scsh, C, coff, mask_word] {
            unsigned ml = 0, mh = 0;
            epilogue_half(accs[0], 0, 0, scsh, C, coff, first[0], ml);
            epilogue_half(accs[0], 0, 1, scsh, C, coff, first[1], mh);
            if (MASKS) mask_store(mask_word, ml, mh);
        };
    };
    // A layer boundary, software-pipelined: the epilogue of channel block cb + 1 of the finished layer (`prev`) rides in the two
    // steps of the next layer that consume block cb's pieces (block 0's are in `first`).  keep: where the pieces are kept for a
    // later pass (h3), or null.  last_after0: the next boundary's first_epilogue, or nothing.
    auto boundary = [&](auto act_is_a, f32x16 (&prev)[4], const float *scsh, int C, int coff, int mask_off, int s0, f32x16 (&next)[4], XPN *keep,
                        auto last_after0) {
        XPN cur[2] = {first[0], first[1]}, nxt[2];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            unsigned nl = 0, nh = 0;
            if (keep) { keep[2 * cb] = cur[0]; keep[2 * cb + 1] = cur[1]; }
            if (cb < 3) {
                step(act_is_a, s0 + 2 * cb, cur[0], next, [&] { epilogue_half(prev[cb + 1], cb + 1, 0, scsh, C, coff, nxt[0], nl); }, nothing);
                step(act_is_a, s0 + 2 * cb + 1, cur[1], next, [&] { epilogue_half(prev[cb + 1], cb + 1, 1, scsh, C, coff, nxt[1], nh); }, nothing);
                if (MASKS) mask_store(mask_off + cb + 1, nl, nh);
                cur[0] = nxt[0]; cur[1] = nxt[1];
            } else {
                step(act_is_a, s0 + 2 * cb, cur[0], next, nothing, nothing);
                step(act_is_a, s0 + 2 * cb + 1, cur[1], next, nothing, last_after0);
            }
        }
    };

    // ---- layer 1: 64 -> 128 (steps 0-3) ----
    f32x16 acc1[4] = {}, acc2[4] = {}, acc3[4] = {}, acc4[4] = {};
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) step(ActB{}, kb, act1[kb], acc1, nothing, nothing);
    step(ActB{}, 3, act1[3], acc1, nothing, first_epilogue(acc1, cst + X3_SC1, 128, 0, MASK_OFF2));
    GA_STAMP(0, 3);
    // ---- layer 2: 128 -> 128 (steps 4-11) ----
    boundary(ActB{}, acc1, cst + X3_SC1, 128, 0, MASK_OFF2, 4, acc2, nullptr, first_epilogue(acc2, cst + X3_SC2, 128, 0, MASK_OFF3));
    GA_STAMP(0, 4);
    // ---- layers 3 + 4 by halves: h4[:, 128 half ..] feeds K half `half` of layer 4 (one chain over K = 256, ascending) ----
    XPN act3[8];
    boundary(ActB{}, acc2, cst + X3_SC2, 128, 0, MASK_OFF3, 12, acc3, act3, first_epilogue(acc3, cst + X3_SC3, 256, 0, MASK_OFF4));   // layer 3, channels 0 .. 127 (steps 12-19)
    boundary(ActA{}, acc3, cst + X3_SC3, 256, 0, MASK_OFF4, 20, acc4, nullptr, nothing);                                             // layer 4, K half 0 (steps 20-27)
    GA_STAMP(0, 5);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc3[cb] = f32x16{};
#pragma unroll
    for (int kb = 0; kb < 7; ++kb) step(ActB{}, 28 + kb, act3[kb], acc3, nothing, nothing);                                          // layer 3, channels 128 .. 255 (steps 28-35)
    step(ActB{}, 35, act3[7], acc3, nothing, first_epilogue(acc3, cst + X3_SC3, 256, 128, MASK_OFF4 + 4));
    boundary(ActA{}, acc3, cst + X3_SC3, 256, 128, MASK_OFF4 + 4, 36, acc4, nullptr, nothing);                                       // layer 4, K half 1 (steps 36-43)

    GA_STAMP(0, 6);
    // f16x2's range guard: a wave that split a scaled activation beyond the fp16 range gives +inf as its pool maxima (the latent,
    // the reconstruction and every loss of that cloud become inf / NaN) and raises the model's flag -- never a silent wrong number
    const bool tripped = NP == 2 && __builtin_amdgcn_ballot_w64(gmax > H2_ACT_LIMIT) != 0;
    if (tripped && lane == 0) atomicOr(A.range_flag, 1);
    // ---- layer 4's BN + ReLU and the max-pool from the registers (lane = channel 32 cb + p, registers = points) ----
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const int col = 32 * cb + p;
        const float sc4 = cst[X3_SC4 + col], sh4 = cst[X3_SC4 + 128 + col];
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(fmaf(acc4[cb][r], sc4, sh4), 0.f);
        if (n0 + 32 > n) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (n0 + acc_row(r, h) >= n) v[r] = -2.f;             // never the maximum, never equal to it
        }
        float mx = -1.f;
        int arg = INT_MAX, cnt = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, v[r]);
#pragma unroll
        for (int r = 15; r >= 0; --r) {                               // rows ascend with r: the last hit kept is the lowest row
            const bool hit = v[r] == mx;
            arg = hit ? n0 + acc_row(r, h) : arg;
            cnt += hit ? 1 : 0;
        }
        const float m2 = __shfl_xor(mx, 32);
        const int a2 = __shfl_xor(arg, 32), k2 = __shfl_xor(cnt, 32);
        if (m2 > mx) { mx = m2; arg = a2; cnt = k2; }
        else if (m2 == mx) { arg = a2 < arg ? a2 : arg; cnt += k2; }
        if (tripped) { mx = INFINITY; arg = n0 < n ? n0 : n - 1; cnt = 1; }
        if (h == 0) { redm[wave * 128 + col] = mx; reda[wave * 128 + col] = arg; redc[wave * 128 + col] = cnt; }
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int c = threadIdx.x;
        float m = redm[c];
        int a = reda[c], k = redc[c];
#pragma unroll
        for (int w = 1; w < 4; ++w) {                                  // waves hold ascending point ranges: the first hit stays
            const float m2 = redm[w * 128 + c];
            if (m2 > m) { m = m2; a = reda[w * 128 + c]; k = redc[w * 128 + c]; }
            else if (m2 == m) { k += redc[w * 128 + c]; }
        }
        const size_t o = ((size_t)b * tiles + tile) * 128 + c;
        pmax[o] = m; parg[o] = a; pcnt[o] = k;
    }
    if (MASKS) {   // the tile's mask rows are contiguous in HBM: the two lanes' halves OR-ed, coalesced stores
        const int t0 = tile * X3_POINTS, live = n - t0 < X3_POINTS ? n - t0 : X3_POINTS;
        unsigned *dst = masks + ((size_t)b * n + t0) * MASK_WORDS;
        for (int e = threadIdx.x; e < live * MASK_WORDS; e += X3_THREADS) dst[e] = mtile[e] | mtile[X3_POINTS * MASK_WORDS + e];
    }
    GA_STAMP(0, 7);
}

// ------------------------------------------------------------------------------------------
// The same forward for launches that would leave CUs idle (few clouds): the four waves of a workgroup share ONE 32-point
// unit, wave w computing output-channel block w of every layer -- a quarter of the MFMAs per wave, four times the workgroups.
// A layer's pieces are exchanged through LDS in exactly the fragment form every wave reads them in (all four waves hold
// the same points, so the B operand of lane (point, h) is the same bytes in each): 6 ds_write_b128 + a barrier + 24
// ds_read_b128 per layer and wave.  A wave uses only the weight fragments of its own channel block -- nothing to share, so
// they come straight into registers, X3S_AHEAD steps ahead (a step is 6 MFMAs here, ~100 ns, against an L2 round trip of
// ~1 us: the LDS ring's three steps of run-ahead left this form waiting for weights, 15.7 us per workgroup on an idle chip).
// Same chain per output element, same bits (tests/test_gpu_encoder_x3.py).
// ------------------------------------------------------------------------------------------
// Two instantiations: <10 steps of weights in flight (120 registers), one workgroup per CU> while the launch has at most one
// workgroup per CU anyway (B <= 4 at N = 2048), <6 steps (72 registers: 236 in all), two workgroups per CU> beyond -- the second
// workgroup's exchanges and prologue run under the first one's MFMAs (B = 8: 0.0761 -> 0.0691 ms per iteration; B = 4 with the
// shallower run-ahead: 0.0634 -> 0.0641).
__host__ __device__ constexpr int x3s_xbuf_words(int np) { return 8 * np * X3_FRAG_WORDS; }   // one layer's pieces: 8 sixteen-k blocks x NP x 1 KiB
constexpr size_t x3s_lds_bytes(int np, bool masks) {
    return 2 * x3s_xbuf_words(np) * 4 + sizeof(float) * (X3_CONST_FLOATS + 32 * 3) + (masks ? sizeof(unsigned) * 32 * MASK_WORDS : 0);
}

template <int NP, bool MASKS, int X3S_AHEAD, int WG_PER_CU>
__global__ __launch_bounds__(X3_THREADS, WG_PER_CU) void encoder_fwd3s_kernel(DeviceAE A, int n, const float *x, const float *pert, float *adv_out,
                                                                     float *pmax, int *parg, int *pcnt, unsigned *masks, FusedAdam fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds_w[];
    using XPN = XP<NP>;
    using NPc = std::integral_constant<int, NP>;
    constexpr int STEP_WORDS = xp_step_words(NP), XBUF_WORDS = x3s_xbuf_words(NP);
    unsigned *xbuf = lds_w;                                                  // [2][8][NP][64 lanes][4 words]
    float *cst = reinterpret_cast<float *>(xbuf + 2 * XBUF_WORDS);
    float *pcs = cst + X3_CONST_FLOATS;                                      // [32][3] the unit's points
    unsigned *mtile = reinterpret_cast<unsigned *>(pcs + 32 * 3);            // [32 points][MASK_WORDS] (MASKS)

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x, b = blockIdx.y, tiles = gridDim.x;
    const int n0 = tile * 32;
    GA_STAMP(1, 0);
    // this wave's fragments of step s: NP KiB at a fixed stride -- buffer loads with the step as an immediate offset
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(xp_image(A, NPc{})) + wave * NP * X3_FRAG_WORDS, 0, 0x7fffffff, 0x00020000);
    const unsigned lb = (unsigned)lane * 16u;
    float gmax = 0.f;
    XPN wr[X3S_AHEAD];
    auto fetch = [&](int s) {
#pragma unroll
        for (int q = 0; q < NP; ++q) wr[s % X3S_AHEAD].p[q] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, lb, (s * STEP_WORDS + q * X3_FRAG_WORDS) * 4, 0);
    };
    // the unit's points: wave 0 loads them (and applies / stores the pending Adam step), the others take them from LDS; the small
    // loads go first (loads return in order: behind thirty weight fragments they would wait for all of those)
    if (wave == 0) {
        X3Point pt;
        x3_load_point(n, b, n0 + p, x, pert, fa, pt);
        if (h == 0) {
            x3_store_point(pt, adv_out, fa);
            pcs[p * 3] = pt.v[0]; pcs[p * 3 + 1] = pt.v[1]; pcs[p * 3 + 2] = pt.v[2];
        }
    }
    x3_stage_constants(xp_consts(A, NPc{}), cst, X3_THREADS);
#pragma unroll
    for (int s = 0; s < X3S_AHEAD; ++s) fetch(s);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (a raw barrier: __syncthreads()'s fence would drain the weight loads)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    float pc[3] = {pcs[p * 3], pcs[p * 3 + 1], pcs[p * 3 + 2]};
    GA_STAMP(1, 1);

    XPN act1[4];
    {
        unsigned m01[2];
        x3_layer0<NP, MASKS>(cst, pc, h, act1, m01, gmax);
        if (MASKS && wave == 0) {
            m01[0] |= __shfl_xor(m01[0], 32); m01[1] |= __shfl_xor(m01[1], 32);
            if (h == 0) { mtile[p * MASK_WORDS] = m01[0]; mtile[p * MASK_WORDS + 1] = m01[1]; }
        }
    }
    auto step = [&](auto act_is_a, int s, const XPN &a, f32x16 &acc) {
        constexpr bool ACT_IS_A = decltype(act_is_a)::value;
        xp_mfma<ACT_IS_A>(wr[s % X3S_AHEAD], a, acc);
        if (s + X3S_AHEAD < X3_STEPS) fetch(s + X3S_AHEAD);
    };
    // epilogue of this wave's channel block -> the pieces of sixteen-k blocks 2 w, 2 w + 1 into exchange buffer `buf`, then all eight back
    auto exchange = [&](const f32x16 &acc, const float *scsh, int C, int coff, int mask_word, int buf, XPN (&out)[8]) {
        XPN lo, hi;
        unsigned ml = 0, mh = 0;
        x3_epilogue_half<NP, MASKS>(acc, scsh + coff + 32 * wave + 4 * h, scsh + C + coff + 32 * wave + 4 * h, 0, lo, ml, gmax);
        x3_epilogue_half<NP, MASKS>(acc, scsh + coff + 32 * wave + 4 * h, scsh + C + coff + 32 * wave + 4 * h, 1, hi, mh, gmax);
        u32x4 *xb = reinterpret_cast<u32x4 *>(xbuf + buf * XBUF_WORDS) + lane;
#pragma unroll
        for (int q = 0; q < NP; ++q) { xb[((2 * wave) * NP + q) * 64] = lo.p[q]; xb[((2 * wave + 1) * NP + q) * 64] = hi.p[q]; }
        if (MASKS) {
            unsigned m = x3_mask_bits(ml, mh, h);
            m |= __shfl_xor(m, 32);
            if (h == 0) mtile[p * MASK_WORDS + mask_word + wave] = m;
        }
        // a raw barrier: __syncthreads() would drain the weight loads in flight (its fence waits for vmcnt(0))
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int kb = 0; kb < 8; ++kb)
#pragma unroll
            for (int q = 0; q < NP; ++q) out[kb].p[q] = xb[(kb * NP + q) * 64];
    };
    using ActB = std::integral_constant<bool, false>;
    using ActA = std::integral_constant<bool, true>;

    XPN act2[8], act3[8], act4[8];
    {
        f32x16 acc = {};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) step(ActB{}, kb, act1[kb], acc);
        GA_STAMP(1, 2);
        exchange(acc, cst + X3_SC1, 128, 0, MASK_OFF2, 0, act2);
        GA_STAMP(1, 3);
    }
    {
        f32x16 acc = {};
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) step(ActB{}, 4 + kb, act2[kb], acc);
        GA_STAMP(1, 4);
        exchange(acc, cst + X3_SC2, 128, 0, MASK_OFF3, 1, act3);
        GA_STAMP(1, 5);
    }
    f32x16 acc4 = {};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x16 acc = {};
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) step(ActB{}, (half ? 28 : 12) + kb, act3[kb], acc);
        exchange(acc, cst + X3_SC3, 256, 128 * half, MASK_OFF4 + 4 * half, half, act4);
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) step(ActA{}, (half ? 36 : 20) + kb, act4[kb], acc4);
    }
    GA_STAMP(1, 6);
    const bool tripped = NP == 2 && __builtin_amdgcn_ballot_w64(gmax > H2_ACT_LIMIT) != 0;   // (the range guard, as above: this wave's channels)
    if (tripped && lane == 0) atomicOr(A.range_flag, 1);
    {   // BN + ReLU and the pool of this wave's 32 channels over the unit's 32 points
        const int col = 32 * wave + p;
        const float sc4 = cst[X3_SC4 + col], sh4 = cst[X3_SC4 + 128 + col];
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(fmaf(acc4[r], sc4, sh4), 0.f);
        if (n0 + 32 > n) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (n0 + acc_row(r, h) >= n) v[r] = -2.f;
        }
        float mx = -1.f;
        int arg = INT_MAX, cnt = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, v[r]);
#pragma unroll
        for (int r = 15; r >= 0; --r) {
            const bool hit = v[r] == mx;
            arg = hit ? n0 + acc_row(r, h) : arg;
            cnt += hit ? 1 : 0;
        }
        const float m2 = __shfl_xor(mx, 32);
        const int a2 = __shfl_xor(arg, 32), k2 = __shfl_xor(cnt, 32);
        if (m2 > mx) { mx = m2; arg = a2; cnt = k2; }
        else if (m2 == mx) { arg = a2 < arg ? a2 : arg; cnt += k2; }
        if (tripped) { mx = INFINITY; arg = n0 < n ? n0 : n - 1; cnt = 1; }
        if (h == 0) {
            const size_t o = ((size_t)b * tiles + tile) * 128 + col;
            pmax[o] = mx; parg[o] = arg; pcnt[o] = cnt;
        }
    }
    if (MASKS) {
        __syncthreads();
        const int live = n - n0 < 32 ? n - n0 : 32;
        unsigned *dst = masks + ((size_t)b * n + n0) * MASK_WORDS;
        for (int e = threadIdx.x; e < live * MASK_WORDS; e += X3_THREADS) dst[e] = mtile[e];
    }
    GA_STAMP(1, 7);
}

// Points per workgroup of the x3 forward for a batch of b clouds: 32 (the split form) while 128-point workgroups would leave
// half of the CUs without one.
int encoder_x3_points(int b, int n) { return (long)b * cdiv(n, X3_POINTS) * 2 <= (long)X3_SPLIT_HALVES * kCUs ? 32 : X3_POINTS; }

template <int NP>
static int launch_encoder_fwd_xp(const DeviceAE &A, int b, const float *x, const float *pert, float *adv_out, float *pmax, int *parg,
                                 int *pcnt, unsigned *masks, hipStream_t stream, hipEvent_t start, hipEvent_t stop, const FusedAdam &fa) {
    static DeviceOnce attr;
    if (int rc = attr.run([]() -> int {
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_fwd3_kernel<NP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)x3_lds_bytes(NP, true)));
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_fwd3_kernel<NP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)x3_lds_bytes(NP, false)));
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_fwd3s_kernel<NP, true, 10, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)x3s_lds_bytes(NP, true)));
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_fwd3s_kernel<NP, false, 10, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)x3s_lds_bytes(NP, false)));
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_fwd3s_kernel<NP, true, 6, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)x3s_lds_bytes(NP, true)));
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_fwd3s_kernel<NP, false, 6, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)x3s_lds_bytes(NP, false)));
            return GEOADV_OK;
        })) return rc;
    const dim3 block(X3_THREADS);
    if (encoder_x3_points(b, A.n_points) == 32) {
        const dim3 grid(cdiv(A.n_points, 32), b);
        const unsigned lds = (unsigned)x3s_lds_bytes(NP, masks != nullptr);
        const bool two = (long)grid.x * b > kCUs;          // more workgroups than CUs: two per CU
#define X3S_LAUNCH(M, AH, W)                                                                                                                     \
        do {                                                                                                                                      \
            if (start && stop) hipExtLaunchKernelGGL((encoder_fwd3s_kernel<NP, M, AH, W>), grid, block, lds, stream, start, stop, 0, A, A.n_points, x, pert, adv_out, pmax, parg, pcnt, masks, fa); \
            else encoder_fwd3s_kernel<NP, M, AH, W><<<grid, block, lds, stream>>>(A, A.n_points, x, pert, adv_out, pmax, parg, pcnt, masks, fa);  \
        } while (0)
        if (masks) { if (two) X3S_LAUNCH(true, 6, 2); else X3S_LAUNCH(true, 10, 1); }
        else { if (two) X3S_LAUNCH(false, 6, 2); else X3S_LAUNCH(false, 10, 1); }
#undef X3S_LAUNCH
        GA_LAUNCH_CHECK();
        return GEOADV_OK;
    }
    const dim3 grid(cdiv(A.n_points, X3_POINTS), b);
    const unsigned lds = (unsigned)x3_lds_bytes(NP, masks != nullptr);
    if (masks) {
        if (start && stop) hipExtLaunchKernelGGL((encoder_fwd3_kernel<NP, true>), grid, block, lds, stream, start, stop, 0, A, A.n_points, x, pert, adv_out, pmax, parg, pcnt, masks, fa);
        else encoder_fwd3_kernel<NP, true><<<grid, block, lds, stream>>>(A, A.n_points, x, pert, adv_out, pmax, parg, pcnt, masks, fa);
    } else {
        if (start && stop) hipExtLaunchKernelGGL((encoder_fwd3_kernel<NP, false>), grid, block, lds, stream, start, stop, 0, A, A.n_points, x, pert, adv_out, pmax, parg, pcnt, masks, fa);
        else encoder_fwd3_kernel<NP, false><<<grid, block, lds, stream>>>(A, A.n_points, x, pert, adv_out, pmax, parg, pcnt, masks, fa);
    }
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

// the forward in the model's piece arithmetic (bf16x3 or f16x2: DeviceAE::enc_arith)
int launch_encoder_fwd_x3(const DeviceAE &A, int b, const float *x, const float *pert, float *adv_out, float *pmax, int *parg,
                          int *pcnt, unsigned *masks, hipStream_t stream, hipEvent_t start, hipEvent_t stop, const FusedAdam &fa) {
    if (A.enc_arith == GEOADV_ENC_ARITH_F16X2) return launch_encoder_fwd_xp<2>(A, b, x, pert, adv_out, pmax, parg, pcnt, masks, stream, start, stop, fa);
    return launch_encoder_fwd_xp<3>(A, b, x, pert, adv_out, pmax, parg, pcnt, masks, stream, start, stop, fa);
}

}  // namespace geoadv
GA_STAMPS_GETTER(geoadv_debug_stamps_encoder_x3)
