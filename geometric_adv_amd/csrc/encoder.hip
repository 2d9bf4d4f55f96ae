// PointNet-style encoder of the victim auto-encoder, forward and backward-to-input, for gfx950.
//
// Reference semantics: src/encoders_decoders.py:37-72 with the widths of src/ae_templates.py:22
// (3 -> 64 -> 128 -> 128 -> 256 -> 128): five per-point layers [conv1d k=1 (= x@W + b),
// batch-norm in inference mode, ReLU], then a max over the points of each cloud.  In the
// reference that is ~25 TF ops per forward; here it is ONE kernel: a workgroup owns a tile of
// points, keeps their activations in LDS, chains the four wide layers on
// v_mfma_f32_32x32x2_f32 (exact fp32: the 1e-5 Chamfer tolerance rules out bf16/fp16 operands)
// and reduces the symmetric max-pool in registers.  Nothing but the points, the weights (L2
// resident, pre-packed into MFMA fragment order) and 3*128 words per tile touches HBM -- the
// kernel is MFMA bound.
//
// Backward-to-input (weights are frozen, var_list = pert only, adv_ae.py:153): the max-pool
// passes gradient only to the <= 128 "critical" points of a cloud, so the backward kernel
// re-runs the forward for just those rows (keeping the ReLU masks as bytes in LDS) and chains
// the transposed layers.  Exact ties in the pool are handled like TF's _MinOrMaxGrad (equal
// split): a cloud with a tied positive maximum is flagged and processed densely instead.
//
// Bit-identity of the recompute: the backward decides "is this row the arg-max" by comparing its
// recomputed h5 with z, so every forward layer has ONE canonical accumulation order, whatever
// the tile height or the wave assignment: layers 1 and 2 sum two independent K-half chains
// (half0 + half1), layers 3 and 4 run one chain in ascending k.
#include "ae.h"
#include "mfma_tile.h"
#include "encoder_jac.h"
#include "encoder_x3.h"
#include "decoder_tail.h"
#include <hip/hip_ext.h>
#include <limits.h>
#include <stdlib.h>

namespace geoadv {

constexpr int fwd_kc(int nout) { return nout == 256 ? 1 : 2; }   // the canonical order of forward layers 1-3
constexpr int KC_L4 = 1;                                         // layer 4 (K = 256): one chain in ascending k

template <int ROWS> struct EncLds {
    static constexpr int P_FLOATS = ROWS * (256 + 4);
    static constexpr int Q_FLOATS = ROWS * (128 + 4);
    static constexpr int SCRATCH_FLOATS = ROWS == 32 ? 3 * 2 * 16 * 64 : 16 * 64 * 4;   // K-part hand-off
    static constexpr int MASK_BYTES = ROWS * (64 + 128 + 128 + 256);
    static constexpr size_t bwd_bytes = sizeof(float) * (P_FLOATS + Q_FLOATS + SCRATCH_FLOATS + ROWS * 3) + sizeof(int) * ROWS + MASK_BYTES;
};

// Hidden forward layer: out = relu(in @ W * scale + shift) (+ ReLU mask bytes for the backward).
template <int ROWS, int NOUT, bool SAVE_MASK>
__device__ __forceinline__ void fwd_layer(const float *in, int s_in, float *out, int s_out, const PackedLayer &L,
                                          const float *scale, const float *shift, unsigned char *mask, float *scratch) {
    const int i = threadIdx.x & 31;
    const int wave = threadIdx.x >> 6;
    constexpr int CB = NOUT / 32;
    const int col_of_lane = ((wave % ((CB * (ROWS / 32) > 8) ? CB : CB * (ROWS / 32))) % CB) * 32 + i;
    const float sc = scale[col_of_lane], sh = shift[col_of_lane];
    layer_gemm<ROWS, NOUT, fwd_kc(NOUT)>(in, s_in, L, scratch, [&](int row, int col, float a) {
        const float v = fmaxf(fmaf(a, sc, sh), 0.f);
        out[row * s_out + col] = v;
        if (SAVE_MASK) mask[row * NOUT + col] = v > 0.f;
    });
}

// Layer 0 (fan-in 3) on the VALU: 8 channels per thread.
template <int ROWS, bool SAVE_MASK>
__device__ __forceinline__ void fwd_layer0(const float *pts /*LDS [ROWS][3]*/, float *out, int s_out, const DeviceAE &A,
                                           unsigned char *mask) {
    const int C1 = 64;
    for (int e = threadIdx.x; e < ROWS * 8; e += ENC_THREADS) {
        const int row = e >> 3, c0 = (e & 7) * 8;
        const float x = pts[row * 3], y = pts[row * 3 + 1], z = pts[row * 3 + 2];
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
}

// ------------------------------------------------------------------------------------------
// Forward kernel.  grid = (tiles of 64 points per cloud, clouds).  Outputs per tile and channel: the maximum of h5 over
// the tile's valid rows, the first row attaining it, and how many rows attain it.  Same arithmetic and canonical
// accumulation order as the recomputing backward below; the 256-wide h4 is never held whole: layer 3 is computed in
// two column halves and each half is consumed at once by the matching K-half of layer 4 (whose
// canonical order IS "K-half 0 + K-half 1").  Two 64 x 132 buffers (67.6 KB) instead of 100 KB, so
// TWO workgroups share a CU and one's epilogues / barriers / first-operand latencies hide under
// the other's MFMAs.
// ------------------------------------------------------------------------------------------
// ReLU masks of h1..h4 for the sparse backward (which then needs no forward recompute): MASK_WORDS 32-bit words per
// point, bit c&31 of word OFF_L + (c>>5) = [h_L[c] > 0]; h1: words 0-1, h2: 2-5, h3: 6-9, h4: 10-17.  The h5 mask is
// implied (the arg-max of a channel with z > 0 is positive).  Staged per tile in LDS, written out coalesced.
// (MASK_WORDS, MASK_OFF2..4: encoder_jac.h)
// ROWS = 64: 8 waves, two workgroups per CU (the B = 32 shape).  ROWS = 32: 4 waves owning one row block -- the same chains,
// the same bits -- for launches that would otherwise leave CUs without a tile (batch * n / 64 < 256, i.e. B <= 7 at
// N = 2048: what each GPU sees when ONE batch of 32 is split over 8): a tile's five dependent layers are ~19 us of MFMA
// time on one CU at 64 rows however few tiles there are, ~9 us at 32.
template <int ROWS> struct Fwd2 {
    static constexpr int THREADS = ROWS * 8;
    static constexpr int BUF_FLOATS = ROWS * 132;
    static constexpr size_t LDS_BYTES = sizeof(float) * (2 * BUF_FLOATS + ROWS * 3 + 256) + sizeof(int) * 512;
    static constexpr size_t LDS_BYTES_MASKS = LDS_BYTES + sizeof(unsigned) * ROWS * MASK_WORDS;
};

// lane r (< 16) collects the two mask words of accumulator register r: rows acc_row(r, 0) and acc_row(r, 1).
// v_writelane_b32 moves each half of the ballot (an SGPR pair) into that lane with ONE VALU instruction; the portable
// form `if (lane == r) { wl = lo; wh = hi; }` costs a v_mov + v_cndmask per word (a select cannot read two SGPR operands).
#define GA_WRITELANE_CASE(L) case L: asm("v_writelane_b32 %0, %1, " #L : "+v"(dst) : "s"(src)); break;
__device__ __forceinline__ void writelane16(unsigned &dst, unsigned src, int lane_sel) {   // lane_sel: constant after unrolling
    switch (lane_sel) {
        GA_WRITELANE_CASE(0) GA_WRITELANE_CASE(1) GA_WRITELANE_CASE(2) GA_WRITELANE_CASE(3)
        GA_WRITELANE_CASE(4) GA_WRITELANE_CASE(5) GA_WRITELANE_CASE(6) GA_WRITELANE_CASE(7)
        GA_WRITELANE_CASE(8) GA_WRITELANE_CASE(9) GA_WRITELANE_CASE(10) GA_WRITELANE_CASE(11)
        GA_WRITELANE_CASE(12) GA_WRITELANE_CASE(13) GA_WRITELANE_CASE(14) GA_WRITELANE_CASE(15)
    }
}
#undef GA_WRITELANE_CASE
#define MASK_COLLECT(r, positive, wl, wh)                                   \
    do {                                                                    \
        const unsigned long long bal_ = __ballot(positive);                 \
        writelane16(wl, (unsigned)bal_, r);                                 \
        writelane16(wh, (unsigned)(bal_ >> 32), r);                         \
    } while (0)

// B-fragment ring carried ACROSS chains: while the last four k-groups of a chain run, the freed
// slots are refilled with the first four fragments of the NEXT chain (weights do not depend on the
// activations, so the request may cross the epilogue and the barrier).  Without it every chain start
// exposes one L2 round trip (~900 cycles, 8 chains per tile = 13 % of the tile).
// (BRing, ring_fill: mfma_tile.h, shared with the training kernels)
// NT (chain length in k-groups) is a compile-time constant -- the encoder widths are fixed (ae_create checks them) -- so
// the chain is fully unrolled: the first MFMA takes the inline constant 0 as its accumulator (no 16 x v_mov per chain),
// every A / B address is base + immediate, and no select survives.  That matters more than it looks: plain VALU
// instructions do NOT overlap with the matrix pipe on this part (tools/mfma_probe.py: 4 MFMA + 16 v_add_f32 per group
// runs 14 % slower than the MFMAs alone), so every VALU instruction of this kernel is paid in MFMA time.
template <int NT, bool HAS_NEXT>
__device__ __forceinline__ void chain_ring(const float *ar, int at0, const FragSrc &cur, unsigned lb, BRing &ring,
                                           const FragSrc &next, f32x16 (&acc)[1]) {
    static_assert(NT % 4 == 0, "chain lengths are multiples of four k-groups");
    float4 a0[1], a1[1];
    a0[0] = *reinterpret_cast<const float4 *>(ar + 8 * at0);
#pragma unroll
    for (int t = 0; t < NT; t += 4) {
        // every refill is UNCONDITIONAL (an always valid address: the chain's own first fragments if nothing follows):
        // with a branch around a load the compiler can no longer count outstanding loads and degrades the
        // s_waitcnt vmcnt(3) below to vmcnt(2)/(1)/(0)
        const bool more = t + 4 < NT;
        const FragSrc &src = more ? cur : (HAS_NEXT ? next : cur);
        const int g = more ? t + 4 : 0;
        a1[0] = *reinterpret_cast<const float4 *>(ar + 8 * (at0 + t + 1));
        __builtin_amdgcn_sched_barrier(0);       // the next A fragment is requested BEFORE this group's MFMAs issue
        mfma_group<1>(a0, ring.b[0], acc);
        ring.b[0] = ld_frag(src, g + 0, lb);
        __builtin_amdgcn_sched_barrier(0);       // keep the refill HERE (the scheduler would sink all four to the loop end)
        a0[0] = *reinterpret_cast<const float4 *>(ar + 8 * (at0 + t + 2));
        __builtin_amdgcn_sched_barrier(0);
        mfma_group<1>(a1, ring.b[1], acc);
        ring.b[1] = ld_frag(src, g + 1, lb);
        __builtin_amdgcn_sched_barrier(0);
        a1[0] = *reinterpret_cast<const float4 *>(ar + 8 * (at0 + t + 3));
        __builtin_amdgcn_sched_barrier(0);
        mfma_group<1>(a0, ring.b[2], acc);
        ring.b[2] = ld_frag(src, g + 2, lb);
        __builtin_amdgcn_sched_barrier(0);
        a0[0] = *reinterpret_cast<const float4 *>(ar + 8 * (at0 + (more ? t + 4 : t)));
        __builtin_amdgcn_sched_barrier(0);
        mfma_group<1>(a1, ring.b[3], acc);
        ring.b[3] = ld_frag(src, g + 3, lb);
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ int orow_of(int rb) { return rb * 32; }

template <bool MASKS, int ROWS>
__global__ __launch_bounds__(ROWS * 8, 4) void encoder_fwd2_kernel(DeviceAE A, int n, const float *x, const float *pert,
                                                                  float *adv_out, float *pmax, int *parg, int *pcnt,
                                                                  unsigned *masks, FusedAdam fa) {
    constexpr int THREADS = Fwd2<ROWS>::THREADS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    GA_STAMP(0, 0);
    float *bufA = lds;
    float *bufB = bufA + Fwd2<ROWS>::BUF_FLOATS;
    float *pts = bufB + Fwd2<ROWS>::BUF_FLOATS;       // [ROWS][3]
    float *redm = pts + ROWS * 3;                     // [2][128]
    int *reda = reinterpret_cast<int *>(redm + 256);
    int *redc = reda + 256;
    unsigned *mtile = reinterpret_cast<unsigned *>(redc + 256);   // [64][MASK_WORDS] (MASKS only)

    // the wave index as a SCALAR: everything derived from it (weight bases, row block offsets) then lives in SGPRs, the
    // B-fragment loads use the SGPR-base + lane-offset form and their address updates are SALU work -- VALU instructions
    // cost matrix-pipe time here, SALU instructions do not
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, i = lane & 31;
    const int cb = wave & 3, rb = wave >> 2;          // this wave's (column block, row block) unit in every 128-wide product
    // fragment bases of this wave's eight chains
    constexpr int kg1 = 64 >> 3, kg2 = 128 >> 3, kg3 = 128 >> 3, kg4 = 256 >> 3;   // k-groups of layers 1-4 (widths fixed, ae.hip)
    const FragSrc w1 = frag_src(A.enc_fwd[1].w, cb * kg1), w2 = frag_src(A.enc_fwd[2].w, cb * kg2);     // wave-uniform
    const FragSrc w3a = frag_src(A.enc_fwd[3].w, cb * kg3), w3b = frag_src(A.enc_fwd[3].w, (4 + cb) * kg3);
    const FragSrc w4 = frag_src(A.enc_fwd[4].w, cb * kg4);
    const unsigned lb = (unsigned)lane * 16u;         // this lane's 16 bytes of every 1 KiB fragment
    // First of all, the loads the first barrier waits for (loads return in order: issued behind the fourteen per-lane
    // constants below they would wait for those too): the tile's points, and layer 0's 320 constants (W0 [3][64], scale,
    // shift), which go through LDS -- one global load per thread instead of the forty that "every thread fetches the 8
    // channels it computes" costs.  They are only REQUESTED here; the LDS stores follow the other requests.
    float *l0c = redm;                                 // (the pool's reduction buffers are free until the end)
    const int tile = blockIdx.x, b = blockIdx.y, tiles = gridDim.x;
    const int n0 = tile * ROWS;
    float pv = 0.f, pp = 0.f, ag = 0.f, agd = 0.f, am = 0.f, av = 0.f;
    size_t pg = 0;
    bool pvalid = false;
    constexpr int L0_LOADERS = THREADS - ROWS * 3;    // 320 (64-row form) or 160 threads: one or two constants each
    constexpr int L0_PER = (320 + L0_LOADERS - 1) / L0_LOADERS;
    float l0v[L0_PER] = {};
    if (threadIdx.x < ROWS * 3) {
        const int r = threadIdx.x / 3, a = threadIdx.x % 3;
        int p = n0 + r;
        pvalid = p < n;
        p = pvalid ? p : n - 1;
        pg = ((size_t)b * n + p) * 3 + a;
        pv = x[pg];
        if (pert) pp = pert[pg];
        if (fa.m) { ag = fa.g_enc[pg]; agd = fa.g_dist[pg]; am = fa.m[pg]; av = fa.v[pg]; }      // (uniform branch)
    } else {
#pragma unroll
        for (int q = 0; q < L0_PER; ++q) {
            const int e = threadIdx.x - ROWS * 3 + q * L0_LOADERS;
            if (e < 320) l0v[q] = e < 192 ? A.w0[e] : (e < 256 ? A.scale[0][e - 192] : A.shift[0][e - 256]);
        }
    }
    BRing ring;
    ring_fill(ring, w1, lb);                          // in flight during layer 0
    // every per-lane constant is requested up front too (a load at its point of use costs an exposed
    // L2 round trip per epilogue): BN scale/shift of this lane's column in layers 1-4 ...
    const int ccol = cb * 32 + i;
    const float sc1 = A.scale[1][ccol], sh1 = A.shift[1][ccol], sc2 = A.scale[2][ccol], sh2 = A.shift[2][ccol];
    const float sc3a = A.scale[3][ccol], sh3a = A.shift[3][ccol], sc3b = A.scale[3][128 + ccol], sh3b = A.shift[3][128 + ccol];
    const float sc4 = A.scale[4][ccol], sh4 = A.shift[4][ccol];
    if (threadIdx.x < ROWS * 3) {
        float v = pert ? pv + pp : pv;
        if (fa.m) {   // the pending Adam step of this coordinate (attack.hip adam_kernel, the same operations in the same order)
            float g = ag;
            g += agd;
            float m = am, vv = av;
            m += (g - m) * fa.one_minus_b1;
            vv += (g * g - vv) * fa.one_minus_b2;
            const float pnew = pp - (m * fa.alpha) / (sqrtf(vv) + fa.eps);
            v = pv + pnew;
            if (pvalid) {   // (padding rows repeat the cloud's last point and are never pooled: they only must not store)
                fa.g_enc[pg] = 0.f;
                if (fa.grad_out) fa.grad_out[pg] = g;
                fa.m[pg] = m; fa.v[pg] = vv; fa.pert[pg] = pnew;
            }
        }
        pts[threadIdx.x] = v;
        if (adv_out && pvalid) adv_out[pg] = v;
    } else {
#pragma unroll
        for (int q = 0; q < L0_PER; ++q) {
            const int e = threadIdx.x - ROWS * 3 + q * L0_LOADERS;
            if (e < 320) l0c[e] = l0v[q];
        }
    }
    __syncthreads();
    {   // layer 0 (fan-in 3) on the VALU, same arithmetic as fwd_layer0
        const int row = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 8;
        const float px = pts[row * 3], py = pts[row * 3 + 1], pz = pts[row * 3 + 2];
        float l0w[3][8], l0s[8], l0t[8];               // the 8 channels this thread computes (row = tid/8, channels 8*(tid%8) ..)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            l0w[0][c] = l0c[c0 + c]; l0w[1][c] = l0c[64 + c0 + c]; l0w[2][c] = l0c[128 + c0 + c];
            l0s[c] = l0c[192 + c0 + c]; l0t[c] = l0c[256 + c0 + c];
        }
        unsigned bits = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float a = px * l0w[0][c];
            a = fmaf(py, l0w[1][c], a);
            a = fmaf(pz, l0w[2][c], a);
            const float v = fmaxf(fmaf(a, l0s[c], l0t[c]), 0.f);
            bufA[row * 68 + c0 + c] = v;
            if (MASKS) bits |= (v > 0.f ? 1u : 0u) << c;
        }
        if (MASKS) reinterpret_cast<unsigned char *>(mtile)[row * (4 * MASK_WORDS) + (threadIdx.x & 7)] = (unsigned char)bits;
    }
    __syncthreads();
    GA_STAMP(0, 1);
    const int mrow = orow_of(rb) + (lane & 3) + 8 * ((lane & 15) >> 2);    // lanes 0-15: row acc_row(lane, 0) of this wave's block

    const int orow = orow_of(rb);                     // accumulator rows of this wave: orow + acc_row(r, h)
    // epilogue stores: one lane-dependent base per buffer, the register's row as an immediate offset
    float *bufA_w = bufA + (orow + 4 * h) * 132 + ccol, *bufB_w = bufB + (orow + 4 * h) * 132 + ccol;
    // ---- layer 1: 64 -> 128, canonical K halves (4 + 4 k-groups) ----
    {
        const float *ar = bufA + (orow + i) * 68 + 4 * h;
        f32x16 acc[1] = {}, part[1] = {};
        chain_ring<kg1 / 2, true>(ar, 0, w1, lb, ring, frag_at(w1, kg1 / 2), acc);
        chain_ring<kg1 / 2, true>(ar, kg1 / 2, frag_at(w1, kg1 / 2), lb, ring, w2, part);
        unsigned wl = 0, wh = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = fmaxf(fmaf(acc[0][r] + part[0][r], sc1, sh1), 0.f);
            bufB_w[acc_row(r, 0) * 132] = v;
            if (MASKS) MASK_COLLECT(r, v > 0.f, wl, wh);
        }
        if (MASKS && lane < 16) { mtile[mrow * MASK_WORDS + MASK_OFF2 + cb] = wl; mtile[(mrow + 4) * MASK_WORDS + MASK_OFF2 + cb] = wh; }
    }
    __syncthreads();
    // ---- layer 2: 128 -> 128, canonical K halves (8 + 8) ----
    {
        const float *ar = bufB + (orow + i) * 132 + 4 * h;
        f32x16 acc[1] = {}, part[1] = {};
        chain_ring<kg2 / 2, true>(ar, 0, w2, lb, ring, frag_at(w2, kg2 / 2), acc);
        chain_ring<kg2 / 2, true>(ar, kg2 / 2, frag_at(w2, kg2 / 2), lb, ring, w3a, part);
        unsigned wl = 0, wh = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = fmaxf(fmaf(acc[0][r] + part[0][r], sc2, sh2), 0.f);
            bufA_w[acc_row(r, 0) * 132] = v;
            if (MASKS) MASK_COLLECT(r, v > 0.f, wl, wh);
        }
        if (MASKS && lane < 16) { mtile[mrow * MASK_WORDS + MASK_OFF3 + cb] = wl; mtile[(mrow + 4) * MASK_WORDS + MASK_OFF3 + cb] = wh; }
    }
    __syncthreads();
    // ---- layers 3 + 4 interleaved by halves: h4[:, 128*half ..] feeds K-half `half` of layer 4 ----
    f32x16 acc4[1] = {};                              // layer 4: ONE chain over K = 256 (canonical), fed half by half
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        {
            const float *ar = bufA + (orow + i) * 132 + 4 * h;
            f32x16 acc[1] = {};
            chain_ring<kg3, true>(ar, 0, half ? w3b : w3a, lb, ring, frag_at(w4, (kg4 / 2) * half), acc);   // one full-K chain
            const float sc = half ? sc3b : sc3a, sh = half ? sh3b : sh3a;
            unsigned wl = 0, wh = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = fmaxf(fmaf(acc[0][r], sc, sh), 0.f);
                bufB_w[acc_row(r, 0) * 132] = v;
                if (MASKS) MASK_COLLECT(r, v > 0.f, wl, wh);
            }
            if (MASKS && lane < 16) {
                mtile[mrow * MASK_WORDS + MASK_OFF4 + 4 * half + cb] = wl; mtile[(mrow + 4) * MASK_WORDS + MASK_OFF4 + 4 * half + cb] = wh;
            }
        }
        __syncthreads();
        {
            const float *ar = bufB + (orow + i) * 132 + 4 * h;
            if (half == 0) chain_ring<kg4 / 2, true>(ar, 0, w4, lb, ring, w3b, acc4);
            else chain_ring<kg4 / 2, false>(ar, 0, frag_at(w4, kg4 / 2), lb, ring, w4, acc4);
        }
        if (half == 0) __syncthreads();               // bufB is rewritten by the second half of layer 3
    }
    GA_STAMP(0, 2);
    // BN + ReLU and the max-pool from the registers: maximum, FIRST row attaining it, number of rows attaining it.
    // Two branch-free passes (max, then compare) -- a third of the VALU instructions of the if / else-if form.
    const int col = ccol;
    float mx = -1.f;
    int arg = INT_MAX, cnt = 0;
    {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(fmaf(acc4[0][r], sc4, sh4), 0.f);
        if (n0 + ROWS > n) {                              // only the last tile of a cloud can hold rows beyond n (uniform branch)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (n0 + orow + acc_row(r, h) >= n) v[r] = -2.f;            // never the maximum, never equal to it
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, v[r]);
#pragma unroll
        for (int r = 15; r >= 0; --r) {                   // rows ascend with r: the last hit kept is the lowest row
            const bool hit = v[r] == mx;
            arg = hit ? n0 + orow + acc_row(r, h) : arg;
            cnt += hit ? 1 : 0;
        }
    }
    {
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
        if (ROWS == 64) {
            const float m2 = redm[128 + c];
            if (m2 > m) { m = m2; a = reda[128 + c]; k = redc[128 + c]; }
            else if (m2 == m) { k += redc[128 + c]; }
        }
        const size_t o = ((size_t)b * tiles + tile) * 128 + c;
        pmax[o] = m; parg[o] = a; pcnt[o] = k;
    }
    if (MASKS) {   // the tile's mask rows are contiguous in HBM (issuing these stores before the last chain is slower:
        const int live = n - n0 < ROWS ? n - n0 : ROWS;   // they count in vmcnt and stall the fragment ring)
        unsigned *dst = masks + ((size_t)b * n + n0) * MASK_WORDS;
        for (int e = threadIdx.x; e < live * MASK_WORDS; e += THREADS) dst[e] = mtile[e];
    }
    GA_STAMP(0, 7);
}

// ------------------------------------------------------------------------------------------
// Backward kernel over a list of rows.  grid = (row tiles, clouds).  rows: [b][rows_per_cloud]
// point indices (duplicates allowed: every listed row is written with the same value), or null
// for "all points".  A cloud takes part only if dense_flag[b] == want_dense (the sparse launch
// skips flagged clouds, the dense launch the others).  g_enc[b][row][3] = dL/d adv via the encoder.
// ------------------------------------------------------------------------------------------
template <int ROWS>
__device__ __forceinline__ void encoder_bwd_tile(const DeviceAE &A, int n, const float *adv, const int *rows,
                                                 int rows_per_cloud, const float *z, const int *zcnt, const float *dz,
                                                 float *g_enc, const int b, const int r0) {
    using LD = EncLds<ROWS>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *bufP = lds;
    float *bufQ = bufP + LD::P_FLOATS;
    float *scratch = bufQ + LD::Q_FLOATS;
    float *pts = scratch + LD::SCRATCH_FLOATS;                     // [ROWS][3]
    int *rowid = reinterpret_cast<int *>(pts + ROWS * 3);          // [ROWS]
    unsigned char *m1 = reinterpret_cast<unsigned char *>(rowid + ROWS);
    unsigned char *m2 = m1 + ROWS * 64;
    unsigned char *m3 = m2 + ROWS * 128;
    unsigned char *m4 = m3 + ROWS * 128;

    if (threadIdx.x < ROWS) {
        int rr = r0 + threadIdx.x;
        int p;
        if (rows) { rr = rr < rows_per_cloud ? rr : rows_per_cloud - 1; p = rows[(size_t)b * rows_per_cloud + rr]; }
        else p = rr < n ? rr : n - 1;
        rowid[threadIdx.x] = p;
    }
    __syncthreads();
    if (threadIdx.x < ROWS * 3) {
        const int r = threadIdx.x / 3, a = threadIdx.x % 3;
        pts[threadIdx.x] = adv[((size_t)b * n + rowid[r]) * 3 + a];
    }
    __syncthreads();
    // forward recompute, bit-identical to the forward kernel (canonical accumulation order)
    fwd_layer0<ROWS, true>(pts, bufQ, 68, A, m1);
    __syncthreads();
    const bool x3 = A.enc_arith != GEOADV_ENC_ARITH_F32;             // (uniform: the arithmetic the forward used, encoder_x3.h)
    if (x3) {
        auto bn_relu = [&](int L, float *out, int s_out, unsigned char *mask, int width) {
            return [=, &A](int row, int c, float a) {
                const float v = fmaxf(fmaf(a, A.scale[L][c], A.shift[L][c]), 0.f);
                out[row * s_out + c] = v;
                mask[row * width + c] = v > 0.f;
            };
        };
        // layer 4 forward -> da4 into bufQ, as below
        auto pool_grad = [&](int row, int c, float a) {
            const float sc = A.scale[4][c];
            const float v = fmaxf(fmaf(a, sc, A.shift[4][c]), 0.f);
            const float zc = z[(size_t)b * 128 + c];
            const int kc = zcnt[(size_t)b * 128 + c];
            const float gz = (kc > 1 ? (1.0f / (float)kc) : 1.0f) * dz[(size_t)b * 128 + c];
            bufQ[row * 132 + c] = (v == zc && v > 0.f) ? gz * sc : 0.f;
        };
        if (A.enc_arith == GEOADV_ENC_ARITH_F16X2) {
            xp_layer_lds<2, 1, ROWS>(bufQ, 68, A.enc_h2, A.h2_act_scale[0], A.h2_unscale[1], bn_relu(1, bufP, 132, m2, 128));
            __syncthreads();
            xp_layer_lds<2, 2, ROWS>(bufP, 132, A.enc_h2, A.h2_act_scale[1], A.h2_unscale[2], bn_relu(2, bufQ, 132, m3, 128));
            __syncthreads();
            xp_layer_lds<2, 3, ROWS>(bufQ, 132, A.enc_h2, A.h2_act_scale[2], A.h2_unscale[3], bn_relu(3, bufP, 260, m4, 256));
            __syncthreads();
            xp_layer_lds<2, 4, ROWS>(bufP, 260, A.enc_h2, A.h2_act_scale[3], A.h2_unscale[4], pool_grad);
        } else {
            xp_layer_lds<3, 1, ROWS>(bufQ, 68, A.enc_x3, 1.f, 1.f, bn_relu(1, bufP, 132, m2, 128));
            __syncthreads();
            xp_layer_lds<3, 2, ROWS>(bufP, 132, A.enc_x3, 1.f, 1.f, bn_relu(2, bufQ, 132, m3, 128));
            __syncthreads();
            xp_layer_lds<3, 3, ROWS>(bufQ, 132, A.enc_x3, 1.f, 1.f, bn_relu(3, bufP, 260, m4, 256));
            __syncthreads();
            xp_layer_lds<3, 4, ROWS>(bufP, 260, A.enc_x3, 1.f, 1.f, pool_grad);
        }
    } else {
    fwd_layer<ROWS, 128, true>(bufQ, 68, bufP, 132, A.enc_fwd[1], A.scale[1], A.shift[1], m2, scratch);
    __syncthreads();
    fwd_layer<ROWS, 128, true>(bufP, 132, bufQ, 132, A.enc_fwd[2], A.scale[2], A.shift[2], m3, scratch);
    __syncthreads();
    fwd_layer<ROWS, 256, true>(bufQ, 132, bufP, 260, A.enc_fwd[3], A.scale[3], A.shift[3], m4, scratch);
    __syncthreads();
    }

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (!x3) {   // layer 4 forward -> da4 = dz/cnt * [h5 == z, z > 0] * scale4   into bufQ (128 wide)
        constexpr int UNITS = 4 * (ROWS / 32);
        const int col = ((wave % UNITS) % 4) * 32 + (lane & 31);
        const float sc = A.scale[4][col], sh = A.shift[4][col];
        const float zc = z[(size_t)b * 128 + col];
        const int kc = zcnt[(size_t)b * 128 + col];
        const float gz = (kc > 1 ? (1.0f / (float)kc) : 1.0f) * dz[(size_t)b * 128 + col];
        layer_gemm<ROWS, 128, KC_L4>(bufP, 260, A.enc_fwd[4], scratch, [&](int row, int c, float a) {
            const float v = fmaxf(fmaf(a, sc, sh), 0.f);
            bufQ[row * 132 + c] = (v == zc && v > 0.f) ? gz * sc : 0.f;
        });
    }
    __syncthreads();
    // dh4 = da4 @ W4^T (128 -> 256); da3 = dh4 * mask4 * scale3   into bufP (256 wide)
    {
        const int col = (wave % 8) * 32 + (lane & 31);
        const float sc = A.scale[3][col];
        layer_gemm<ROWS, 256, 0>(bufQ, 132, A.enc_bwd[4], scratch,
                                 [&](int row, int c, float a) { bufP[row * 260 + c] = m4[row * 256 + c] ? a * sc : 0.f; });
    }
    __syncthreads();
    // dh3 = da3 @ W3^T (256 -> 128); da2 = dh3 * mask3 * scale2   into bufQ
    {
        constexpr int UNITS = 4 * (ROWS / 32);
        const int col = ((wave % UNITS) % 4) * 32 + (lane & 31);
        const float sc = A.scale[2][col];
        layer_gemm<ROWS, 128, 0>(bufP, 260, A.enc_bwd[3], scratch,
                                 [&](int row, int c, float a) { bufQ[row * 132 + c] = m3[row * 128 + c] ? a * sc : 0.f; });
    }
    __syncthreads();
    // dh2 = da2 @ W2^T (128 -> 128); da1 = dh2 * mask2 * scale1   into bufP (stride 132)
    {
        constexpr int UNITS = 4 * (ROWS / 32);
        const int col = ((wave % UNITS) % 4) * 32 + (lane & 31);
        const float sc = A.scale[1][col];
        layer_gemm<ROWS, 128, 0>(bufQ, 132, A.enc_bwd[2], scratch,
                                 [&](int row, int c, float a) { bufP[row * 132 + c] = m2[row * 128 + c] ? a * sc : 0.f; });
    }
    __syncthreads();
    // dh1 = da1 @ W1^T (128 -> 64); da0 = dh1 * mask1 * scale0   into bufQ (stride 68)
    {
        constexpr int UNITS = 2 * (ROWS / 32);
        const int col = ((wave % UNITS) % 2) * 32 + (lane & 31);
        const float sc = A.scale[0][col];
        layer_gemm<ROWS, 64, 0>(bufP, 132, A.enc_bwd[1], scratch,
                                [&](int row, int c, float a) { bufQ[row * 68 + c] = m1[row * 64 + c] ? a * sc : 0.f; });
    }
    __syncthreads();
    // dh0 = da0 @ W0^T (64 -> 3) on the VALU
    if (threadIdx.x < ROWS * 3) {
        const int r = threadIdx.x / 3, a = threadIdx.x % 3;
        float s = 0.f;
#pragma unroll 8
        for (int c = 0; c < 64; ++c) s = fmaf(bufQ[r * 68 + c], A.w0[a * 64 + c], s);
        const bool live = rows ? true : (r0 + r < n);
        if (live) g_enc[((size_t)b * n + rowid[r]) * 3 + a] = s;
    }
}


// Sparse backward from the forward's ReLU masks: no recompute.  A workgroup handles 16 of the cloud's 128 critical rows
// (slot c = the arg-max row of channel c; a row listed several times is computed several times with the same result) on
// the 16x16x4 MFMA shape (mfma_tile.h: layer_gemm16): 8 workgroups per cloud -- every CU at B = 32, where 32-row tiles
// left half of them idle -- and half the matrix time per workgroup.  da4[slot][c] = dz[c] * scale4[c] where that slot's
// row IS the arg-max of channel c and z > 0 (clouds with a tied maximum are flagged and go through the dense, recomputing
// path instead).  The products are the recomputing path's, summed in the 16x16x4 pipe's order instead of the 32x32x2
// one's: the two paths agree to rounding, not bit for bit.
constexpr int BWM_ROWS = 16;
constexpr int BWM_SCALES = 64 + 128 + 128 + 256;       // scale0 .. scale3, staged once
constexpr size_t BWM_LDS_BYTES = sizeof(float) * (BWM_ROWS * 260 + BWM_ROWS * 132 + BWM_SCALES + 128) +
                                 sizeof(int) * (BWM_ROWS + 128) + sizeof(unsigned) * BWM_ROWS * MASK_WORDS;

__device__ __forceinline__ void encoder_bwd_masked_body(const DeviceAE &A, int n, const unsigned *masks, const int *rows,
                                                        const float *z, const float *dz, const int *dense_flag,
                                                        float *g_enc, const int bx, const int b) {
    if (dense_flag[b] != 0) return;
    constexpr int ROWS = BWM_ROWS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *bufP = lds;                                  // [16][260]
    float *bufQ = bufP + ROWS * 260;                    // [16][132]
    float *sc0 = bufQ + ROWS * 132;                     // BN scales of layers 0..3
    float *sc1 = sc0 + 64, *sc2 = sc1 + 128, *sc3 = sc2 + 128;
    float *dzs = sc0 + BWM_SCALES;                      // [128] dz * scale4 where z > 0
    int *rowid = reinterpret_cast<int *>(dzs + 128);    // [16]
    int *crit = rowid + ROWS;                           // [128]
    unsigned *mw = reinterpret_cast<unsigned *>(crit + 128);   // [16][MASK_WORDS]
    const int r0 = bx * ROWS;
    // weights are requested one layer ahead of their use (mfma_tile.h: Frag16): the first two before anything else
    Frag16<128, 256> f4;
    Frag16<256, 128> f3;
    Frag16<128, 128> f2;
    Frag16<128, 64> f1;
    frag16_load(f4, A.enc_bwd16[4]);
    if (threadIdx.x < 128) {
        const int c = threadIdx.x;
        crit[c] = rows[(size_t)b * 128 + c];
        dzs[c] = z[(size_t)b * 128 + c] > 0.f ? dz[(size_t)b * 128 + c] * A.scale[4][c] : 0.f;
        sc1[c] = A.scale[1][c]; sc2[c] = A.scale[2][c];
        if (c < 64) sc0[c] = A.scale[0][c];
    } else if (threadIdx.x < 384) {
        sc3[threadIdx.x - 128] = A.scale[3][threadIdx.x - 128];
    }
    __syncthreads();
    GA_STAMP(1, 1);
    if (threadIdx.x < ROWS) rowid[threadIdx.x] = crit[r0 + threadIdx.x];
    __syncthreads();
    for (int e = threadIdx.x; e < ROWS * MASK_WORDS; e += ENC_THREADS)
        mw[e] = masks[((size_t)b * n + rowid[e / MASK_WORDS]) * MASK_WORDS + e % MASK_WORDS];
    for (int e = threadIdx.x; e < ROWS * 128; e += ENC_THREADS) {
        const int s = e >> 7, c = e & 127;
        bufQ[s * 132 + c] = crit[c] == rowid[s] ? dzs[c] : 0.f;
    }
    __syncthreads();
    GA_STAMP(1, 2);
    auto bit = [&](int row, int off, int c) { return (mw[row * MASK_WORDS + off + (c >> 5)] >> (c & 31)) & 1u; };
    // dh4 = da4 @ W4^T (128 -> 256); da3 = dh4 * mask4 * scale3   into bufP
    frag16_load(f3, A.enc_bwd16[3]);
    layer_gemm16(bufQ, 132, f4, [&](int row, int c, float a) { bufP[row * 260 + c] = bit(row, MASK_OFF4, c) ? a * sc3[c] : 0.f; });
    __syncthreads();
    GA_STAMP(1, 3);
    // dh3 = da3 @ W3^T (256 -> 128); da2 = dh3 * mask3 * scale2   into bufQ
    frag16_load(f2, A.enc_bwd16[2]);
    layer_gemm16(bufP, 260, f3, [&](int row, int c, float a) { bufQ[row * 132 + c] = bit(row, MASK_OFF3, c) ? a * sc2[c] : 0.f; });
    __syncthreads();
    GA_STAMP(1, 4);
    // dh2 = da2 @ W2^T (128 -> 128); da1 = dh2 * mask2 * scale1   into bufP (stride 132)
    frag16_load(f1, A.enc_bwd16[1]);
    layer_gemm16(bufQ, 132, f2, [&](int row, int c, float a) { bufP[row * 132 + c] = bit(row, MASK_OFF2, c) ? a * sc1[c] : 0.f; });
    __syncthreads();
    GA_STAMP(1, 5);
    // dh1 = da1 @ W1^T (128 -> 64); da0 = dh1 * mask1 * scale0   into bufQ (stride 68)
    layer_gemm16(bufP, 132, f1, [&](int row, int c, float a) { bufQ[row * 68 + c] = bit(row, 0, c) ? a * sc0[c] : 0.f; });
    __syncthreads();
    GA_STAMP(1, 6);
    if (threadIdx.x < ROWS * 3) {   // dh0 = da0 @ W0^T (64 -> 3) on the VALU
        const int r = threadIdx.x / 3, a = threadIdx.x % 3;
        float s = 0.f;
#pragma unroll 8
        for (int c = 0; c < 64; ++c) s = fmaf(bufQ[r * 68 + c], A.w0[a * 64 + c], s);
        g_enc[((size_t)b * n + rowid[r]) * 3 + a] = s;
    }
}

// Sparse launch: grid (128 / ROWS, batch): block (tile, b) handles 32 of cloud b's 128 critical rows; flagged clouds
// (exact tie in the max-pool) are skipped.  Dense launch: grid (n / ROWS, DENSE_SLOTS): the flagged clouds -- almost
// never any -- are dealt round-robin to the DENSE_SLOTS block rows, which process every point of them; with no flagged
// cloud the 2 x n/64 blocks exit after reading `batch` flags (< 1 us instead of ~5 us for a full (n/64, batch) grid).
template <int ROWS, bool want_dense>
__global__ __launch_bounds__(ENC_THREADS) void encoder_bwd_kernel(DeviceAE A, int n, int batch, const float *adv,
                                                                  const int *rows, int rows_per_cloud,
                                                                  const float *z, const int *zcnt, const float *dz,
                                                                  const int *dense_flag, float *g_enc) {
    if (!want_dense) {
        const int b = blockIdx.y;
        if (dense_flag[b] != 0) return;
        encoder_bwd_tile<ROWS>(A, n, adv, rows, rows_per_cloud, z, zcnt, dz, g_enc, b, blockIdx.x * ROWS);
        return;
    }
    __shared__ int any_flag;                           // fast path: one parallel look at the flags, usually all zero
    if (threadIdx.x == 0) any_flag = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < batch; b += ENC_THREADS)
        if (dense_flag[b] != 0) atomicOr(&any_flag, 1);
    __syncthreads();
    if (!any_flag) return;
    int rank = 0;
    for (int b = 0; b < batch; ++b) {
        if (dense_flag[b] == 0) continue;
        if ((rank++ % (int)gridDim.y) != (int)blockIdx.y) continue;
        encoder_bwd_tile<ROWS>(A, n, adv, nullptr, 0, z, zcnt, dz, g_enc, b, blockIdx.x * ROWS);
        __syncthreads();                               // LDS is reused for the next flagged cloud
    }
}

// Both backward launches in one: blocks [0, 8 * batch) (or 4 * batch: 32-row form) = masked sparse path, the rest = the dense path for clouds with a
// tied maximum (its blocks read the flags and leave at once when there is none -- which is almost always).  A launch of
// its own for that check costs ~4 us per iteration.
constexpr int BWD_MERGED_DENSE_SLOTS = 2;
__global__ __launch_bounds__(ENC_THREADS) void encoder_bwd_merged_kernel(DeviceAE A, int n, int batch, const unsigned *masks,
                                                                        const int *rows, const float *z, const int *zcnt,
                                                                        const float *dz, const int *dense_flag, const float *adv,
                                                                        float *g_enc) {
    const int per = 128 / BWM_ROWS;
    const int nm = per * batch;
    GA_STAMP(1, 0);
    if ((int)blockIdx.x < nm) {
        encoder_bwd_masked_body(A, n, masks, rows, z, dz, dense_flag, g_enc, blockIdx.x % per, blockIdx.x / per);
        GA_STAMP(1, 7);
        return;
    }
    const int d = blockIdx.x - nm, tiles = (n + 63) / 64, tile = d % tiles, slot = d / tiles;
    __shared__ int any_flag;
    if (threadIdx.x == 0) any_flag = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < batch; b += ENC_THREADS)
        if (dense_flag[b] != 0) atomicOr(&any_flag, 1);
    __syncthreads();
    if (!any_flag) return;
    int rank = 0;
    for (int b = 0; b < batch; ++b) {
        if (dense_flag[b] == 0) continue;
        if ((rank++ % BWD_MERGED_DENSE_SLOTS) != slot) continue;
        encoder_bwd_tile<64>(A, n, adv, nullptr, 0, z, zcnt, dz, g_enc, b, tile * 64);
        __syncthreads();
    }
}

constexpr int BWD_DENSE_SLOTS = 2;
constexpr int BWD_SPARSE_ROWS = 32;
constexpr int BWD_DENSE_ROWS = 64;

static int set_lds_attr_once() {
    static DeviceOnce once;
    return once.run([]() -> int {
        GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_fwd2_kernel<false, 64>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)Fwd2<64>::LDS_BYTES));
        GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_fwd2_kernel<true, 64>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)Fwd2<64>::LDS_BYTES_MASKS));
        GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_bwd_merged_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)EncLds<64>::bwd_bytes));
        GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_bwd_kernel<BWD_SPARSE_ROWS, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)EncLds<BWD_SPARSE_ROWS>::bwd_bytes));
        GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(encoder_bwd_kernel<BWD_DENSE_ROWS, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)EncLds<BWD_DENSE_ROWS>::bwd_bytes));
        return GEOADV_OK;
    });
}

// Tile height of the forward for a batch of b clouds: 32 rows when 64-row tiles would not even give every CU one tile.
#ifndef ENC_ROWS32_BELOW
#define ENC_ROWS32_BELOW 2      // 32-row tiles while 64-row ones give a CU fewer than two workgroups (measured at B = 12: 0.1235 -> 0.1135 ms per iteration; B = 8, 16: unchanged)
#endif
int encoder_fwd_rows(int b, int n) { return (long)b * cdiv(n, 64) < ENC_ROWS32_BELOW * kCUs ? 32 : 64; }
// pool partials per cloud of a forward launch
int encoder_tiles(const DeviceAE &A, int b) {
    return A.enc_arith != GEOADV_ENC_ARITH_F32 ? cdiv(A.n_points, encoder_x3_points(b, A.n_points)) : cdiv(A.n_points, encoder_fwd_rows(b, A.n_points));
}
int encoder_tiles_max(int n) { return cdiv(n, 32); }              // what the pool-partial buffers are sized for

// Words of ReLU mask per point the forward leaves for the sparse backward.
int encoder_mask_words() { return MASK_WORDS; }

template <bool MASKS, int ROWS>
static void launch_fwd2(const DeviceAE &A, int b, const float *x, const float *pert, float *adv_out, float *pmax, int *parg, int *pcnt,
                        unsigned *masks, hipStream_t stream, hipEvent_t start, hipEvent_t stop, const FusedAdam &fa) {
    const dim3 grid(cdiv(A.n_points, ROWS), b), block(Fwd2<ROWS>::THREADS);
    const unsigned lds = (unsigned)(MASKS ? Fwd2<ROWS>::LDS_BYTES_MASKS : Fwd2<ROWS>::LDS_BYTES);
    if (start && stop)
        hipExtLaunchKernelGGL((encoder_fwd2_kernel<MASKS, ROWS>), grid, block, lds, stream, start, stop, 0,
                              A, A.n_points, x, pert, adv_out, pmax, parg, pcnt, masks, fa);
    else
        encoder_fwd2_kernel<MASKS, ROWS><<<grid, block, lds, stream>>>(A, A.n_points, x, pert, adv_out, pmax, parg, pcnt, masks, fa);
}

// pmax/parg/pcnt: [b][encoder_tiles(A, b)][128]; masks: [b][n][MASK_WORDS] or null (plain forward: geoadv_ae_forward, recomputing backward)
// start / stop (optional): events that receive the kernel's own begin / end time stamps (geoadv_attack_profile).
// fused (optional): a pending Adam step on pert, applied by the point loaders before they form adv = x + pert (needs pert
// and adv_out)
int launch_encoder_fwd(const DeviceAE &A, int b, const float *x, const float *pert, float *adv_out, float *pmax,
                       int *parg, int *pcnt, unsigned *masks, hipStream_t stream, hipEvent_t start, hipEvent_t stop,
                       const FusedAdam *fused) {
    if (int st = set_lds_attr_once()) return st;
    if (b <= 0) return GEOADV_OK;
    FusedAdam fa{};
    if (fused) {
        GA_REQUIRE(pert && adv_out && fused->m && fused->pert == pert, "encoder_fwd: the fused Adam step needs pert and adv_out");
        fa = *fused;
    }
    if (A.enc_arith != GEOADV_ENC_ARITH_F32) return launch_encoder_fwd_x3(A, b, x, pert, adv_out, pmax, parg, pcnt, masks, stream, start, stop, fa);
    const bool small = encoder_fwd_rows(b, A.n_points) == 32;
    if (masks) {
        if (small) launch_fwd2<true, 32>(A, b, x, pert, adv_out, pmax, parg, pcnt, masks, stream, start, stop, fa);
        else launch_fwd2<true, 64>(A, b, x, pert, adv_out, pmax, parg, pcnt, masks, stream, start, stop, fa);
    } else {
        if (small) launch_fwd2<false, 32>(A, b, x, pert, adv_out, pmax, parg, pcnt, nullptr, stream, start, stop, fa);
        else launch_fwd2<false, 64>(A, b, x, pert, adv_out, pmax, parg, pcnt, nullptr, stream, start, stop, fa);
    }
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

// The pool Jacobian (encoder_jac.h) as a launch of its own: grid = 8 workgroups per cloud.
__global__ __launch_bounds__(ENC_THREADS) void encoder_jac_kernel(DeviceAE A, JacArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    GA_STAMP(2, 0);
    encoder_jac_block<true>(A, a, lds, blockIdx.x % (128 / JAC_ROWS), blockIdx.x / (128 / JAC_ROWS));
    GA_STAMP(2, 7);
}
int launch_encoder_jac(const DeviceAE &A, int b, const JacArgs &a, hipStream_t stream) {
    if (b <= 0) return GEOADV_OK;
    encoder_jac_kernel<<<b * (128 / JAC_ROWS), ENC_THREADS, JAC_LDS_BYTES, stream>>>(A, a);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}
// The decoder backward's tail (dd2 -> dd1 -> dz + the pool Jacobian's apply, decoder_tail.h) and the dense recomputing
// backward for clouds with a tied pool maximum in ONE launch: blocks [0, batch) = one cloud's tail each (1024 threads);
// blocks [batch, ...) = 32-row dense tiles (their upper eight waves leave at once) dealt to DENSE_SLOTS block rows.  A dense
// block needs dz of its cloud, which a tail block of the SAME launch produces: it waits for that block's flag -- only where a
// cloud IS tied (duplicated points; almost never), otherwise it leaves after one look at the flags and the launch costs what
// the tail alone costs, where a dense launch of its own was 4.1 us + a boundary on every step.  Progress: tail blocks never
// wait; a dense block holds a whole CU's LDS, and the caller (attack.hip, do_step) only uses this launch while batch + 2 n / 32
// workgroups fit the 256 CUs at once, so every tail block is resident whatever the dispatch order -- larger clouds take two
// plain launches.  The spin is bounded all the same; a timeout sets a word that geoadv_attack_status reports as GEOADV_EHIP.
constexpr int TD_DENSE_ROWS = 32, TD_DENSE_SLOTS = 2;

__global__ __launch_bounds__(LD_THREADS) void decoder_tail_dense_kernel(DeviceAE A, TailDenseArgs a) {
    if ((int)blockIdx.x < a.batch) {
        decoder_bwd_tail_body(A, a.batch, a.chunks, a.partial, a.d1, a.d2, a.dz, a.ja, blockIdx.x, a.ready, a.epoch);
        return;
    }
    if (threadIdx.x >= ENC_THREADS) return;
    __shared__ int any_flag;                           // fast path: one parallel look at the flags, usually all zero
    if (threadIdx.x == 0) any_flag = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < a.batch; b += ENC_THREADS)
        if (a.dense_flag[b] != 0) atomicOr(&any_flag, 1);
    __syncthreads();
    if (!any_flag) return;
    const int d = blockIdx.x - a.batch, tiles = (a.n + TD_DENSE_ROWS - 1) / TD_DENSE_ROWS, tile = d % tiles, slot = d / tiles;
    int rank = 0;
    for (int b = 0; b < a.batch; ++b) {
        if (a.dense_flag[b] == 0) continue;
        if ((rank++ % TD_DENSE_SLOTS) != slot) continue;
        if (threadIdx.x == 0) {                        // ONE lane polls the cloud's flag (relaxed, agent scope), bounded
            unsigned spins = 0;
            while (__hip_atomic_load(a.ready + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
                __builtin_amdgcn_s_sleep(16);
                if (++spins > (1u << 24)) { *a.spin_timeout = 1; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // ONE acquire after the match ...
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                               // ... which the barrier holds for the other waves' loads of dz
        encoder_bwd_tile<TD_DENSE_ROWS>(A, a.n, a.adv, nullptr, 0, a.z, a.zcnt, a.dz, a.g_enc, b, tile * TD_DENSE_ROWS);
        __syncthreads();                               // LDS is reused for the next flagged cloud
    }
}

int launch_decoder_tail_dense(const DeviceAE &A, const TailDenseArgs &a, hipStream_t stream) {
    if (a.batch <= 0) return GEOADV_OK;
    static DeviceOnce attr;
    if (int rc = attr.run([]() -> int {
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(decoder_tail_dense_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)EncLds<TD_DENSE_ROWS>::bwd_bytes));
            return GEOADV_OK;
        })) return rc;
    const int grid = a.batch + cdiv(a.n, TD_DENSE_ROWS) * TD_DENSE_SLOTS;
    decoder_tail_dense_kernel<<<grid, LD_THREADS, EncLds<TD_DENSE_ROWS>::bwd_bytes, stream>>>(A, a);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

// The dense, recomputing backward alone: only clouds with a tied pool maximum do work (their gradient cannot come from the
// Jacobian); with no such cloud the 2 x n / 64 workgroups leave after one look at the flags.
int launch_encoder_bwd_dense(const DeviceAE &A, int b, const float *adv, const float *z, const int *zcnt, const float *dz,
                             const int *dense_flag, float *g_enc, hipStream_t stream) {
    if (int st = set_lds_attr_once()) return st;
    if (b <= 0) return GEOADV_OK;
    encoder_bwd_kernel<BWD_DENSE_ROWS, true><<<dim3(cdiv(A.n_points, BWD_DENSE_ROWS), BWD_DENSE_SLOTS), ENC_THREADS, EncLds<BWD_DENSE_ROWS>::bwd_bytes, stream>>>(
        A, A.n_points, b, adv, nullptr, 0, z, zcnt, dz, dense_flag, g_enc);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

// masks != null: one launch = masked sparse blocks (128 critical rows of every un-flagged cloud) + dense recomputing blocks
// that only do work for clouds with a tied pool maximum.  masks == null (cfg.recompute_backward, the A/B of
// tests/test_gpu_attack.py): the recomputing kernel for both, sparse then dense.  g_enc must be zeroed by the caller
// (rows that are not critical keep gradient 0).
int launch_encoder_bwd(const DeviceAE &A, int b, const float *adv, const int *crit_rows, const float *z,
                       const int *zcnt, const float *dz, const int *dense_flag, float *g_enc, const unsigned *masks,
                       hipStream_t stream) {
    if (int st = set_lds_attr_once()) return st;
    if (b <= 0) return GEOADV_OK;
    if (masks) {
        // 16-row workgroups (8 per cloud) at EVERY batch size: the tile shape fixes the summation order, so a cloud's gradient
        // bits do not depend on the batch it sits in (a 32-row throughput form for B > 32 was 1.5 % faster at
        // B = 256 and is gone -- large batches of the output-space attack use the pool Jacobian anyway, encoder_jac.h)
        const size_t lds = std::max(BWM_LDS_BYTES, EncLds<64>::bwd_bytes);
        const int grid = (128 / BWM_ROWS) * b + cdiv(A.n_points, 64) * BWD_MERGED_DENSE_SLOTS;
        encoder_bwd_merged_kernel<<<grid, ENC_THREADS, lds, stream>>>(A, A.n_points, b, masks, crit_rows, z, zcnt, dz, dense_flag, adv, g_enc);
        GA_LAUNCH_CHECK();
        return GEOADV_OK;
    }
    encoder_bwd_kernel<BWD_SPARSE_ROWS, false><<<dim3(128 / BWD_SPARSE_ROWS, b), ENC_THREADS, EncLds<BWD_SPARSE_ROWS>::bwd_bytes, stream>>>(
        A, A.n_points, b, adv, crit_rows, 128, z, zcnt, dz, dense_flag, g_enc);
    GA_LAUNCH_CHECK();
    encoder_bwd_kernel<BWD_DENSE_ROWS, true><<<dim3(cdiv(A.n_points, BWD_DENSE_ROWS), BWD_DENSE_SLOTS), ENC_THREADS, EncLds<BWD_DENSE_ROWS>::bwd_bytes, stream>>>(
        A, A.n_points, b, adv, nullptr, 0, z, zcnt, dz, dense_flag, g_enc);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

}  // namespace geoadv
GA_STAMPS_GETTER(geoadv_debug_stamps_encoder)
