// MFMA tile helpers shared by the encoder kernels (encoder.hip) and the training kernels (train.hip):
// fragment loads, the B-fragment register ring, and layer_gemm = [ROWS x K] LDS tile @ packed weights.
#pragma once
#include "ae.h"

namespace geoadv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ENC_THREADS = 512;                 // 8 waves

// accumulator register -> row inside a 32-row block (C/D layout of the 32x32 MFMA shapes)
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// One chain over k-groups [t0, t1) (a k-group = 8 k values = 4 MFMA k-steps); RM row blocks of
// 32 share each B fragment.  A fragments: ds_read_b128 from the LDS activation tile (row stride
// s_in = width + 4 floats keeps them bank-conflict free); B fragments: one coalesced 1 KiB
// global_load_dwordx4 per k-group from the packed weights, prefetched one group ahead.
// A ring refill; the sched_barrier after each call keeps it between the MFMA groups (a volatile load would be
// followed by vmcnt(0); without the barrier the scheduler clusters all four refills at the loop end).
__device__ __forceinline__ float4 ld_pinned(const float4 *p) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4 t = *reinterpret_cast<const v4 *>(p);
    return make_float4(t.x, t.y, t.z, t.w);
}

// B fragments come through BUFFER loads: descriptor (the layer's packed weights) and byte offset of the fragment in SGPRs,
// the lane's 16 bytes in one VGPR, 0 / 1 / 2 / 3 KiB as the instruction's immediate -- so stepping to the next fragments is
// scalar arithmetic.  (A global_load from `uniform pointer + lane` ends up with a 64-bit VGPR address and a
// v_lshl_add_u64 per step: the compiler hoists the sum.)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct FragSrc {
    __amdgpu_buffer_rsrc_t rsrc;
    int off;                                               // wave-uniform byte offset of the chain's first fragment
};
__device__ __forceinline__ FragSrc frag_src(const float *w, int first_kgroup) {
    return FragSrc{__builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w), 0, 0x7fffffff, 0x00020000), first_kgroup * 1024};
}
__device__ __forceinline__ FragSrc frag_at(const FragSrc &f, int kgroups) { return FragSrc{f.rsrc, f.off + kgroups * 1024}; }
__device__ __forceinline__ float4 ld_frag(const FragSrc &f, int kgroup, unsigned lane_bytes) {
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(f.rsrc, lane_bytes, f.off + kgroup * 1024, 0);
    return make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
}
template <int RM>
__device__ __forceinline__ void mfma_group(const float4 (&a)[RM], const float4 &b, f32x16 (&acc)[RM]) {
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) {
        acc[rm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rm].x, b.x, acc[rm], 0, 0, 0);
        acc[rm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rm].y, b.y, acc[rm], 0, 0, 0);
        acc[rm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rm].z, b.z, acc[rm], 0, 0, 0);
        acc[rm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rm].w, b.w, acc[rm], 0, 0, 0);
    }
}

// Chain lengths (t1 - t0) are multiples of 4 for every layer of the template architecture.
// Software pipeline: a wave consumes one 1 KiB B fragment per ~512 cycles (two waves share a
// SIMD's MFMA pipe) while an L2 hit takes ~800 cycles under load, so B fragments run through a
// 4-slot register ring (4 loads in flight); A fragments (LDS, ~130 cycles) are fetched one
// k-group ahead.
// cb (the column block) must be wave-uniform and held in an SGPR (layer_gemm reads the wave index with readfirstlane):
// the fragment offsets are then scalar and the refills cost no VALU instruction.
template <int RM>
__device__ __forceinline__ void gemm_chain(const float *in, int s_in, int row0, const PackedLayer &L, int cb, int t0,
                                           int t1, f32x16 (&acc)[RM]) {
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5, i = lane & 31;
    const int kg = L.K >> 3;
    const FragSrc bp = frag_src(L.w, cb * kg);
    const unsigned lb = (unsigned)lane * 16u;
    const float *ar[RM];
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) ar[rm] = in + (row0 + rm * 32 + i) * s_in + 4 * h;
    float4 b0 = ld_frag(bp, t0, lb), b1 = ld_frag(bp, t0 + 1, lb), b2 = ld_frag(bp, t0 + 2, lb), b3 = ld_frag(bp, t0 + 3, lb);
    float4 a0[RM], a1[RM];
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) a0[rm] = *reinterpret_cast<const float4 *>(ar[rm] + 8 * t0);
    for (int t = t0; t < t1; t += 4) {
        // refills are unconditional (clamped index; the last iteration re-reads its own fragments): a branch
        // around a load makes the compiler drain the ring with vmcnt(0) every four k-groups
        const int tn = t + 4 < t1 ? t + 4 : t;
#pragma unroll
        for (int rm = 0; rm < RM; ++rm) a1[rm] = *reinterpret_cast<const float4 *>(ar[rm] + 8 * (t + 1));
        __builtin_amdgcn_sched_barrier(0);           // A prefetch stays ahead of the MFMA group
        mfma_group<RM>(a0, b0, acc);
        b0 = ld_frag(bp, tn, lb);
        __builtin_amdgcn_sched_barrier(0);           // pin the refill between the MFMA groups
#pragma unroll
        for (int rm = 0; rm < RM; ++rm) a0[rm] = *reinterpret_cast<const float4 *>(ar[rm] + 8 * (t + 2));
        __builtin_amdgcn_sched_barrier(0);
        mfma_group<RM>(a1, b1, acc);
        b1 = ld_frag(bp, tn + 1, lb);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rm = 0; rm < RM; ++rm) a1[rm] = *reinterpret_cast<const float4 *>(ar[rm] + 8 * (t + 3));
        __builtin_amdgcn_sched_barrier(0);
        mfma_group<RM>(a0, b2, acc);
        b2 = ld_frag(bp, tn + 2, lb);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rm = 0; rm < RM; ++rm) a0[rm] = *reinterpret_cast<const float4 *>(ar[rm] + 8 * tn);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group<RM>(a1, b3, acc);
        b3 = ld_frag(bp, tn + 3, lb);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// B-fragment ring carried ACROSS chains: while the last four k-groups of a chain run, the freed
// slots are refilled with the first four fragments of the NEXT chain (weights do not depend on the
// activations, so the request may cross the epilogue and the barrier).  Without it every chain start
// exposes one L2 round trip (~900 cycles, 8 chains per tile = 13 % of the attack encoder's tile).
struct BRing { float4 b[4]; };
__device__ __forceinline__ void ring_fill(BRing &r, const FragSrc &f, unsigned lb) {
    r.b[0] = ld_frag(f, 0, lb); r.b[1] = ld_frag(f, 1, lb); r.b[2] = ld_frag(f, 2, lb); r.b[3] = ld_frag(f, 3, lb);
}
// NT (chain length in k-groups) is a compile-time constant -- the encoder widths are fixed (ae_create checks them) -- so
// the chain is fully unrolled: the first MFMA takes the inline constant 0 as its accumulator (no 16 x v_mov per chain),
// every A / B address is base + immediate, and no select survives.  That matters more than it looks: plain VALU
// instructions do NOT overlap with the matrix pipe on this part (tools/mfma_probe.py: 4 MFMA + 16 v_add_f32 per group
// runs 14 % slower than the MFMAs alone), so every VALU instruction of such a kernel is paid in MFMA time.
// (encoder.hip keeps its own one-row-block form, chain_ring; this is the same chain for the training kernels:)
// RM row blocks share each B fragment (ar[rm] = the lane's row of block rm); HAS_NEXT = false refills the ring with the
// chain's OWN first fragments -- what a persistent workgroup that runs the same chain on tile after tile wants.
template <int NT, bool HAS_NEXT, int RM>
__device__ __forceinline__ void chain_ring_rm(const float *const (&ar)[RM], int at0, const FragSrc &cur, unsigned lb, BRing &ring,
                                           const FragSrc &next, f32x16 (&acc)[RM]) {
    static_assert(NT % 4 == 0, "chain lengths are multiples of four k-groups");
    float4 a0[RM], a1[RM];
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) a0[rm] = *reinterpret_cast<const float4 *>(ar[rm] + 8 * at0);
#pragma unroll
    for (int t = 0; t < NT; t += 4) {
        // every refill is UNCONDITIONAL (an always valid address: the chain's own first fragments if nothing follows):
        // with a branch around a load the compiler can no longer count outstanding loads and degrades the
        // s_waitcnt vmcnt(3) below to vmcnt(2)/(1)/(0)
        const bool more = t + 4 < NT;
        const FragSrc &src = more ? cur : (HAS_NEXT ? next : cur);
        const int g = more ? t + 4 : 0;
#pragma unroll
        for (int rm = 0; rm < RM; ++rm) a1[rm] = *reinterpret_cast<const float4 *>(ar[rm] + 8 * (at0 + t + 1));
        __builtin_amdgcn_sched_barrier(0);       // the next A fragment is requested BEFORE this group's MFMAs issue
        mfma_group<RM>(a0, ring.b[0], acc);
        ring.b[0] = ld_frag(src, g + 0, lb);
        __builtin_amdgcn_sched_barrier(0);       // keep the refill HERE (the scheduler would sink all four to the loop end)
#pragma unroll
        for (int rm = 0; rm < RM; ++rm) a0[rm] = *reinterpret_cast<const float4 *>(ar[rm] + 8 * (at0 + t + 2));
        __builtin_amdgcn_sched_barrier(0);
        mfma_group<RM>(a1, ring.b[1], acc);
        ring.b[1] = ld_frag(src, g + 1, lb);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rm = 0; rm < RM; ++rm) a1[rm] = *reinterpret_cast<const float4 *>(ar[rm] + 8 * (at0 + t + 3));
        __builtin_amdgcn_sched_barrier(0);
        mfma_group<RM>(a0, ring.b[2], acc);
        ring.b[2] = ld_frag(src, g + 2, lb);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rm = 0; rm < RM; ++rm) a0[rm] = *reinterpret_cast<const float4 *>(ar[rm] + 8 * (at0 + (more ? t + 4 : t)));
        __builtin_amdgcn_sched_barrier(0);
        mfma_group<RM>(a1, ring.b[3], acc);
        ring.b[3] = ld_frag(src, g + 3, lb);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// out tile [ROWS][NOUT] = in tile [ROWS][K] @ W, handed element-wise to epi(row, col, value).
// KC = number of canonical K parts (independent chains summed part0 + part1 + ...); KC == 0
// picks whatever keeps all 8 waves busy.  Must be called by every wave of the workgroup.
template <int ROWS, int NOUT, int KC_REQ, class Epi>
__device__ __forceinline__ void layer_gemm(const float *in, int s_in, const PackedLayer &L, float *scratch, Epi epi) {
    constexpr int CB = NOUT / 32, RB = ROWS / 32;
    constexpr bool RM2 = (CB * RB > 8);                      // 64 rows x 256 columns: two row blocks per wave
    constexpr int UNITS = RM2 ? CB : CB * RB;                // (column block, row block) units handed to waves
    constexpr int SPARE = 8 / UNITS;                         // waves available per unit
    constexpr int KC = KC_REQ > 0 ? KC_REQ : SPARE;
    constexpr int KS = (SPARE >= KC) ? KC : 1;               // K parts computed by different waves, or all by one
    static_assert(KS == KC || KS == 1, "bad K split");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;     // scalar: see gemm_chain
    const int h = lane >> 5, i = lane & 31;
    const int unit = wave % UNITS, ks = wave / UNITS;
    const int cb = unit % CB, rb = RM2 ? 0 : unit / CB;
    const int kg = L.K >> 3;
    constexpr int RM = RM2 ? 2 : 1;
    f32x16 acc[RM] = {};
    if (KS == KC) {
        if (ks < KS) gemm_chain<RM>(in, s_in, rb * 32, L, cb, ks * kg / KC, (ks + 1) * kg / KC, acc);
        if (KC > 1) {
            if (ks > 0 && ks < KS) {
#pragma unroll
                for (int r = 0; r < 16; ++r) scratch[(((ks - 1) * UNITS + unit) * 16 + r) * 64 + lane] = acc[0][r];
            }
            __syncthreads();
            if (ks == 0) {
#pragma unroll
                for (int p = 1; p < KC; ++p)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][r] += scratch[(((p - 1) * UNITS + unit) * 16 + r) * 64 + lane];
            }
        }
    } else if (ks == 0) {   // one wave walks the canonical parts one after the other
        gemm_chain<RM>(in, s_in, rb * 32, L, cb, 0, kg / KC, acc);
#pragma unroll
        for (int p = 1; p < KC; ++p) {
            f32x16 part[RM] = {};
            gemm_chain<RM>(in, s_in, rb * 32, L, cb, p * kg / KC, (p + 1) * kg / KC, part);
#pragma unroll
            for (int rm = 0; rm < RM; ++rm) acc[rm] += part[rm];
        }
    }
    if (ks == 0) {
        const int col = cb * 32 + i;
#pragma unroll
        for (int rm = 0; rm < RM; ++rm)
#pragma unroll
            for (int r = 0; r < 16; ++r) epi((rb + rm) * 32 + acc_row(r, h), col, acc[rm][r]);
    }
}

// ------------------------------------------------------------------------------------------
// 16-row tiles on v_mfma_f32_16x16x4_f32 (PackedLayer packed16, ae.h): out tile [16][NOUT] = in tile [16][K] @ W.
// For work that has only a few rows per cloud -- the masked encoder backward: 128 critical rows -- 16-row tiles give twice
// the workgroups of 32-row ones, i.e. every CU instead of half of them at B = 32, and half the MFMA time per workgroup.
// A operand of k-step u of k-group t: in[lane & 15][16 t + 4 (lane >> 4) + u] (one ds_read_b128 per k-group); B: the
// lane's 16 bytes of the packed fragment; C: rows 4 (lane >> 4) + r, column lane & 15.  NOUT / 16 column tiles are dealt
// to the 8 waves, CN = NOUT / 128 per wave (sharing the A fragments) -- or one each to the first NOUT / 16 waves.  K and
// NOUT are compile-time: the chain is fully unrolled and all of a wave's fragment loads of a layer are in flight at once
// (at most 16 fragments = 64 VGPRs).  Must be called by every wave of the workgroup; no barrier inside.
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A wave's B fragments of one layer.  Loading them is a step of its own (frag16_load) so that the caller can request the
// NEXT layer's weights before it multiplies the current one: weights do not depend on activations, and a layer of this
// size is one L2 round trip of latency against ~1 us of MFMA time.
template <int K, int NOUT> struct Frag16 {
    static constexpr int TILES = NOUT / 16, CN = TILES >= 8 ? TILES / 8 : 1, KG = K / 16;
    float4 b[CN][KG];
};
template <int K, int NOUT>
__device__ __forceinline__ void frag16_load(Frag16<K, NOUT> &f, const PackedLayer &L) {
    using F = Frag16<K, NOUT>;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave * F::CN >= F::TILES) return;
    const unsigned lb = (unsigned)(threadIdx.x & 63) * 16u;
#pragma unroll
    for (int cn = 0; cn < F::CN; ++cn) {
        const FragSrc bp = frag_src(L.w, (wave * F::CN + cn) * F::KG);
#pragma unroll
        for (int t = 0; t < F::KG; ++t) f.b[cn][t] = ld_frag(bp, t, lb);
    }
}
template <int K, int NOUT, class Epi>
__device__ __forceinline__ void layer_gemm16(const float *in, int s_in, const Frag16<K, NOUT> &f, Epi epi) {
    using F = Frag16<K, NOUT>;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wave * F::CN >= F::TILES) return;
    const int q = lane >> 4, m = lane & 15;
    const float *ar = in + m * s_in + 4 * q;
    f32x4 acc[F::CN] = {};
#pragma unroll
    for (int t = 0; t < F::KG; ++t) {
        const float4 a = *reinterpret_cast<const float4 *>(ar + 16 * t);
#pragma unroll
        for (int cn = 0; cn < F::CN; ++cn) {
            acc[cn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, f.b[cn][t].x, acc[cn], 0, 0, 0);
            acc[cn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, f.b[cn][t].y, acc[cn], 0, 0, 0);
            acc[cn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, f.b[cn][t].z, acc[cn], 0, 0, 0);
            acc[cn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, f.b[cn][t].w, acc[cn], 0, 0, 0);
        }
    }
#pragma unroll
    for (int cn = 0; cn < F::CN; ++cn)
#pragma unroll
        for (int r = 0; r < 4; ++r) epi(4 * q + r, (wave * F::CN + cn) * 16 + m, acc[cn][r]);
}

// The output column layer_gemm<ROWS, NOUT, *> hands to the calling lane's epilogue (constant per lane).
template <int ROWS, int NOUT>
__device__ __forceinline__ int layer_gemm_lane_col() {
    constexpr int CB = NOUT / 32, RB = ROWS / 32;
    constexpr int UNITS = (CB * RB > 8) ? CB : CB * RB;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    return ((wave % UNITS) % CB) * 32 + (threadIdx.x & 31);
}

}  // namespace geoadv
