// Symmetric Chamfer scan, SCREENED ON THE MATRIX PIPE (round 6).  Same contract as chamfer_sym_kernel (chamfer_sym.hip): every
// output is the reference's bits (tf_nndistance.cpp:21-43: d = ((dx*dx)+(dy*dy))+(dz*dz), strict '<', lowest index wins) -- but
// the 8 separately rounded VALU instructions per pair are no longer spent on every pair.
//
// Idea.  |p - q|^2 = |p|^2 + |q|^2 - 2 p.q is a K = 16 inner product once every fp32 operand is carried as two fp16 pieces
// (11 + 11 bits; products (1,1), (1,2), (2,1) per coordinate = 9 slots, the two norms as two pieces each = 4 slots, a bias and
// two padding slots): ONE v_mfma_f32_32x32x16_f16 evaluates 32 x 32 approximate distances a[i][j] with
//     | a[i][j] - s^2 * d_ref[i][j] |  <=  eps          for every pair of the workgroup,
// where s is the workgroup's power-of-two scale and eps a RIGOROUS bound computed from the workgroup's own data (below).  The
// approximations only SCREEN: minima of a over chunks of pairs are kept (one min3 per two pairs and direction instead of ten
// instructions a pair), and a chunk can hold the exact minimum (or a tie for it) only if its approximate minimum is within
// 2 eps of the smallest one.  Almost always that leaves ONE chunk (8 columns of a row, 16 rows of a column), which is then
// evaluated with the reference's arithmetic -- the reported distance and index are exact evaluations, never approximations; when
// several chunks qualify (exact ties, near ties: a few per cent of the queries on unit-cube clouds) all of them are evaluated.
// Degenerate data (non-finite or absurdly scaled coordinates, a cloud collapsed to a point) make every chunk qualify: the
// kernel then IS the exact scan, slowly.
//
// Error bound (scaled units: u = s (p - c), v = s (q - c), c = the centre of the rows' bounding box, s = 2^k with the largest
// |coordinate| in [2^13, 2^14); U, V = largest |u|, |v|; T = 2 (U^2 + V^2) >= (U + V)^2):
//   reference rounding of d_ref (5 roundings a coordinate) ................ 6 * 2^-24 T
//   centring in fp32 (u, v carry a relative 2^-24 each) .................... 2^-22 T
//   norms in fp32 + their two fp16 pieces .................................. 3 * 2^-24 T + 2^-22 T
//   coordinate pieces: dropped (2,2) products and the remainders ........... 1.5 * 2^-22 T
//   the MFMA's 16 additions, each rounded OR truncated to fp32 ............. 16 * 2^-23 * 1.01 T
//   fp16 subnormals flushed by the matrix pipe (values below 2^-14) ........ < 32 absolute
// sum < 0.87 * 2^-18 T + 32  =>  eps = 2^-18 T + 32.  (profiles/r05_bf16x3_probe.jsonl measured 5-6 units of 2^-24 sum|a||b| for
// such chains; the bound assumes nothing about the pipe's internal order.)  A bias 2^j >= 2 eps rides in a spare K slot so that
// every a is a POSITIVE float: positive floats order like their bit patterns, and all minima / medians below are integer ones
// (no canonicalisation of values that come out of the matrix pipe or LDS), with a chunk number in the low mantissa bits.
//
// Layout.  As chamfer_sym_kernel: a workgroup = 8 waves = a slice of C = 32 NCT columns (S stages of them) against 2048 rows;
// a wave owns 256 rows = 8 row tiles.  Both operands live in REGISTERS for the whole scan (A: 8 tiles x 4 VGPRs, built once;
// B: NCT tiles x 4 VGPRs per stage, built by one thread per column and exchanged through LDS), the 64 tile products of a stage
// are unrolled.  Per product (32 x 32 pairs, 16 per lane): rows -- element-wise running minimum over the column tiles (a lane's
// accumulator register r is row 8 (r / 4) + 4 (lane / 32) + r % 4, its lane is the column: the chunk of a row is "the columns
// 32 ct + lane % 32"), one min3 per two products; columns -- minimum of the lane's 16 registers (8 min3) keyed with the chunk
// number and folded into a running (smallest, second smallest) pair by one median and one minimum.  After a row tile's last
// column tile the 16 x 64 running minima are transposed through LDS (as the unscreened kernel does every 16 columns); two lanes
// per row find the two smallest of its 32 chunk minima, certify, and evaluate the winning chunk exactly.  Columns: the waves'
// pairs meet in LDS after the last row tile; two threads per column certify and evaluate 16 rows.
#pragma once
#include "common.h"
#include <limits.h>
#include <math.h>

#pragma clang fp contract(off)

namespace geoadv {

typedef _Float16 mx_f16x8 __attribute__((ext_vector_type(8)));
typedef float mx_f32x16 __attribute__((ext_vector_type(16)));

constexpr int MX_THREADS = 512;
constexpr int MX_WAVES = 8;
constexpr int MX_RT = 8;                      // row tiles of 32 per wave
constexpr int MX_WROWS = 32 * MX_RT;          // 256 rows per wave
constexpr int MX_ROWS = MX_WAVES * MX_WROWS;  // 2048 rows per workgroup (row super-tile)
constexpr int MX_CMAX = 256;                  // columns per stage at most
constexpr int MX_MAX_STAGES = 8;
constexpr int MX_TSTRIDE = 68;                // dwords per accumulator register in the transpose buffer (64 lanes + pad)
constexpr int MX_BSTRIDE = 12;                // dwords per column in the B exchange buffer (8 used: 48-byte stride, conflict-free b128 reads)
constexpr unsigned MX_ROW_IDMASK = 31u;       // row direction: 32 chunks (the lanes of a half wave)
constexpr unsigned MX_COL_IDMASK = 127u;      // column direction: 8 waves x 8 row tiles x 2 lane halves
constexpr unsigned MX_KEY_MAX = 0x7f7fffffu;  // FLT_MAX: above every finite approximation

// LDS (dwords): columns of all stages as float4 | B exchange | rows as float4 | transpose buffers (later the column keys) | row results | scratch
constexpr size_t MX_OFF_COLS = 0;
constexpr size_t MX_OFF_B = MX_OFF_COLS + 4 * MX_CMAX * MX_MAX_STAGES;
constexpr size_t MX_OFF_ROWS = MX_OFF_B + (size_t)MX_CMAX * MX_BSTRIDE;
constexpr size_t MX_OFF_T = MX_OFF_ROWS + 4 * (size_t)MX_ROWS;
constexpr size_t MX_OFF_BEST = MX_OFF_T + (size_t)MX_WAVES * 16 * MX_TSTRIDE;
constexpr size_t MX_OFF_RED = MX_OFF_BEST + 2 * (size_t)MX_ROWS;
constexpr int MX_ICAP = 160;                  // uncertified rows a wave can list (more: every row of the wave is redone at the end)
constexpr int MX_CCAP = 32;                   // ... and deferred columns (more: answered on the spot)
constexpr size_t MX_OFF_ITEMS = MX_OFF_RED + 192;
constexpr size_t MX_LDS_DWORDS = MX_OFF_ITEMS + (size_t)MX_WAVES * (MX_ICAP + MX_CCAP);
constexpr size_t MX_LDS_BYTES = 4 * MX_LDS_DWORDS;
static_assert(MX_OFF_B % 4 == 0 && MX_OFF_ROWS % 4 == 0 && MX_OFF_T % 4 == 0, "16-byte aligned regions");
static_assert(2 * MX_CMAX <= 16 * MX_TSTRIDE, "a wave's column keys fit its transpose buffer");
static_assert((MX_CMAX * MX_MAX_STAGES) % MX_THREADS == 0, "whole columns per thread in the prologue");
static_assert(MX_LDS_BYTES + 2048 <= 160 * 1024, "one workgroup's dynamic LDS (+ the host kernel's static words) fits a CU");

// (plain C, not inline assembly: these read MFMA results, and the wait states between a matrix instruction and a VALU read of
// its result are inserted by the compiler only for instructions it knows -- the first build, with v_min3_u32 from an asm
// statement, read stale accumulators a few times per thousand queries)
__device__ __forceinline__ unsigned mx_min3(unsigned a, unsigned b, unsigned c) { return min(min(a, b), c); }
__device__ __forceinline__ unsigned mx_med3(unsigned a, unsigned b, unsigned c) {       // (its operands are VALU results)
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned mx_dpp_xor1(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float mx_dpp_xor1f(float v) { return __uint_as_float(mx_dpp_xor1(__float_as_uint(v))); }

__device__ __forceinline__ float mx_sqdist(float tx, float ty, float tz, float qx, float qy, float qz) {
    const float dx = tx - qx, dy = ty - qy, dz = tz - qz;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    return (xx + yy) + zz;
}

// minimum / maximum over the 64 lanes (NaNs skipped), every lane gets the result
__device__ __forceinline__ float mx_wave_min(float v) {
#define MX_DPP(CTRL) v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false)))
    MX_DPP(0xB1); MX_DPP(0x4E); MX_DPP(0x141); MX_DPP(0x140);
#undef MX_DPP
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fminf(fminf(r0, r1), fminf(r2, r3));
}
__device__ __forceinline__ float mx_wave_max(float v) { return -mx_wave_min(-v); }

// two fp16 pieces of an fp32: x = hi + lo + (at most 2^-22 |x|, or 2^-25 absolute); the remainder is exact
__device__ __forceinline__ void mx_split(float x, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// one (distance, index) candidate into a running lexicographic minimum (NaN distances never enter)
__device__ __forceinline__ void mx_take(float d, int i, float &bd, int &bi) {
    const bool t = (d < bd) | ((d == bd) & (i < bi));              // (selects, not branches)
    bd = t ? d : bd;
    bi = t ? i : bi;
}

// What a workgroup needs to know about the launch (filled from ChamferSymArgs by the kernel in chamfer_sym.hip)
struct MxView {
    const float *P, *Q;        // this cloud's rows / columns
    int n, m;
    int rt, cs;                // row super-tile, column slice
    int C, S;                  // columns per stage, stages
};

// out_row(j, d, i): row j of the cloud has minimum (d, i) over this workgroup's columns; out_col(k, d, i): column k over its rows
template <int NCT, class OutRow, class OutCol>
__device__ __forceinline__ void mx_scan_block(const MxView &v, unsigned *lds, OutRow out_row, OutCol out_col) {
    float4 *colv = reinterpret_cast<float4 *>(lds + MX_OFF_COLS);       // [S * C]  (x, y, z, -)  padding: +inf
    unsigned *bex = lds + MX_OFF_B;
    float4 *rowv = reinterpret_cast<float4 *>(lds + MX_OFF_ROWS);       // [2048]   padding: +inf
    unsigned *tbase = lds + MX_OFF_T;
    // [2048] running row minima over the stages as ONE word each, (distance bits << 32) | index: squared distances are >= +0 (or
    // +inf), so the order of the words is the lexicographic order of (distance, index) -- a listed row may be answered for two
    // stages in the same batch below, by two groups of lanes at once: ds_min_u64, not read-merge-write (several stages only)
    unsigned long long *rbest = reinterpret_cast<unsigned long long *>(lds + MX_OFF_BEST);
    auto rb_pack = [](float d, int i) { return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)i; };
    float *red = reinterpret_cast<float *>(lds + MX_OFF_RED);
    unsigned *ritems = lds + MX_OFF_ITEMS + (size_t)(threadIdx.x >> 6) * (MX_ICAP + MX_CCAP), *citems = ritems + MX_ICAP;   // this wave's lists

    const int n = v.n, m = v.m, C = v.C, S = v.S;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int mrow = lane & 31, g = lane >> 5;
    const int q0 = v.rt * MX_ROWS + wave * MX_WROWS;               // first row of this wave
    const int cbase = v.cs * S * C;                                // first column of this workgroup
    const int ncols = min(S * C, m - cbase);                       // real columns of this workgroup (>= 1)

    // ---- rows (clamped: a padding row repeats the last one for the bounding box and is neutralised in its operand) and columns ----
    float px[MX_RT], py[MX_RT], pz[MX_RT];
#pragma unroll
    for (int r = 0; r < MX_RT; ++r) {
        int j = q0 + r * 32 + mrow;
        j = j < n ? j : n - 1;
        px[r] = v.P[3 * (size_t)j]; py[r] = v.P[3 * (size_t)j + 1]; pz[r] = v.P[3 * (size_t)j + 2];
    }
    constexpr int QU = MX_CMAX * MX_MAX_STAGES / MX_THREADS;       // columns per thread at most: t, t + 512, ... of the workgroup's S * C
    float qx[QU], qy[QU], qz[QU];
#pragma unroll
    for (int u = 0; u < QU; ++u) {
        int k = t + u * MX_THREADS;
        k = k < ncols ? k : ncols - 1;                             // (clamped for the box; staged as +inf below)
        qx[u] = v.Q[3 * (size_t)(cbase + k)]; qy[u] = v.Q[3 * (size_t)(cbase + k) + 1]; qz[u] = v.Q[3 * (size_t)(cbase + k) + 2];
    }
    // bounding boxes of the workgroup's rows and columns: ONE exchange
    {
        float lo[6] = {px[0], py[0], pz[0], qx[0], qy[0], qz[0]}, hi[6] = {px[0], py[0], pz[0], qx[0], qy[0], qz[0]};
#pragma unroll
        for (int r = 1; r < MX_RT; ++r) {
            lo[0] = fminf(lo[0], px[r]); hi[0] = fmaxf(hi[0], px[r]);
            lo[1] = fminf(lo[1], py[r]); hi[1] = fmaxf(hi[1], py[r]);
            lo[2] = fminf(lo[2], pz[r]); hi[2] = fmaxf(hi[2], pz[r]);
        }
#pragma unroll
        for (int u = 1; u < QU; ++u) {
            lo[3] = fminf(lo[3], qx[u]); hi[3] = fmaxf(hi[3], qx[u]);
            lo[4] = fminf(lo[4], qy[u]); hi[4] = fmaxf(hi[4], qy[u]);
            lo[5] = fminf(lo[5], qz[u]); hi[5] = fmaxf(hi[5], qz[u]);
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) { lo[c] = mx_wave_min(lo[c]); hi[c] = mx_wave_max(hi[c]); }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 6; ++c) { red[wave * 12 + c] = lo[c]; red[wave * 12 + 6 + c] = hi[c]; }
        }
    }
    // the rows and the columns into LDS for the exact evaluations (padding: +inf -- such a distance is inf or NaN and never wins)
    if (g == 0) {
#pragma unroll
        for (int r = 0; r < MX_RT; ++r) {
            const bool pad = q0 + r * 32 + mrow >= n;
            rowv[wave * MX_WROWS + r * 32 + mrow] = pad ? make_float4(INFINITY, INFINITY, INFINITY, 0.f) : make_float4(px[r], py[r], pz[r], 0.f);
        }
    }
#pragma unroll
    for (int u = 0; u < QU; ++u) {
        const int k = t + u * MX_THREADS;
        if (k < S * C) colv[k] = k < ncols ? make_float4(qx[u], qy[u], qz[u], 0.f) : make_float4(INFINITY, INFINITY, INFINITY, 0.f);
    }
    __syncthreads();
    float cen[3], big = 0.f, boxn2 = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float lo = red[c], hi = red[6 + c], clo = red[3 + c], chi = red[9 + c];
#pragma unroll
        for (int w = 1; w < MX_WAVES; ++w) {
            lo = fminf(lo, red[w * 12 + c]); hi = fmaxf(hi, red[w * 12 + 6 + c]);
            clo = fminf(clo, red[w * 12 + 3 + c]); chi = fmaxf(chi, red[w * 12 + 9 + c]);
        }
        cen[c] = 0.5f * lo + 0.5f * hi;
        const float re = fmaxf(hi - cen[c], cen[c] - lo), ce = fmaxf(chi - cen[c], cen[c] - clo);
        big = fmaxf(big, fmaxf(re, ce));
        boxn2 += re * re + ce * ce;
    }
    // scale: the largest centred |coordinate| into [2^13, 2^14).  Outside a sane range (or not finite) every chunk is a candidate.
    const bool sane = big >= 0x1p-40f && big <= 0x1p40f;           // (false for NaN / inf / collapsed clouds)
    const int ex = (int)((__float_as_uint(big) >> 23) & 0xff) - 127;   // big = f * 2^ex, 1 <= f < 2
    const float s = sane ? __uint_as_float((unsigned)(13 - ex + 127) << 23) : 1.0f;
    // the bias only has to be >= 2 eps: from the boxes' corners (known now); eps itself from the true norms (known after the operands)
    const float eps_box = (2.0f * boxn2 * (s * s) * 1.0001f) * 0x1p-18f + 32.0f;
    const float bias = __uint_as_float((__float_as_uint(2.0f * eps_box) + 0x007fffffu) & 0x7f800000u);   // a power of two (<= 2^16)

    // ---- B operands of stage 0 (one thread per column builds all 16 slots), largest norms, A operands ----
    // slot  0..8 : x11 x12 x21 y11 y12 y21 z11 z12 z21      A = -2 h_a, B = g_b
    //       9,10 : |u|^2 2^-14 pieces x 2^14                 11,12 : 2^14 x |v|^2 2^-14 pieces
    //       13   : bias x 1        14 : 2^15 x (column padding ? 65504 : 0)        15 : (row padding ? 65504 : 0) x 2^15
    auto build_b = [&](int sbeg) {
        if (t < C) {
            const int k = sbeg + t;
            const bool pad = k >= ncols;
            const float4 q = colv[k];
            const float wx = pad ? 0.f : (q.x - cen[0]) * s, wy = pad ? 0.f : (q.y - cen[1]) * s, wz = pad ? 0.f : (q.z - cen[2]) * s;
            _Float16 gx1, gx2, gy1, gy2, gz1, gz2, n1, n2;
            mx_split(wx, gx1, gx2); mx_split(wy, gy1, gy2); mx_split(wz, gz1, gz2);
            const float nn = ((wx * wx + wy * wy) + wz * wz) * 0x1p-14f;
            mx_split(nn, n1, n2);
            if (pad) { n1 = (_Float16)65504.0f; n2 = (_Float16)65504.0f; }
            const mx_f16x8 b0 = {gx1, gx2, gx1, gy1, gy2, gy1, gz1, gz2};
            const mx_f16x8 b1 = {gz1, (_Float16)16384.0f, (_Float16)16384.0f, n1, n2, (_Float16)1.0f,
                                 pad ? (_Float16)65504.0f : (_Float16)0.0f, (_Float16)32768.0f};
            *reinterpret_cast<mx_f16x8 *>(bex + (size_t)t * MX_BSTRIDE) = b0;
            *reinterpret_cast<mx_f16x8 *>(bex + (size_t)t * MX_BSTRIDE + 4) = b1;
        }
    };
    build_b(0);
    float rn2 = 0.f, cn2 = 0.f;
#pragma unroll
    for (int u = 0; u < QU; ++u) {
        const float cx = qx[u] - cen[0], cy = qy[u] - cen[1], cz = qz[u] - cen[2];
        cn2 = fmaxf(cn2, (cx * cx + cy * cy) + cz * cz);
    }
    mx_f16x8 Af[MX_RT];
#pragma unroll
    for (int r = 0; r < MX_RT; ++r) {
        const bool pad = q0 + r * 32 + mrow >= n;
        const float cx = px[r] - cen[0], cy = py[r] - cen[1], cz = pz[r] - cen[2];
        rn2 = fmaxf(rn2, (cx * cx + cy * cy) + cz * cz);
        const float ux = pad ? 0.f : cx * s, uy = pad ? 0.f : cy * s, uz = pad ? 0.f : cz * s;
        _Float16 hx1, hx2, hy1, hy2, hz1, hz2, n1, n2;
        mx_split(ux, hx1, hx2); mx_split(uy, hy1, hy2); mx_split(uz, hz1, hz2);
        const float nn = ((ux * ux + uy * uy) + uz * uz) * 0x1p-14f;
        mx_split(nn, n1, n2);
        if (pad) { n1 = (_Float16)65504.0f; n2 = (_Float16)65504.0f; }
        const _Float16 m2 = (_Float16)-2.0f;
        mx_f16x8 a0 = {m2 * hx1, m2 * hx1, m2 * hx2, m2 * hy1, m2 * hy1, m2 * hy2, m2 * hz1, m2 * hz1};
        mx_f16x8 a1 = {m2 * hz2, n1, n2, (_Float16)16384.0f, (_Float16)16384.0f, (_Float16)bias, (_Float16)32768.0f,
                       pad ? (_Float16)65504.0f : (_Float16)0.0f};
        Af[r] = g ? a1 : a0;
    }
    rn2 = mx_wave_max(rn2); cn2 = mx_wave_max(cn2);
    if (lane == 0) { red[128 + wave * 2] = rn2; red[128 + wave * 2 + 1] = cn2; }
    __syncthreads();                                               // B exchange of stage 0 and the norms are in LDS
#pragma unroll
    for (int w = 0; w < MX_WAVES; ++w) { rn2 = fmaxf(rn2, red[128 + w * 2]); cn2 = fmaxf(cn2, red[128 + w * 2 + 1]); }
    const float T = 2.0f * (rn2 + cn2) * (s * s) * 1.0001f;
    const float eps = T * 0x1p-18f + 32.0f;
    // candidate threshold as key bits: value <= v1 + 2 eps + the keys' truncation (2^-16 v1 at most)
    auto thr_bits = [&](unsigned m1, unsigned idmask) -> unsigned {
        const float v1 = __uint_as_float(m1 & ~idmask);
        const float th = (v1 + 2.0f * eps) + v1 * 0x1p-15f;
        return sane ? (__float_as_uint(th) | idmask) : 0xffffffffu;
    };
    GA_STAMP(0, 1);

    unsigned *tb = tbase + wave * (16 * MX_TSTRIDE);
    const int rowsub_r = lane >> 2, rowsub_h = (lane >> 1) & 1, half = lane & 1;      // the row phase's lane roles
    const int rowin = 8 * (rowsub_r >> 2) + 4 * rowsub_h + (rowsub_r & 3);             // row inside a tile served by this lane pair
    // queries that could not be certified: answered after the scan by whole waves (wave-uniform masks: bit = the lane that served it)

    // Queries that cannot be certified are LISTED (stage, row / column inside the workgroup) and answered after the last stage by
    // whole groups of lanes: answered stage by stage they cost a round of LDS latency on one wave while seven waited at the
    // stage's barrier -- 10-12 % of the kernel at 5 rows per workgroup and stage.
    int nritems = 0, ncitems = 0, stages_done = 0;                 // (wave-uniform)
#pragma unroll 1
    for (int st = 0; st < S; ++st) {
        const int sbeg = st * C;                                   // first column of the stage inside the workgroup
        if (sbeg >= ncols) break;                                  // (uniform)
        mx_f16x8 Bf[NCT];                                         // (a later stage's exchange was filled at the end of the one before)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) Bf[ct] = *reinterpret_cast<const mx_f16x8 *>(bex + (size_t)(ct * 32 + mrow) * MX_BSTRIDE + 4 * g);
        const bool more = st + 1 < S && sbeg + C < ncols;          // another stage follows
        GA_STAMP(0, 2);

        unsigned cm1[NCT], cm2[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) { cm1[ct] = MX_KEY_MAX; cm2[ct] = MX_KEY_MAX; }
        const mx_f32x16 zero = {0.f};
        // the products of a stage as one software pipeline: step q = (row tile q / NP, column tile pair q % NP); the matrix
        // instructions of step q + 1 are issued before step q's minima (two accumulator sets)
        constexpr int NP = NCT / 2, STEPS = MX_RT * NP;
        mx_f32x16 acc[2][2];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[0], Bf[0], zero, 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[0], Bf[1], zero, 0, 0, 0);
        unsigned run[16];
#pragma unroll
        for (int q = 0; q < STEPS; ++q) {
            const int r = q / NP, ct = 2 * (q % NP), cur = q & 1;
            if (q + 1 < STEPS) {
                const int rn = (q + 1) / NP, ctn = 2 * ((q + 1) % NP);
                acc[cur ^ 1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[rn], Bf[ctn], zero, 0, 0, 0);
                acc[cur ^ 1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[rn], Bf[ctn + 1], zero, 0, 0, 0);
            }
            unsigned e0[16], e1[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { e0[i] = __float_as_uint(acc[cur][0][i]); e1[i] = __float_as_uint(acc[cur][1][i]); }
            if (ct == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) run[i] = min(e0[i], e1[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) run[i] = mx_min3(run[i], e0[i], e1[i]);
            }
            // columns: minimum over the lane's 16 rows of the tile, keyed with (wave, row tile, lane half)
            {
                const unsigned cid = ((unsigned)wave << 4) | ((unsigned)r << 1) | (unsigned)g;
                unsigned c0 = mx_min3(mx_min3(e0[0], e0[1], e0[2]), mx_min3(e0[3], e0[4], e0[5]), mx_min3(e0[6], e0[7], e0[8]));
                c0 = mx_min3(c0, mx_min3(e0[9], e0[10], e0[11]), mx_min3(e0[12], e0[13], e0[14]));
                c0 = min(c0, e0[15]);
                unsigned c1 = mx_min3(mx_min3(e1[0], e1[1], e1[2]), mx_min3(e1[3], e1[4], e1[5]), mx_min3(e1[6], e1[7], e1[8]));
                c1 = mx_min3(c1, mx_min3(e1[9], e1[10], e1[11]), mx_min3(e1[12], e1[13], e1[14]));
                c1 = min(c1, e1[15]);
                const unsigned k0 = (c0 & ~MX_COL_IDMASK) | cid, k1 = (c1 & ~MX_COL_IDMASK) | cid;
                cm2[ct] = mx_med3(k0, cm1[ct], cm2[ct]); cm1[ct] = min(cm1[ct], k0);
                cm2[ct + 1] = mx_med3(k1, cm1[ct + 1], cm2[ct + 1]); cm1[ct + 1] = min(cm1[ct + 1], k1);
            }
            if (ct + 2 == NCT) {
                // ---- rows of tile r: transpose the running minima, two lanes per row; straight-line (an uncertified row only sets a bit) ----
#pragma unroll
                for (int i = 0; i < 16; ++i) tb[i * MX_TSTRIDE + lane] = run[i];
                const uint4 *src = reinterpret_cast<const uint4 *>(tb + rowsub_r * MX_TSTRIDE + 32 * rowsub_h + 16 * half);
                const uint4 w0 = src[0], w1 = src[1], w2 = src[2], w3 = src[3];
                const float4 tp = rowv[wave * MX_WROWS + r * 32 + rowin];
                const unsigned vals[16] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
                unsigned m1 = (vals[0] & ~MX_ROW_IDMASK) | (unsigned)(16 * half), m2 = MX_KEY_MAX;
#pragma unroll
                for (int i = 1; i < 16; ++i) {
                    const unsigned key = (vals[i] & ~MX_ROW_IDMASK) | (unsigned)(16 * half + i);
                    m2 = mx_med3(key, m1, m2);
                    m1 = min(m1, key);
                }
                const unsigned o1 = mx_dpp_xor1(m1), o2 = mx_dpp_xor1(m2);       // the pair's other half: smallest two of the four
                const unsigned M1 = min(m1, o1), M2 = min(max(m1, o1), min(m2, o2));
                const bool amb = M2 <= thr_bits(M1, MX_ROW_IDMASK);
                const int jrow = wave * MX_WROWS + r * 32 + rowin; // row inside the workgroup
                {
                    const unsigned long long am = __ballot(amb && half == 0);
                    if (amb && half == 0) {
                        const int slot = nritems + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(am >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)am, 0u));
                        if (slot < MX_ICAP) ritems[slot] = ((unsigned)st << 11) | (unsigned)jrow;
                        if (S > 1 && st == 0) rbest[jrow] = 0x7f8000007fffffffull;   // (+inf, no index: later stages and the list fold into it)
                    }
                    nritems += __builtin_popcountll(am);
                }
                // the chunk of M1: its NCT columns, split over the pair
                const int ch = (int)(M1 & MX_ROW_IDMASK);
                float bd = INFINITY;
                int bi = INT_MAX;
                {
                    float dd[NCT / 2];
#pragma unroll
                    for (int c2 = 0; c2 < NCT / 2; ++c2) {
                        const float4 q = colv[sbeg + (2 * c2 + half) * 32 + ch];
                        dd[c2] = mx_sqdist(q.x, q.y, q.z, tp.x, tp.y, tp.z);
                        bd = fminf(bd, dd[c2]);                    // (NaN distances skipped; all NaN or inf: bd stays inf)
                    }
#pragma unroll
                    for (int c2 = NCT / 2 - 1; c2 >= 0; --c2)      // descending: the last hit kept is the lowest column
                        bi = dd[c2] == bd ? sbeg + (2 * c2 + half) * 32 + ch : bi;
                }
                {
                    const float od = mx_dpp_xor1f(bd);
                    const int oi = (int)mx_dpp_xor1((unsigned)bi);
                    mx_take(od, oi, bd, bi);
                }
                if (bi == INT_MAX) bi = sbeg;                      // every distance inf / NaN: any valid index (uncertified anyway)
                if (half == 0 && !amb) {                           // (an uncertified row is answered from the list, after the last stage)
                    if (S > 1) {
                        const unsigned long long nw = rb_pack(bd, bi);
                        rbest[jrow] = st > 0 ? min(rbest[jrow], nw) : nw;   // (this lane alone touches the row during the stages)
                    } else if (v.rt * MX_ROWS + jrow < n) out_row(v.rt * MX_ROWS + jrow, bd, cbase + bi);
                }
            }
        }
        GA_STAMP(0, 3);
        // ---- columns: merge the lane halves, park the waves' pairs in LDS (the transpose buffers are free), two threads per column ----
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const unsigned o1 = (unsigned)__shfl_xor((int)cm1[ct], 32), o2 = (unsigned)__shfl_xor((int)cm2[ct], 32);
            const unsigned M1 = min(cm1[ct], o1), M2 = min(max(cm1[ct], o1), min(cm2[ct], o2));
            if (g == 0) { tb[2 * (ct * 32 + mrow)] = M1; tb[2 * (ct * 32 + mrow) + 1] = M2; }
        }
        __syncthreads();
        GA_STAMP(0, 4);
        {
            const int col = t >> 1, hf = t & 1;
            const int cc = col < C ? col : 0;
            unsigned M1 = MX_KEY_MAX, M2 = MX_KEY_MAX;
#pragma unroll
            for (int w = 0; w < MX_WAVES; ++w) {
                const uint2 p = *reinterpret_cast<const uint2 *>(tbase + (size_t)w * (16 * MX_TSTRIDE) + 2 * cc);
                M2 = min(max(M1, p.x), min(M2, p.y)); M1 = min(M1, p.x);
            }
            const int k = sbeg + cc;
            const unsigned thr = thr_bits(M1, MX_COL_IDMASK);
            const bool amb = M2 <= thr;
            const float4 tq = colv[k];
            float bd = INFINITY;
            int bi = INT_MAX;
            // the 16 rows of chunk (wave w, row tile r, lane half h): 8 (i / 4) + 4 h + i % 4, i = 0..15; this thread takes i / 4 in {2 hf, 2 hf + 1}
            auto chunk = [&](unsigned key) {
                const int cw = (int)(key >> 4) & 7, cr = (int)(key >> 1) & 7, chh = (int)key & 1;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int j = cw * MX_WROWS + cr * 32 + 8 * (2 * hf + q) + 4 * chh;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float4 p = rowv[j + u];
                        mx_take(mx_sqdist(p.x, p.y, p.z, tq.x, tq.y, tq.z), j + u, bd, bi);
                    }
                }
            };
            chunk(M1);
            bool defer = false;
            if (amb) {                                             // (a few per cent of the columns) every stored candidate in reach; a wave whose
                for (int w = 0; w < MX_WAVES; ++w) {               // SECOND chunk is in reach may hide a third: that column goes to a whole wave
                    const uint2 p = *reinterpret_cast<const uint2 *>(tbase + (size_t)w * (16 * MX_TSTRIDE) + 2 * cc);
                    if (p.x <= thr && p.x != M1) chunk(p.x);
                    if (p.y <= thr) { chunk(p.y); defer = true; }
                }
            }
            {
                const float od = mx_dpp_xor1f(bd);
                const int oi = (int)mx_dpp_xor1((unsigned)bi);
                mx_take(od, oi, bd, bi);
            }
            if (bi == INT_MAX) bi = 0;
            const bool live = hf == 0 && col < C && k < ncols;
            if (!defer && live) out_col(cbase + k, bd, min(v.rt * MX_ROWS + bi, n - 1));
            {
                const unsigned long long cm = __ballot(defer && live);
                if (defer && live) {
                    const int slot = ncitems + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
                    if (slot < MX_CCAP) citems[slot] = (unsigned)k;
                }
                ncitems += __builtin_popcountll(cm);
            }
        }
        stages_done = st + 1;
        GA_STAMP(0, 5);
        if (more) {
            build_b(sbeg + C);                                     // the next stage's operands (the exchange was read out at this stage's start)
            __syncthreads();                                       // ... and the column keys and the transpose buffers are free again
        }
    }
    // ---- the listed queries, with the reference's arithmetic on every pair of theirs.  Rows: FOUR at a time, one per 16-lane row of
    // the wave over the C columns of the item's stage (all requested before the first is used; four DPP steps fold a row of lanes).
    // A wave whose list overflowed (degenerate data: everything uncertified) redoes every row of its own at every stage.
    {
        const int grp = lane >> 4, sub = lane & 15;
        const bool rover = nritems > MX_ICAP;
#ifdef MX_EXP_NO_ITEMS
        const int count = 0;
#else
        const int count = rover ? MX_WROWS * stages_done : nritems;
#endif
#ifdef GA_STAMPS
        if (lane == 0 && blockIdx.x < GA_STAMP_BLOCKS) {           // diagnostic build: listed rows / columns of the workgroup into stamp slot 1
            atomicAdd(&ga_stamps[(1 * GA_STAMP_BLOCKS + blockIdx.x) * 8 + 1], (unsigned long long)nritems);
            atomicAdd(&ga_stamps[(1 * GA_STAMP_BLOCKS + blockIdx.x) * 8 + 2], (unsigned long long)ncitems);
            atomicAdd(&ga_stamps[(1 * GA_STAMP_BLOCKS + blockIdx.x) * 8 + 3], 1ull);
        }
#endif
#pragma unroll 1
        for (int base = 0; base < count; base += 4) {
            const int e = base + grp;
            const bool have = e < count;
            const unsigned item = !have ? 0u : rover ? (((unsigned)(e / MX_WROWS) << 11) | (unsigned)(wave * MX_WROWS + e % MX_WROWS)) : ritems[e];
            const int ist = (int)(item >> 11), jrow = (int)(item & 2047u), isb = ist * C;
            const bool live = have && v.rt * MX_ROWS + jrow < n;
            const float4 tp = rowv[jrow];
            float bd = INFINITY;
            int bi = INT_MAX;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                float4 q[NCT];
#pragma unroll
                for (int u = 0; u < NCT; ++u) q[u] = colv[isb + (h2 * NCT + u) * 16 + sub];
#pragma unroll
                for (int u = 0; u < NCT; ++u) mx_take(mx_sqdist(q[u].x, q[u].y, q[u].z, tp.x, tp.y, tp.z), isb + (h2 * NCT + u) * 16 + sub, bd, bi);
            }
#define MX_LEX(CTRL)                                                                                      \
            {                                                                                             \
                const float d2_ = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bd), CTRL, 0xf, 0xf, false)); \
                const int k2_ = __builtin_amdgcn_update_dpp(0, bi, CTRL, 0xf, 0xf, false);                \
                mx_take(d2_, k2_, bd, bi);                                                                \
            }
            MX_LEX(0xB1) MX_LEX(0x4E) MX_LEX(0x141) MX_LEX(0x140)
#undef MX_LEX
            if (bi == INT_MAX) bi = isb;
            if (live && sub == 0) {
                if (S > 1) atomicMin(&rbest[jrow], rb_pack(bd, bi));    // (another group of this batch may hold the same row's other stage)
                else out_row(v.rt * MX_ROWS + jrow, bd, cbase + bi);
            }
        }
        // columns whose candidates may be incomplete: a whole wave over all 2048 rows, eight requests ahead.  (A wave whose list
        // overflowed redoes every column of its own: 32 per stage.)
        const bool cover = ncitems > MX_CCAP;
        const int ccount = cover ? 32 * stages_done : ncitems;
#pragma unroll 1
        for (int e = 0; e < ccount; ++e) {
            const int k = cover ? (e / 32) * C + wave * 32 + e % 32 : (int)citems[e];
            if (k >= ncols || (cover && wave * 32 + e % 32 >= C)) continue;
            const float4 tq = colv[k];
            float bd = INFINITY;
            int bi = INT_MAX;
#pragma unroll 1
            for (int j0 = 0; j0 < MX_ROWS; j0 += 8 * 64) {
                float4 p[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) p[u] = rowv[j0 + u * 64 + lane];
#pragma unroll
                for (int u = 0; u < 8; ++u) mx_take(mx_sqdist(p[u].x, p[u].y, p[u].z, tq.x, tq.y, tq.z), j0 + u * 64 + lane, bd, bi);
            }
            wave_lexmin(bd, bi);
            if (bi == INT_MAX) bi = 0;
            if (lane == 0) out_col(cbase + k, bd, min(v.rt * MX_ROWS + bi, n - 1));
        }
    }
    // rows of several stages leave now: their running minimum is complete (each entry was written by the lane that reads it, or by
    // a lane of the same wave: program order)
    if (S > 1) {
        // (the lane's row inside its wave's tiles, derived afresh: kept alive across the stage loop it was the one register the
        // NCT = 8 instantiation spilled -- and a kernel with a scratch segment costs microseconds to launch)
        int t2 = threadIdx.x;
        asm volatile("" : "+v"(t2));
        const int sr2 = (t2 & 63) >> 2, rowin2 = 8 * (sr2 >> 2) + 4 * ((t2 >> 1) & 1) + (sr2 & 3), wrow2 = (t2 >> 6) * MX_WROWS + rowin2;
#pragma unroll
        for (int r = 0; r < MX_RT; ++r) {
            const int jrow = wrow2 + r * 32;
            if ((t2 & 1) == 0 && v.rt * MX_ROWS + jrow < n)
                out_row(v.rt * MX_ROWS + jrow, __uint_as_float((unsigned)(rbest[jrow] >> 32)), cbase + (int)(unsigned)rbest[jrow]);
        }
    }
}

}  // namespace geoadv
