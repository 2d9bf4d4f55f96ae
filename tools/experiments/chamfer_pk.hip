// Symmetric Chamfer with PACKED results, for the attack loop: nn_distance(P, Q) in ONE launch, no finish pass.
//
// As chamfer_sym.hip, every pair distance d(P_j, Q_k) = ((dx*dx)+(dy*dy))+(dz*dz) is evaluated once and serves both
// directions (bit-identical whichever cloud is the "query": tf_nndistance.cpp:21-43,79-80).  What is new is how the
// two reductions leave the kernel.  A result is ONE 64-bit word  (float bits of the distance << 32) | index :
// squared distances are >= +0, so their order as floats is their order as unsigned integers, and the unsigned minimum
// of such words is exactly the reference's rule -- smallest distance, lowest index among equals (strict '<' in ascending
// order).  A workgroup owns a tile of 256 rows and a slice of the columns; whatever it does not cover completely it
// contributes with a 64-bit atomic minimum (integer minima commute: deterministic), and a side it covers alone it simply
// stores.  So there are no partial-minima arrays in HBM (chamfer_sym.hip writes 16 B per (tile, column) and reads them back
// in its finish kernel: 8.4 MB each way at B = 32, N = 2048) and no second launch; the consumers (losses, Chamfer
// gradients) read dist = high word, idx = low word.  Words that receive atomics must hold ~0 before the launch: the
// caller resets them in an earlier launch of the same iteration (decoder.hip / chamfer_grid.h).
//
// Row side: as before (running minimum per chunk of 8 columns, winning chunk re-scanned, waves merged through LDS).
// Column side: a lane holds FOUR CONSECUTIVE rows (row = 4 * lane + r), so row order is (lane, r) order.  Per round of 16
// columns the lanes' minima over their 4 rows go through the LDS transpose; lane (column, quarter) then knows the
// minimum over 16 lanes AND the lowest lane attaining it (15 v_min + 16 compare/select pairs), a two-step DPP
// lexicographic minimum over the four quarters gives the tile's (minimum, lowest lane), and the row inside that lane
// is found by re-evaluating its 4 rows (from a 3 KB SoA copy of the tile in LDS) -- 4 distance evaluations per
// (tile, column) instead of the finish kernel's 64 per column, and exact ties resolve to the lowest row by construction.
#include "common.h"
#include <limits.h>
#include <math.h>

#pragma clang fp contract(off)

#ifndef PK_VARIANT
#define PK_VARIANT 0     // experiments (tools/debug/build_variants.sh): 1 = no column output, 2 = plain stores for atomics, 3 = no row-in-lane resolution
#endif

namespace geoadv {

typedef unsigned long long u64;

struct ChamferPk {
    const float *p, *q;        // [b][n][3] rows, [b][m][3] columns
    u64 *row, *col;            // [b][n], [b][m]: (dist bits << 32) | idx  = nn_distance outputs (0,1) and (2,3)
};
struct ChamferPkArgs {
    ChamferPk pr[2];
    int n, m, tiles, clouds, pairs, csplit;
    const int *need[2];        // per pair: null = every cloud; else int[8 * clouds] (see chamfer_sym.hip)
};

constexpr int PK_THREADS = 512;
constexpr int PK_WAVES = 8;
constexpr int PK_R = 4;                      // rows per lane: rows q0 + 4 * lane + r
constexpr int PK_ROWS = kWave * PK_R;        // 256 rows per workgroup
constexpr int PK_CHUNK = 8;                  // columns per arg-min chunk
constexpr int PK_ROUND = 16;                 // columns per transpose round
constexpr int PK_STAGE = 2048;               // columns per LDS stage
constexpr int PK_TSTRIDE = 68;               // floats per column in the transpose buffer (64 lanes + pad)
constexpr int PK_MAX_SPLIT = 8;

__device__ __forceinline__ float sqdist_p(float tx, float ty, float tz, float qx, float qy, float qz) {
    const float dx = tx - qx, dy = ty - qy, dz = tz - qz;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    return (xx + yy) + zz;
}

__device__ __forceinline__ bool pk_needed(const int *need, int c) {
    if (!need) return true;
    const int4 lo = reinterpret_cast<const int4 *>(need)[2 * c], hi = reinterpret_cast<const int4 *>(need)[2 * c + 1];
    return (lo.x | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) != 0;
}

__device__ __forceinline__ u64 pk_pack(unsigned dist_bits, int idx) { return ((u64)dist_bits << 32) | (unsigned)idx; }

// lexicographic minimum of (value, lane) with another lane of the quad (DPP quad_perm, register to register)
#define PK_QUAD_LEXMIN(CTRL)                                                                   \
    do {                                                                                       \
        const unsigned v2_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)qv, CTRL, 0xf, 0xf, false); \
        const int l2_ = __builtin_amdgcn_update_dpp(0, ql, CTRL, 0xf, 0xf, false);              \
        const bool take_ = v2_ < qv || (v2_ == qv && l2_ < ql);                                 \
        qv = take_ ? v2_ : qv;                                                                  \
        ql = take_ ? l2_ : ql;                                                                  \
    } while (0)

#if PK_VARIANT == 20
__device__ unsigned long long pk_stamps[8 * 4096];
#define PK_STAMP(i)                                                                                     \
    do {                                                                                                \
        if (threadIdx.x == 0 && blockIdx.x < 4096) {                                                     \
            unsigned long long t_;                                                                      \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
            pk_stamps[blockIdx.x * 8 + (i)] = t_;                                                        \
        }                                                                                               \
    } while (0)
#else
#define PK_STAMP(i)
#endif

__global__ __launch_bounds__(PK_THREADS, 4) void chamfer_pk_kernel(ChamferPkArgs a) {
    constexpr int R = PK_R;
    PK_STAMP(0);
    const int lin = blockIdx.x;                            // XCD-aware mapping, see chamfer_scan_kernel
    const int xcd = lin & 7, slot = lin >> 3;
    const int per = a.tiles * a.csplit;                    // workgroups per (pair, cloud) group
    const int group = (slot / per) * 8 + xcd, sub = slot % per;
    if (group >= a.clouds * a.pairs) return;
    const int tile = sub % a.tiles, cs = sub / a.tiles;    // row tile, column slice
    const int pi = group / a.clouds, c = group % a.clouds;
    if (!pk_needed(a.need[pi], c)) return;
    const ChamferPk pr = a.pr[pi];
    const int n = a.n, m = a.m;
    const int q0 = tile * PK_ROWS;
    const int mround = (m + PK_ROUND - 1) / PK_ROUND;
    const int mbeg = min(m, (mround * cs / a.csplit) * PK_ROUND), mend = min(m, (mround * (cs + 1) / a.csplit) * PK_ROUND);
    const float *P = pr.p + (size_t)c * n * 3;
    const float *Q = pr.q + (size_t)c * m * 3;
    u64 *rowout = pr.row + (size_t)c * n, *colout = pr.col + (size_t)c * m;

    __shared__ __attribute__((aligned(16))) float stage[3 * PK_STAGE];
    __shared__ __attribute__((aligned(16))) float tbuf[PK_WAVES][PK_ROUND * PK_TSTRIDE];
    __shared__ __attribute__((aligned(16))) float prow[3][PK_ROWS];       // the tile's rows, SoA, index 4 * lane + r
    float *sx = stage, *sy = stage + PK_STAGE, *sz = stage + 2 * PK_STAGE;
    static_assert(2 * PK_WAVES * PK_ROWS <= 3 * PK_STAGE, "merge arrays must fit in the stage buffer");
    float (*mdist)[PK_ROWS] = reinterpret_cast<float (*)[PK_ROWS]>(stage);
    int (*midx)[PK_ROWS] = reinterpret_cast<int (*)[PK_ROWS]>(stage + PK_WAVES * PK_ROWS);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float px[R], py[R], pz[R], best[R];
    int bestk[R];
    if (q0 + PK_ROWS <= n) {                               // a full tile: 48 contiguous bytes per lane
        const float4 *src = reinterpret_cast<const float4 *>(P + 3 * (size_t)(q0 + 4 * lane));
        const float4 u0 = src[0], u1 = src[1], u2 = src[2];
        px[0] = u0.x; py[0] = u0.y; pz[0] = u0.z; px[1] = u0.w; py[1] = u1.x; pz[1] = u1.y;
        px[2] = u1.z; py[2] = u1.w; pz[2] = u2.x; px[3] = u2.y; py[3] = u2.z; pz[3] = u2.w;
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int j = q0 + 4 * lane + r;
            j = j < n ? j : n - 1;                         // padding rows repeat the cloud's last row: same distances, and the
            px[r] = P[3 * j]; py[r] = P[3 * j + 1]; pz[r] = P[3 * j + 2];   // genuine row has the lower (lane, r) -- it wins every tie
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { best[r] = INFINITY; bestk[r] = -1; }
    if (wave == 0) {
        *reinterpret_cast<float4 *>(&prow[0][4 * lane]) = make_float4(px[0], px[1], px[2], px[3]);
        *reinterpret_cast<float4 *>(&prow[1][4 * lane]) = make_float4(py[0], py[1], py[2], py[3]);
        *reinterpret_cast<float4 *>(&prow[2][4 * lane]) = make_float4(pz[0], pz[1], pz[2], pz[3]);
    }
    float *tb = tbuf[wave];
    const int col = lane >> 2, quarter = lane & 3;
    PK_STAMP(1);
    for (int t0 = mbeg; t0 < mend; t0 += PK_STAGE) {
        const int cnt = min(PK_STAGE, mend - t0);
        const int cntp = (cnt + PK_ROUND - 1) / PK_ROUND * PK_ROUND;
        __syncthreads();
        for (int e = threadIdx.x; e < cntp; e += PK_THREADS) {
            float x = INFINITY, y = INFINITY, z = INFINITY;
            if (e < cnt) { x = Q[3 * (size_t)(t0 + e)]; y = Q[3 * (size_t)(t0 + e) + 1]; z = Q[3 * (size_t)(t0 + e) + 2]; }
            sx[e] = x; sy[e] = y; sz[e] = z;
        }
        __syncthreads();
        PK_STAMP(2);
        const int nrounds = cntp / PK_ROUND;
        const int rbeg = nrounds * wave / PK_WAVES, rend = nrounds * (wave + 1) / PK_WAVES;
        if (rbeg < rend) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (bestk[r] < 0) bestk[r] = t0 + rbeg * PK_ROUND;
        }
        for (int rd = rbeg; rd < rend; ++rd) {
            const int k0 = rd * PK_ROUND;
            float colp[PK_ROUND];
#pragma unroll
            for (int hf = 0; hf < PK_ROUND / PK_CHUNK; ++hf) {
                float tx[PK_CHUNK], ty[PK_CHUNK], tz[PK_CHUNK];
#pragma unroll
                for (int v = 0; v < PK_CHUNK / 4; ++v) {
                    const float4 xa = *reinterpret_cast<const float4 *>(&sx[k0 + hf * PK_CHUNK + 4 * v]);
                    const float4 ya = *reinterpret_cast<const float4 *>(&sy[k0 + hf * PK_CHUNK + 4 * v]);
                    const float4 za = *reinterpret_cast<const float4 *>(&sz[k0 + hf * PK_CHUNK + 4 * v]);
                    tx[4 * v] = xa.x; tx[4 * v + 1] = xa.y; tx[4 * v + 2] = xa.z; tx[4 * v + 3] = xa.w;
                    ty[4 * v] = ya.x; ty[4 * v + 1] = ya.y; ty[4 * v + 2] = ya.z; ty[4 * v + 3] = ya.w;
                    tz[4 * v] = za.x; tz[4 * v + 1] = za.y; tz[4 * v + 2] = za.z; tz[4 * v + 3] = za.w;
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float cm = INFINITY;
#pragma unroll
                    for (int u = 0; u < PK_CHUNK; ++u) {
                        const float d = sqdist_p(tx[u], ty[u], tz[u], px[r], py[r], pz[r]);
                        cm = fminf(cm, d);
                        colp[hf * PK_CHUNK + u] = r == 0 ? d : fminf(colp[hf * PK_CHUNK + u], d);
                    }
                    if (cm < best[r]) { best[r] = cm; bestk[r] = t0 + k0 + hf * PK_CHUNK; }
                    __builtin_amdgcn_sched_barrier(0);     // one row's eight distances die here (chamfer_sym.hip: register pressure)
                }
            }
            // 64-lane reduction of the 16 column partials through LDS: [column][lane] -> 4 lanes per column
#pragma unroll
            for (int u = 0; u < PK_ROUND; ++u) tb[u * PK_TSTRIDE + lane] = colp[u];
            __builtin_amdgcn_wave_barrier();
            {
                const uint4 *src = reinterpret_cast<const uint4 *>(tb + col * PK_TSTRIDE + quarter * 16);
                const uint4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
                const unsigned vv[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
                unsigned mb = min(min(min(v0.x, v0.y), min(v0.z, v0.w)), min(min(v1.x, v1.y), min(v1.z, v1.w)));
                mb = min(mb, min(min(min(v2.x, v2.y), min(v2.z, v2.w)), min(min(v3.x, v3.y), min(v3.z, v3.w))));
#if PK_VARIANT == 1
                if (mb == 12345u && vv[3] == 7u) colout[0] = mb;
#else
                int first = 15;                            // lowest of this quarter's 16 lanes attaining it
#if PK_VARIANT != 5 && PK_VARIANT != 9
#pragma unroll
                for (int i2 = 14; i2 >= 0; --i2) first = vv[i2] == mb ? i2 : first;
#endif
                unsigned qv = mb;
                int ql = quarter * 16 + first;
#if PK_VARIANT == 7 || PK_VARIANT == 8
                // value first (two DPP minima), then the lowest lane among the quarters that attain it (two more)
                qv = min(qv, (unsigned)__builtin_amdgcn_mov_dpp((int)qv, 0xB1, 0xf, 0xf, true));
                qv = min(qv, (unsigned)__builtin_amdgcn_mov_dpp((int)qv, 0x4E, 0xf, 0xf, true));
                ql = mb == qv ? ql : 255;
                ql = min(ql, __builtin_amdgcn_mov_dpp(ql, 0xB1, 0xf, 0xf, true));
                ql = min(ql, __builtin_amdgcn_mov_dpp(ql, 0x4E, 0xf, 0xf, true));
#elif PK_VARIANT != 6 && PK_VARIANT != 9
                PK_QUAD_LEXMIN(0xB1);                      // quad_perm [1,0,3,2]
                PK_QUAD_LEXMIN(0x4E);                      // quad_perm [2,3,0,1]: all four lanes of the column now agree
#endif
                // which of that lane's 4 rows?  re-evaluate them (descending, so the last hit kept is the lowest row)
                const int kk = k0 + col;                   // column inside the stage
                int f = 0;
#if PK_VARIANT != 3 && (PK_VARIANT < 4 || PK_VARIANT == 8)
                const float cx = sx[kk], cy = sy[kk], cz = sz[kk];
                const float4 rx = *reinterpret_cast<const float4 *>(&prow[0][4 * ql]);
                const float4 ry = *reinterpret_cast<const float4 *>(&prow[1][4 * ql]);
                const float4 rz = *reinterpret_cast<const float4 *>(&prow[2][4 * ql]);
                const float v = __uint_as_float(qv);
                f = sqdist_p(cx, cy, cz, rx.w, ry.w, rz.w) == v ? 3 : f;
                f = sqdist_p(cx, cy, cz, rx.z, ry.z, rz.z) == v ? 2 : f;
                f = sqdist_p(cx, cy, cz, rx.y, ry.y, rz.y) == v ? 1 : f;
                f = sqdist_p(cx, cy, cz, rx.x, ry.x, rz.x) == v ? 0 : f;
#endif
                const int k = t0 + kk;
#if PK_VARIANT == 4
                if (qv == 12345u && ql == 77) colout[0] = qv;
#else
                if (quarter == 0 && k < mend) {
                    int row = q0 + 4 * ql + f;
                    row = row < n ? row : n - 1;           // (a padding row can only win where v is NaN-tainted: out of contract)
                    const u64 w = pk_pack(qv, row);
#if PK_VARIANT == 2
                    colout[k] = w;
#else
                    if (a.tiles == 1) colout[k] = w;
                    else atomicMin(&colout[k], w);
#endif
                }
#endif
#endif
            }
            __builtin_amdgcn_wave_barrier();              // the buffer is rewritten by the next round
        }
    }
    PK_STAMP(3);
    // row minima: first index attaining the minimum inside the winning chunk, then merge the waves (chamfer_sym.hip)
    const bool staged = mend - mbeg <= PK_STAGE;            // uniform
    int found[R];
    if (staged) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            found[r] = INT_MAX;
            if (bestk[r] < 0) continue;                   // (uniform: a wave either scanned columns or did not)
            const int kb = bestk[r] - mbeg;
            const float4 xa = *reinterpret_cast<const float4 *>(&sx[kb]), xb = *reinterpret_cast<const float4 *>(&sx[kb + 4]);
            const float4 ya = *reinterpret_cast<const float4 *>(&sy[kb]), yb = *reinterpret_cast<const float4 *>(&sy[kb + 4]);
            const float4 za = *reinterpret_cast<const float4 *>(&sz[kb]), zb = *reinterpret_cast<const float4 *>(&sz[kb + 4]);
            const float tx[PK_CHUNK] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
            const float ty[PK_CHUNK] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
            const float tz[PK_CHUNK] = {za.x, za.y, za.z, za.w, zb.x, zb.y, zb.z, zb.w};
            int f = bestk[r];
#pragma unroll
            for (int u = PK_CHUNK - 1; u >= 0; --u)
                f = sqdist_p(tx[u], ty[u], tz[u], px[r], py[r], pz[r]) == best[r] ? bestk[r] + u : f;
            found[r] = f;
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            found[r] = INT_MAX;
            if (bestk[r] >= 0) {
                found[r] = bestk[r];
                bool hit = false;
                for (int u = 0; u < PK_CHUNK; ++u) {
                    const int k = bestk[r] + u;
                    if (k < mend) {
                        const float d = sqdist_p(Q[3 * (size_t)k], Q[3 * (size_t)k + 1], Q[3 * (size_t)k + 2], px[r], py[r], pz[r]);
                        if (!hit && d == best[r]) { hit = true; found[r] = k; }
                    }
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mdist[wave][4 * lane + r] = best[r];
        midx[wave][4 * lane + r] = found[r];
    }
    __syncthreads();
    for (int qq = threadIdx.x; qq < PK_ROWS; qq += PK_THREADS) {
        float d = mdist[0][qq];
        int k = midx[0][qq];
#pragma unroll
        for (int w = 1; w < PK_WAVES; ++w) {
            const float dw = mdist[w][qq];
            const int kw = midx[w][qq];
            if (dw < d || (dw == d && kw < k)) { d = dw; k = kw; }
        }
        if (q0 + qq < n) {
            const u64 w = pk_pack(__float_as_uint(d), k);
#if PK_VARIANT == 2
            rowout[q0 + qq] = w;
#else
            if (a.csplit == 1) rowout[q0 + qq] = w;
            else atomicMin(&rowout[q0 + qq], w);
#endif
        }
    }
    PK_STAMP(4);
}

// Whether a side of nn_distance(n rows, m columns) for `b` clouds receives atomics (and must be reset to ~0 before the
// launch): columns whenever there is more than one row tile, rows whenever the columns are sliced.
int chamfer_pk_csplit(int b, int n, int m) {
    const int tiles = cdiv(n, PK_ROWS);
    int cs = 1;
    // column slices so that the grid covers the chip (measured on chamfer_sym.hip: slicing only pays below that); a slice
    // keeps at least 256 columns = two rounds per wave
    while (cs < PK_MAX_SPLIT && (long)tiles * cs * b < kCUs && m / (cs * 2) >= 256) cs *= 2;
#ifdef PK_FORCE_SPLIT
    if (cs < PK_FORCE_SPLIT && m / PK_FORCE_SPLIT >= 256) cs = PK_FORCE_SPLIT;
#endif
    return cs;
}
bool chamfer_pk_rows_atomic(int b, int n, int m) { return chamfer_pk_csplit(b, n, m) > 1; }
bool chamfer_pk_cols_atomic(int n) { return cdiv(n, PK_ROWS) > 1; }

// pairs: up to 2 problems with identical (n, m) >= 1; need1 restricts the SECOND pair to flagged clouds.
int launch_chamfer_pk(const ChamferPk *pairs, int np, int b, int n, int m, const int *need1, hipStream_t stream) {
    if (b <= 0 || np <= 0) return GEOADV_OK;
    ChamferPkArgs a;
    a.need[0] = nullptr; a.need[1] = need1;
    for (int i = 0; i < np; ++i) a.pr[i] = pairs[i];
    a.n = n; a.m = m; a.tiles = cdiv(n, PK_ROWS); a.clouds = b; a.pairs = np;
    a.csplit = chamfer_pk_csplit(b, n, m);                 // (decided on ONE problem: a gated second pair usually adds nothing)
    const unsigned grid = (unsigned)(a.tiles * a.csplit * 8 * cdiv(b * np, 8));
    chamfer_pk_kernel<<<grid, PK_THREADS, 0, stream>>>(a);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

// ---- helpers for callers that keep separate (dist, idx) arrays ----
__global__ void pk_fill_kernel(u64 *p, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[i] = ~0ull;
}
__global__ void pk_pack_kernel(const float *d, const int *i, u64 *out, size_t count) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < count) out[e] = pk_pack(__float_as_uint(d[e]), i[e]);
}
__global__ void pk_unpack_kernel(const u64 *in, float *d, int *i, size_t count) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    const u64 w = in[e];
    if (d) d[e] = __uint_as_float((unsigned)(w >> 32));
    if (i) i[e] = (int)(unsigned)w;
}
int launch_pk_fill(u64 *p, size_t count, hipStream_t stream) {
    if (!count) return GEOADV_OK;
    pk_fill_kernel<<<(unsigned)((count + 255) / 256), 256, 0, stream>>>(p, count);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}
int launch_pk_pack(const float *d, const int *i, u64 *out, size_t count, hipStream_t stream) {
    if (!count) return GEOADV_OK;
    pk_pack_kernel<<<(unsigned)((count + 255) / 256), 256, 0, stream>>>(d, i, out, count);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}
int launch_pk_unpack(const u64 *in, float *d, int *i, size_t count, hipStream_t stream) {
    if (!count) return GEOADV_OK;
    pk_unpack_kernel<<<(unsigned)((count + 255) / 256), 256, 0, stream>>>(in, d, i, count);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

}  // namespace geoadv

using namespace geoadv;

#if PK_VARIANT == 20
extern "C" int geoadv_debug_pk_stamps(unsigned long long *host_out, int count) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pk_stamps), sizeof(unsigned long long) * (size_t)count) == hipSuccess ? 0 : 1;
}
#endif

// nn_distance through the packed symmetric kernel, results unpacked into the op's four arrays.  workspace: b * (n + m)
// 64-bit words of device scratch.  Same results as geoadv_nn_distance bit for bit; n, m >= 1.
extern "C" int geoadv_nn_distance_symmetric(int b, int n, const float *xyz1, int m, const float *xyz2, float *dist1, int *idx1,
                                            float *dist2, int *idx2, unsigned long long *workspace, void *stream) {
    GA_REQUIRE(b >= 0 && n >= 1 && m >= 1, "nn_distance_symmetric: bad dimensions (b=%d n=%d m=%d)", b, n, m);
    GA_REQUIRE(b <= 65535, "nn_distance_symmetric: batch %d exceeds 65535", b);
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(xyz1 && xyz2 && dist1 && idx1 && dist2 && idx2 && workspace, "nn_distance_symmetric: null pointer");
    hipStream_t st = as_stream(stream);
    u64 *row = workspace, *col = workspace + (size_t)b * n;
    if (chamfer_pk_rows_atomic(b, n, m)) { if (int rc = launch_pk_fill(row, (size_t)b * n, st)) return rc; }
    if (chamfer_pk_cols_atomic(n)) { if (int rc = launch_pk_fill(col, (size_t)b * m, st)) return rc; }
    const ChamferPk pr{xyz1, xyz2, row, col};
    if (int rc = launch_chamfer_pk(&pr, 1, b, n, m, nullptr, st)) return rc;
    if (int rc = launch_pk_unpack(row, dist1, idx1, (size_t)b * n, st)) return rc;
    return launch_pk_unpack(col, dist2, idx2, (size_t)b * m, st);
}
