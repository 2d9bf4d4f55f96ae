// Symmetric Chamfer for the attack loop: ONE evaluation of every pair distance serves both
// directions of nn_distance (note 4): d(P_j, Q_k) = ((dx*dx)+(dy*dy))+(dz*dz) is
// bit-identical whichever cloud is the "query" (the differences only change sign), so the row minima
// (dist1/idx1) and the column minima (dist2/idx2) of the same distance matrix are exactly what the two
// scans of the reference op produce (tf_nndistance.cpp:79-80).  Still brute force -- all n*m distances are
// evaluated, none is skipped -- but 8 VALU ops per pair are now spent once instead of twice.
//
// Kernel 1 (chamfer_sym_kernel): as chamfer_scan_kernel, a workgroup of 4 waves owns 64*R rows (P points,
// R per lane, in registers); wave w scans the w-th quarter of the columns (Q points, LDS-staged SoA planes,
// broadcast reads).  Row minima: running min per chunk of 8 columns + re-scan of the winning chunk, as
// before.  Column minima: every lane reduces its R rows in registers (v_min3), then the 64 lanes of the
// wave are reduced through a 4 KB LDS transpose per 16 columns (16 ds_write_b32 + 4 ds_read_b128 per lane:
// ~8 % on top of the distance arithmetic; a DPP butterfly per column would cost 19 %).
// Each column is visited by exactly one wave of the workgroup, so the reduced value IS the column's
// minimum over this row tile; the four partial minima it is folded from (one per 16 lanes = 64 rows) go to
// colpart[tile][k][0..3].
// Kernel 2 (chamfer_sym_finish_kernel): per column, minimum over the (tile, quarter) partials and the LOWEST tile
// attaining it; then only the 64 rows of that tile's winning quarter are re-evaluated to find the lowest row index
// with d == minimum (exactly the reference's tie rule): 1/32 of a scan at n = 2048.
#include "common.h"
#include "chamfer_grid.h"
#include "encoder_jac.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>

#pragma clang fp contract(off)

namespace geoadv {

struct ChamferPair {
    const float *p, *q;        // [b][n][3] rows, [b][m][3] columns
    float *dist1; int *idx1;   // [b][n]  row minima  (nn_distance outputs 0,1)
    float *dist2; int *idx2;   // [b][m]  column minima (outputs 2,3)
};
struct ChamferSymArgs {
    ChamferPair pr[2];
    int n, m, tiles, clouds, pairs, csplit;
    int fin_reps;              // finish kernel: column sub-slices of CF_COLS a workgroup walks after staging the row cloud once
    int pair_base, q_clouds;   // q_clouds > 0: cloud c is the pair (P cloud (pair_base+c)/q_clouds, Q cloud (pair_base+c)%q_clouds)
    float *colpart;            // [pairs][clouds][tiles][m][4]: minima over the four 64-row quarters of a tile (rows 64 r + 16 q + i)
    float *rowpart_d;          // [pairs][clouds][csplit][n]   row minima per column slice
    int *rowpart_i;
    const int *need[2];        // per pair: null = every cloud; else int[8 * clouds], cloud c is computed only if one of
                               // its 8 flags is set (a workgroup of the paired grid search, chamfer_grid.hip, gave up)
    GridRider rider;           // the attack loop: the paired grid search of (adv, source) as extra workgroups of the scan launch
    JacRider jac;              // ... and the encoder's pool Jacobian (encoder_jac.h): needed by the NEXT step's backward only
};

__device__ __forceinline__ bool sym_needed(const int *need, int c) {
    if (!need) return true;
    const int4 lo = reinterpret_cast<const int4 *>(need)[2 * c], hi = reinterpret_cast<const int4 *>(need)[2 * c + 1];
    return (lo.x | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) != 0;
}

constexpr int CS_THREADS = 512;               // 8 waves share one row tile and split the columns 8 ways
constexpr int CS_WAVES = 8;
constexpr int CS_R = 4;                      // rows per lane
constexpr int CS_ROWS = kWave * CS_R;        // 256 rows per workgroup
constexpr int CS_CHUNK = 8;                  // columns per arg-min chunk
constexpr int CS_ROUND = 16;                 // columns per transpose round
constexpr int CS_STAGE = 2048;               // columns per LDS stage
constexpr int CS_TSTRIDE = 68;               // floats per column in the transpose buffer (64 lanes + pad)

__device__ __forceinline__ float sqdist_s(float tx, float ty, float tz, float qx, float qy, float qz) {
    const float dx = tx - qx, dy = ty - qy, dz = tz - qz;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    return (xx + yy) + zz;
}

constexpr size_t CS_LDS_BYTES = sizeof(float) * (3 * CS_STAGE + CS_WAVES * CS_ROUND * CS_TSTRIDE);   // stage planes + transpose buffers

__global__ __launch_bounds__(CS_THREADS, 4) void chamfer_sym_kernel(ChamferSymArgs a) {
    constexpr int R = CS_R;
    extern __shared__ __attribute__((aligned(16))) float stage[];
    if (grid_rider_block<GR_MAX_N>(a.rider)) return;
    if (jac_rider_block(a.jac, stage)) return;
    GA_STAMP(0, 0);
    const int lin = blockIdx.x - a.rider.blocks;           // (the search's workgroups come first); XCD-aware mapping, see chamfer_scan_kernel
    const int xcd = lin & 7, slot = lin >> 3;
    const int per = a.tiles * a.csplit;                    // workgroups per (pair, cloud) group
    const int group = (slot / per) * 8 + xcd, sub = slot % per;
    if (group >= a.clouds * a.pairs) return;
    const int tile = sub % a.tiles, cs = sub / a.tiles;    // row tile, column slice
    const int pi = group / a.clouds, c = group % a.clouds;
    if (!sym_needed(a.need[pi], c)) return;
    const ChamferPair pr = a.pr[pi];
    const int n = a.n, m = a.m;
    const int q0 = tile * CS_ROWS;
    // this workgroup's columns: [mbeg, mend), slices aligned to the transpose round
    const int mround = (m + CS_ROUND - 1) / CS_ROUND;
    const int mbeg = min(m, (mround * cs / a.csplit) * CS_ROUND), mend = min(m, (mround * (cs + 1) / a.csplit) * CS_ROUND);
    const int cp = a.q_clouds > 0 ? (a.pair_base + c) / a.q_clouds : c;
    const int cq = a.q_clouds > 0 ? (a.pair_base + c) % a.q_clouds : c;
    const float *P = pr.p + (size_t)cp * n * 3;
    const float *Q = pr.q + (size_t)cq * m * 3;
    float *colpart = a.colpart + (((size_t)pi * a.clouds + c) * a.tiles + tile) * m * 4;      // [m][4 quarters]

    // all of it in the DYNAMIC region (CS_LDS_BYTES, or a rider's need if larger): a launch that hosts the grid search's
    // workgroups is charged max(scan, search) of LDS per workgroup, not the sum
    float (*tbuf)[CS_ROUND * CS_TSTRIDE] = reinterpret_cast<float (*)[CS_ROUND * CS_TSTRIDE]>(stage + 3 * CS_STAGE);
    float *sx = stage, *sy = stage + CS_STAGE, *sz = stage + 2 * CS_STAGE;
    static_assert(2 * CS_WAVES * CS_ROWS <= 3 * CS_STAGE, "merge arrays must fit in the stage buffer");
    float (*mdist)[CS_ROWS] = reinterpret_cast<float (*)[CS_ROWS]>(stage);
    int (*midx)[CS_ROWS] = reinterpret_cast<int (*)[CS_ROWS]>(stage + CS_WAVES * CS_ROWS);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float px[R], py[R], pz[R], best[R];
    int bestk[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int j = q0 + r * kWave + lane;
        j = j < n ? j : n - 1;                             // padding rows repeat the last row: same distances,
        px[r] = P[3 * j]; py[r] = P[3 * j + 1]; pz[r] = P[3 * j + 2];   // so the column minima are unaffected
        best[r] = INFINITY; bestk[r] = -1;
    }
    float *tb = tbuf[wave];
    for (int t0 = mbeg; t0 < mend; t0 += CS_STAGE) {
        const int cnt = min(CS_STAGE, mend - t0);
        const int cntp = (cnt + CS_ROUND - 1) / CS_ROUND * CS_ROUND;
        __syncthreads();
        for (int e = threadIdx.x; e < cntp; e += CS_THREADS) {
            float x = INFINITY, y = INFINITY, z = INFINITY;
            if (e < cnt) { x = Q[3 * (size_t)(t0 + e)]; y = Q[3 * (size_t)(t0 + e) + 1]; z = Q[3 * (size_t)(t0 + e) + 2]; }
            sx[e] = x; sy[e] = y; sz[e] = z;
        }
        __syncthreads();
        const int nrounds = cntp / CS_ROUND;
        const int rbeg = nrounds * wave / CS_WAVES, rend = nrounds * (wave + 1) / CS_WAVES;
        if (rbeg < rend) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (bestk[r] < 0) bestk[r] = t0 + rbeg * CS_ROUND;
        }
        for (int rd = rbeg; rd < rend; ++rd) {
            const int k0 = rd * CS_ROUND;
            float colp[CS_ROUND];
#pragma unroll
            for (int hf = 0; hf < CS_ROUND / CS_CHUNK; ++hf) {
                float tx[CS_CHUNK], ty[CS_CHUNK], tz[CS_CHUNK];
#pragma unroll
                for (int v = 0; v < CS_CHUNK / 4; ++v) {
                    const float4 xa = *reinterpret_cast<const float4 *>(&sx[k0 + hf * CS_CHUNK + 4 * v]);
                    const float4 ya = *reinterpret_cast<const float4 *>(&sy[k0 + hf * CS_CHUNK + 4 * v]);
                    const float4 za = *reinterpret_cast<const float4 *>(&sz[k0 + hf * CS_CHUNK + 4 * v]);
                    tx[4 * v] = xa.x; tx[4 * v + 1] = xa.y; tx[4 * v + 2] = xa.z; tx[4 * v + 3] = xa.w;
                    ty[4 * v] = ya.x; ty[4 * v + 1] = ya.y; ty[4 * v + 2] = ya.z; ty[4 * v + 3] = ya.w;
                    tz[4 * v] = za.x; tz[4 * v + 1] = za.y; tz[4 * v + 2] = za.z; tz[4 * v + 3] = za.w;
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float cm = INFINITY;
#pragma unroll
                    for (int u = 0; u < CS_CHUNK; ++u) {
                        const float d = sqdist_s(tx[u], ty[u], tz[u], px[r], py[r], pz[r]);
                        cm = fminf(cm, d);
                        colp[hf * CS_CHUNK + u] = r == 0 ? d : fminf(colp[hf * CS_CHUNK + u], d);
                    }
                    if (cm < best[r]) { best[r] = cm; bestk[r] = t0 + k0 + hf * CS_CHUNK; }
                    // one row's eight distances die here: left alone the scheduler evaluates all 64 of the round first and
                    // folds the minima afterwards -- 54 VGPRs of live distances at the 128-register cap, spills around the
                    // loop (measured 32.5 -> 31.2 us).  (Fetching the next half's columns ahead on top of this spills 36
                    // registers: 56 us.)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // 64-lane reduction of the 16 column partials through LDS: [column][lane] -> 4 lanes per column
#pragma unroll
            for (int u = 0; u < CS_ROUND; ++u) tb[u * CS_TSTRIDE + lane] = colp[u];
            __builtin_amdgcn_wave_barrier();
            {
                const int col = lane >> 2, quarter = lane & 3;
                // squared distances are >= +0 (or +inf): their order as floats is their order as unsigned integers, and an
                // integer minimum needs no canonicalising v_max in front of every value that comes back from LDS
                const uint4 *src = reinterpret_cast<const uint4 *>(tb + col * CS_TSTRIDE + quarter * 16);
                const uint4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
                unsigned mb = min(min(min(v0.x, v0.y), min(v0.z, v0.w)), min(min(v1.x, v1.y), min(v1.z, v1.w)));
                mb = min(mb, min(min(min(v2.x, v2.y), min(v2.z, v2.w)), min(min(v3.x, v3.y), min(v3.z, v3.w))));
                // lane (col, quarter) now holds the minimum over lanes 16 quarter .. 16 quarter + 15, i.e. over the rows
                // {64 r + 16 quarter + i} of this tile: all four quarter minima are kept (one coalesced 256-byte store per
                // round) -- the finish kernel then re-evaluates 64 rows per column instead of 256
                const int k = t0 + k0 + col;
                if (k < mend) colpart[(size_t)k * 4 + quarter] = __uint_as_float(mb);
            }
            __builtin_amdgcn_wave_barrier();              // the buffer is rewritten by the next round
        }
    }
    GA_STAMP(0, 1);
    // row minima: first index attaining the minimum inside the winning chunk, then merge the waves.  With a single LDS
    // stage (m <= 2048 per slice: the attack's shape) the chunk is still in the stage buffer; otherwise it is re-read from
    // global memory.  The stage buffer becomes the merge arrays afterwards, hence the barrier between the two loops.
    const bool staged = mend - mbeg <= CS_STAGE;            // uniform
    int found[R];
    if (staged) {
        // branch-free: the chunk's eight columns come back as two ds_read_b128 per plane (chunks start at multiples of 8
        // inside the stage; columns beyond the slice are staged as +inf and can never equal a finite minimum), and the
        // hits are taken in DESCENDING order so that the last one kept is the lowest index -- a sixth of the
        // instructions of the generic loop below, which every wave used to run for its four rows
#pragma unroll
        for (int r = 0; r < R; ++r) {
            found[r] = INT_MAX;
            if (bestk[r] < 0) continue;                   // (uniform: a wave either scanned columns or did not)
            const int kb = bestk[r] - mbeg;
            const float4 xa = *reinterpret_cast<const float4 *>(&sx[kb]), xb = *reinterpret_cast<const float4 *>(&sx[kb + 4]);
            const float4 ya = *reinterpret_cast<const float4 *>(&sy[kb]), yb = *reinterpret_cast<const float4 *>(&sy[kb + 4]);
            const float4 za = *reinterpret_cast<const float4 *>(&sz[kb]), zb = *reinterpret_cast<const float4 *>(&sz[kb + 4]);
            const float tx[CS_CHUNK] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
            const float ty[CS_CHUNK] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
            const float tz[CS_CHUNK] = {za.x, za.y, za.z, za.w, zb.x, zb.y, zb.z, zb.w};
            int f = bestk[r];
#pragma unroll
            for (int u = CS_CHUNK - 1; u >= 0; --u)
                f = sqdist_s(tx[u], ty[u], tz[u], px[r], py[r], pz[r]) == best[r] ? bestk[r] + u : f;
            found[r] = f;
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            found[r] = INT_MAX;
            if (bestk[r] >= 0) {
                found[r] = bestk[r];
                bool hit = false;
                for (int u = 0; u < CS_CHUNK; ++u) {
                    const int k = bestk[r] + u;
                    if (k < mend) {
                        const float d = sqdist_s(Q[3 * (size_t)k], Q[3 * (size_t)k + 1], Q[3 * (size_t)k + 2], px[r], py[r], pz[r]);
                        if (!hit && d == best[r]) { hit = true; found[r] = k; }
                    }
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mdist[wave][r * kWave + lane] = best[r];
        midx[wave][r * kWave + lane] = found[r];
    }
    __syncthreads();
    for (int qq = threadIdx.x; qq < CS_ROWS; qq += CS_THREADS) {
        float d = mdist[0][qq];
        int k = midx[0][qq];
#pragma unroll
        for (int w = 1; w < CS_WAVES; ++w) {
            const float dw = mdist[w][qq];
            const int kw = midx[w][qq];
            if (dw < d || (dw == d && kw < k)) { d = dw; k = kw; }
        }
        if (q0 + qq < n) {
            if (a.csplit == 1) {
                pr.dist1[(size_t)c * n + q0 + qq] = d;
                pr.idx1[(size_t)c * n + q0 + qq] = k;
            } else {   // merged over the column slices by the finish kernel
                const size_t o = (((size_t)pi * a.clouds + c) * a.csplit + cs) * n + q0 + qq;
                a.rowpart_d[o] = d;
                a.rowpart_i[o] = k;
            }
        }
    }
    GA_STAMP(0, 7);
}

// grid = (column slices, clouds * pairs).  Every workgroup takes an equal slice of COLUMNS (so the work is
// balanced even when all column minima fall into one row tile, which is what a collapsed reconstruction
// produces) and keeps the whole row cloud in LDS as SoA planes, one padded segment per row tile.  Thread =
// one column: minimum over the (tile, quarter) partials + the lowest tile attaining it, then the 64 rows of that
// tile's winning quarter are re-evaluated for the lowest one with d == minimum (exactly the reference's tie rule).
// Several quarters of the tile attaining the minimum (exact ties between rows 16 apart or more) are rare: those
// columns walk their further quarters in a loop the other lanes sit out.
constexpr int CF_COLS = 512;                          // columns = threads per workgroup (measured: 128: 10.0, 256: 8.0, 512: 6.9, 1024: 8.9 us at B = 32, N = 2048 -- staging the rows per workgroup against workgroups per chip)
constexpr int CF_THREADS = CF_COLS;
constexpr int CF_SEG = CS_ROWS + 4;                   // floats per tile segment: 16-B aligned, and the pad staggers the banks

// lowest row of quarter q (rows 64 r + 16 q + i) of the tile staged at (sx, sy, sz) whose distance to the column equals v
__device__ __forceinline__ int finish_quarter(const float *sx, const float *sy, const float *sz, int q, float qx, float qy, float qz, float v) {
    int found = INT_MAX;
#pragma unroll
    for (int r = CS_R - 1; r >= 0; --r)                   // descending: the last hit kept is the lowest row
#pragma unroll
        for (int j4 = 3; j4 >= 0; --j4) {
            const int o = 64 * r + 16 * q + 4 * j4;
            const float4 xa = *reinterpret_cast<const float4 *>(sx + o);
            const float4 ya = *reinterpret_cast<const float4 *>(sy + o);
            const float4 za = *reinterpret_cast<const float4 *>(sz + o);
            const float d3 = sqdist_s(qx, qy, qz, xa.w, ya.w, za.w), d2 = sqdist_s(qx, qy, qz, xa.z, ya.z, za.z);
            const float d1 = sqdist_s(qx, qy, qz, xa.y, ya.y, za.y), d0 = sqdist_s(qx, qy, qz, xa.x, ya.x, za.x);
            const int j = 64 * r + 4 * j4;                // (+ 16 q, added by the caller: the selects keep inline constants)
            found = d3 == v ? j + 3 : found;
            found = d2 == v ? j + 2 : found;
            found = d1 == v ? j + 1 : found;
            found = d0 == v ? j : found;
        }
    return found == INT_MAX ? INT_MAX : found + 16 * q;
}

__global__ __launch_bounds__(CF_THREADS) void chamfer_sym_finish_kernel(ChamferSymArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    GA_STAMP(2, 0);
    const int tiles = a.tiles;
    float *rx = lds, *ry = lds + tiles * CF_SEG, *rz = lds + 2 * tiles * CF_SEG;
    const int group = blockIdx.y;
    const int pi = group / a.clouds, c = group % a.clouds;
    if (!sym_needed(a.need[pi], c)) return;
    const ChamferPair pr = a.pr[pi];
    const int n = a.n, m = a.m;
    const int cp = a.q_clouds > 0 ? (a.pair_base + c) / a.q_clouds : c;
    const int cq = a.q_clouds > 0 ? (a.pair_base + c) % a.q_clouds : c;
    const float *P = pr.p + (size_t)cp * n * 3;
    const float *Q = pr.q + (size_t)cq * m * 3;
    const float4 *colpart = reinterpret_cast<const float4 *>(a.colpart) + (((size_t)pi * a.clouds + c) * tiles) * m;
    // this thread's column: its partial minima and coordinates are REQUESTED before the row cloud is staged, so that the
    // two global round trips overlap (the kernel is latency-bound).  Large clouds (fin_reps > 1: staging the rows is
    // ~100 KB per workgroup, one workgroup per CU) walk several column sub-slices per staging.
    constexpr int CP = 8;                                 // tiles fetched up front (n <= 2048); further tiles in a loop
    int k = (blockIdx.x * a.fin_reps) * CF_COLS + threadIdx.x;
    int kc = k < m ? k : m - 1;
    float4 cpv[CP];
#pragma unroll
    for (int t = 0; t < CP; ++t) cpv[t] = colpart[(size_t)(t < tiles ? t : 0) * m + kc];
    float qx = Q[3 * (size_t)kc], qy = Q[3 * (size_t)kc + 1], qz = Q[3 * (size_t)kc + 2];
    for (int e = threadIdx.x; e < tiles * CS_ROWS; e += CF_THREADS) {
        const int o = (e / CS_ROWS) * CF_SEG + (e % CS_ROWS);
        const int src = e < n ? e : n - 1;                // rows beyond the cloud: copies of its last row (a copy can only
        rx[o] = P[3 * (size_t)src]; ry[o] = P[3 * (size_t)src + 1]; rz[o] = P[3 * (size_t)src + 2];   // tie with it, and it is lower)
    }
    __syncthreads();
    if (a.csplit > 1) {   // row minima: lexicographic (distance, index) minimum over the column slices
        for (int j = blockIdx.x * CF_THREADS + threadIdx.x; j < n; j += gridDim.x * CF_THREADS) {
            const size_t o = ((size_t)pi * a.clouds + c) * a.csplit * n + j;
            float d = a.rowpart_d[o];
            int i = a.rowpart_i[o];
            for (int s = 1; s < a.csplit; ++s) {
                const float ds = a.rowpart_d[o + (size_t)s * n];
                const int is = a.rowpart_i[o + (size_t)s * n];
                if (ds < d || (ds == d && is < i)) { d = ds; i = is; }
            }
            pr.dist1[(size_t)c * n + j] = d;
            pr.idx1[(size_t)c * n + j] = i;
        }
    }
    for (int rep = 0; rep < a.fin_reps; ++rep) {
        // minimum over the tiles and the LOWEST tile attaining it (strict compare), with that tile's four quarter minima
        float4 win = cpv[0];
        float v = fminf(fminf(win.x, win.y), fminf(win.z, win.w));
        int bt = 0;
#pragma unroll
        for (int t = 1; t < CP; ++t) {
            const float tv = fminf(fminf(cpv[t].x, cpv[t].y), fminf(cpv[t].z, cpv[t].w));
            if (t < tiles && tv < v) { v = tv; bt = t; win = cpv[t]; }
        }
        for (int t0 = CP; t0 < tiles; t0 += CP) {             // larger clouds: eight tiles in flight per step
            float4 w[CP];
#pragma unroll
            for (int t = 0; t < CP; ++t) w[t] = colpart[(size_t)(t0 + t < tiles ? t0 + t : 0) * m + kc];
#pragma unroll
            for (int t = 0; t < CP; ++t) {
                const float tv = fminf(fminf(w[t].x, w[t].y), fminf(w[t].z, w[t].w));
                if (t0 + t < tiles && tv < v) { v = tv; bt = t0 + t; win = w[t]; }
            }
        }
        const float *sx = rx + bt * CF_SEG, *sy = ry + bt * CF_SEG, *sz = rz + bt * CF_SEG;
        const float qv[4] = {win.x, win.y, win.z, win.w};
        int q1 = 3;                                           // lowest quarter attaining the minimum
#pragma unroll
        for (int q = 2; q >= 0; --q) q1 = qv[q] == v ? q : q1;
        int found = finish_quarter(sx, sy, sz, q1, qx, qy, qz, v);
        for (int q = q1 + 1; q < 4; ++q)                      // exact ties across quarters: rare
            if (qv[q] == v) found = min(found, finish_quarter(sx, sy, sz, q, qx, qy, qz, v));
        if (found == INT_MAX) found = 0;                      // only if v is NaN-tainted (out of contract)
        if (k < m) {
            pr.dist2[(size_t)c * m + k] = v;
            pr.idx2[(size_t)c * m + k] = bt * CS_ROWS + found;
        }
        if (rep + 1 < a.fin_reps) {                       // next sub-slice: its partial minima and coordinates
            k += CF_COLS;
            kc = k < m ? k : m - 1;
#pragma unroll
            for (int t = 0; t < CP; ++t) cpv[t] = colpart[(size_t)(t < tiles ? t : 0) * m + kc];
            qx = Q[3 * (size_t)kc]; qy = Q[3 * (size_t)kc + 1]; qz = Q[3 * (size_t)kc + 2];
        }
    }
    GA_STAMP(2, 7);
}

constexpr int CS_MAX_SPLIT = 4;
size_t chamfer_sym_workspace_floats(int pairs, int b, int n, int m) {
    return 4 * (size_t)pairs * b * cdiv(n, CS_ROWS) * m + 2 * (size_t)pairs * b * CS_MAX_SPLIT * n + 64;
}

// pairs: up to 2 problems with identical (n, m).  Requires n >= 1, m >= 1.
int launch_chamfer_sym_ex(const ChamferPair *pairs, int np, int b, int n, int m, float *workspace, int pair_base,
                          int q_clouds, const int *need1, hipStream_t stream, const GridArgs *rider = nullptr,
                          const JacRider *jac = nullptr);
int launch_chamfer_sym(const ChamferPair *pairs, int np, int b, int n, int m, float *workspace, hipStream_t stream) {
    return launch_chamfer_sym_ex(pairs, np, b, n, m, workspace, 0, 0, nullptr, stream);
}
// need1: per-cloud flags (int[8 * b], 16-byte aligned) restricting the SECOND pair to the clouds that still need the all-pairs kernel
int launch_chamfer_sym_needed(const ChamferPair *pairs, int np, int b, int n, int m, float *workspace, const int *need1,
                              hipStream_t stream) {
    return launch_chamfer_sym_ex(pairs, np, b, n, m, workspace, 0, 0, need1, stream);
}
// ... and with the paired grid search (rider->n <= GR_MAX_N) as 8 * b extra workgroups of the scan launch
// jac (or null): the pool Jacobian's 8 * b workgroups as well (jac->first_block / blocks are set here)
int launch_chamfer_sym_rider(const ChamferPair *pairs, int np, int b, int n, int m, float *workspace, const int *need1,
                             const GridArgs *rider, const JacRider *jac, hipStream_t stream) {
    return launch_chamfer_sym_ex(pairs, np, b, n, m, workspace, 0, 0, need1, stream, rider, jac);
}

int launch_chamfer_sym_ex(const ChamferPair *pairs, int np, int b, int n, int m, float *workspace, int pair_base,
                          int q_clouds, const int *need1, hipStream_t stream, const GridArgs *rider, const JacRider *jac) {
    if (b <= 0 || np <= 0) return GEOADV_OK;
    ChamferSymArgs a;
    a.need[0] = nullptr; a.need[1] = need1;
    a.pair_base = pair_base; a.q_clouds = q_clouds;
    for (int i = 0; i < np; ++i) a.pr[i] = pairs[i];
    a.n = n; a.m = m; a.tiles = cdiv(n, CS_ROWS); a.clouds = b; a.pairs = np; a.colpart = workspace;
    // column slices so that the grid fills the chip (4 workgroups per CU resident): 1, 2 or 4
    a.csplit = 1;
    // (a second pair gated by `need1` usually has no work at all -- the grid search answers it -- so it does not count)
    const int np_live = need1 ? 1 : np;
    while (a.csplit < CS_MAX_SPLIT && (long)a.tiles * a.csplit * b * np_live < 256 && m / (a.csplit * 2) >= 256) a.csplit *= 2;   // measured: slicing only pays when the grid would not even cover the CUs
    a.rowpart_d = workspace + 4 * (size_t)np * b * a.tiles * m;
    a.rowpart_i = reinterpret_cast<int *>(a.rowpart_d + (size_t)np * b * CS_MAX_SPLIT * n);
    static DeviceOnce attr;
    if (int rc = attr.run([]() -> int {
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chamfer_sym_finish_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chamfer_sym_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)std::max(CS_LDS_BYTES, chamfer_grid_lds_bytes(GR_MAX_N))));
            return GEOADV_OK;
        })) return rc;
    unsigned grid = (unsigned)(a.tiles * a.csplit * 8 * cdiv(b * np, 8));
    a.rider.blocks = 0; a.rider.first_block = 0;
    size_t scan_lds = CS_LDS_BYTES;
    if (rider) {
        a.rider.g = *rider;
        a.rider.first_block = 0;                           // dispatched first: its latency-bound workgroups start at once
        a.rider.blocks = b * 2 * GR_QSPLIT;
        grid += (unsigned)a.rider.blocks;
        scan_lds = std::max(scan_lds, chamfer_grid_lds_bytes(rider->n));
    }
    a.jac.blocks = 0; a.jac.first_block = 0;
    if (jac) {
        // LAST in the grid: every workgroup of the launch is charged the scan's 59 KB of LDS, so a CU holds two -- the search
        // and the scan from the start; the Jacobian's workgroups take the search's places as those finish (~13 us into a 30 us
        // scan).  Ahead of the scan they delayed it by their whole run time (51 instead of 33 us).
        a.jac = *jac;
        a.jac.first_block = (int)grid;
        a.jac.blocks = b * (128 / JAC_ROWS);
        grid += (unsigned)a.jac.blocks;
        scan_lds = std::max(scan_lds, JAC_LDS_BYTES);
    }
    chamfer_sym_kernel<<<grid, CS_THREADS, scan_lds, stream>>>(a);
    GA_LAUNCH_CHECK();
    const size_t lds = sizeof(float) * 3 * (size_t)a.tiles * CF_SEG;
    GA_REQUIRE(lds <= 150 * 1024, "chamfer_sym: too many rows (%d)", n);
    // large clouds stage ~100 KB of rows per workgroup: several column sub-slices per staging, as long as every CU still gets a workgroup
    a.fin_reps = a.tiles > 8 ? std::max(1, std::min(4, (int)((long)cdiv(m, CF_COLS) * b * np / kCUs))) : 1;
    chamfer_sym_finish_kernel<<<dim3(cdiv(m, CF_COLS * a.fin_reps), b * np), CF_THREADS, lds, stream>>>(a);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}


// out[pair] = mean(dist1[pair]) + mean(dist2[pair])  (chamfer_dist of prepare_indices_for_attack.py:113-114)
__global__ __launch_bounds__(256) void chamfer_pair_mean_kernel(int n, int m, const float *d1, const float *d2, float *out) {
    __shared__ float sh[8];
    const int c = blockIdx.x, t = threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
    for (int j = t; j < n; j += 256) s1 += d1[(size_t)c * n + j];
    for (int k = t; k < m; k += 256) s2 += d2[(size_t)c * m + k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
    if ((t & 63) == 0) { sh[t >> 6] = s1; sh[4 + (t >> 6)] = s2; }
    __syncthreads();
    if (t == 0) {
        const float a = ((sh[0] + sh[1]) + sh[2]) + sh[3], b = ((sh[4] + sh[5]) + sh[6]) + sh[7];
        out[c] = a * (1.0f / (float)n) + b * (1.0f / (float)m);
    }
}

}  // namespace geoadv

using namespace geoadv;

// Per-pair scratch of geoadv_chamfer_matrix, in floats.
static size_t matrix_floats_per_pair(int n, int m) {
    return (size_t)2 * (n + m) + chamfer_sym_workspace_floats(1, 1, n, m) + 64;
}

extern "C" size_t geoadv_chamfer_matrix_workspace_floats(int na, int nb, int n, int m) {
    if (na <= 0 || nb <= 0 || n <= 0 || m <= 0) return 256;
    const size_t pairs = (size_t)na * nb;
    const size_t chunk = pairs < 8192 ? pairs : 8192;
    return chunk * matrix_floats_per_pair(n, m) + 256;
}

extern "C" int geoadv_chamfer_matrix(int na, int nb, int n, int m, const float *A, const float *B, float *out,
                                     float *workspace, size_t workspace_floats, void *stream) {
    GA_REQUIRE(na >= 0 && nb >= 0 && n >= 1 && m >= 1, "chamfer_matrix: bad dimensions (na=%d nb=%d n=%d m=%d)", na, nb, n, m);
    if (na == 0 || nb == 0) return GEOADV_OK;
    GA_REQUIRE(A && B && out && workspace, "chamfer_matrix: null pointer");
    const size_t per = matrix_floats_per_pair(n, m);
    GA_REQUIRE(workspace_floats >= per + 256, "chamfer_matrix: workspace too small (%zu floats, need >= %zu)", workspace_floats, per + 256);
    size_t chunk = (workspace_floats - 256) / per;
    if (chunk > 32768) chunk = 32768;
    hipStream_t st = as_stream(stream);
    const size_t pairs = (size_t)na * nb;
    for (size_t base = 0; base < pairs; base += chunk) {
        const int cnt = (int)((pairs - base) < chunk ? (pairs - base) : chunk);
        float *d1 = workspace, *d2 = d1 + (size_t)cnt * n;
        int *i1 = reinterpret_cast<int *>(d2 + (size_t)cnt * m), *i2 = i1 + (size_t)cnt * n;
        float *ws = reinterpret_cast<float *>(i2 + (size_t)cnt * m);
        ws += (4 - ((size_t)cnt * 2 * (n + m)) % 4) % 4;       // the column partials are accessed as float4: keep them 16-byte aligned
                                                               // (odd cnt * (n + m); the + 64 floats of slack per pair cover it)
        const ChamferPair pr{A, B, d1, i1, d2, i2};
        GA_REQUIRE(base <= 0x7fffffff, "chamfer_matrix: too many pairs");
        if (int rc = launch_chamfer_sym_ex(&pr, 1, cnt, n, m, ws, (int)base, nb, nullptr, st)) return rc;
        chamfer_pair_mean_kernel<<<cnt, 256, 0, st>>>(n, m, d1, d2, out + base);
        GA_LAUNCH_CHECK();
    }
    return GEOADV_OK;
}
GA_STAMPS_GETTER(geoadv_debug_stamps_chamfer_sym)
