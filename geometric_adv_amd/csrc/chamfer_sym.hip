// Symmetric Chamfer for the attack loop: ONE evaluation of every pair distance serves both
// directions of nn_distance (note 4): d(P_j, Q_k) = ((dx*dx)+(dy*dy))+(dz*dz) is
// bit-identical whichever cloud is the "query" (the differences only change sign), so the row minima
// (dist1/idx1) and the column minima (dist2/idx2) of the same distance matrix are exactly what the two
// scans of the reference op produce (tf_nndistance.cpp:79-80).  Still brute force -- all n*m distances are
// evaluated, none is skipped -- but 8 VALU ops per pair are now spent once instead of twice.
//
// Round 5 layout: a workgroup owns a SLICE OF COLUMNS (Q points, <= 256, LDS-staged SoA planes, broadcast reads) against
// up to 2048 ROWS (P points): each of its 8 waves keeps its own 256 rows in registers (4 per lane) and walks the whole
// slice.  Every column's minimum over those rows is therefore complete INSIDE the workgroup:
//   * per round of 16 columns the 64 lanes of a wave are reduced through a 4 KB LDS transpose (16 ds_write_b32 + 4
//     ds_read_b128 per lane: ~8 % on top of the distance arithmetic) to four quarter minima per column (rows 64 r + 16 q + i
//     of the wave's 256), which stay in LDS (one ds_write_b32 per lane and round) -- until round 4 they went to HBM, 16 x the
//     op's algorithmic bytes, and came back in a second launch;
//   * after the scan the workgroup itself finds, per column, the minimum over its (wave, quarter) entries, the LOWEST
//     wave attaining it, and re-evaluates only the 64 rows of that wave's winning quarter for the lowest row with
//     d == minimum (exactly the reference's tie rule): 1/32 of the scan, two threads per column, rows out of LDS.
//     dist2 / idx2 leave the kernel final (clouds of more than 2048 rows: one (d, i) partial per 2048-row super-tile).
// Row minima: running min per chunk of 8 columns + a re-scan of the winning chunk, as before; a wave's rows meet only
// the workgroup's slice, so every (slice, row) leaves as a (distance, index) partial -- 8 B per row and slice instead
// of 16 B per column and row tile -- and the lexicographic minimum over the slices is taken by whoever reads them next:
// chamfer_sym_merge_kernel (the operator form), or the attack loop's loss / gradient launch on its way in
// (attack.hip: no second Chamfer launch in the iteration at all).
// Clouds of <= 1024 rows put several waves on the same rows (rw row-waves x cw column-waves = 8).
#include "common.h"
#include "chamfer_grid.h"
#include "encoder_jac.h"
#include "chamfer_sym.h"
#include "chamfer_mx.h"
#include "loss_cgrad.h"
#include <atomic>
#include <limits.h>
#include <math.h>
#include <stdlib.h>

#pragma clang fp contract(off)

namespace geoadv {

struct ChamferSymArgs {
    ChamferPair pr[2];
    int n, m, clouds, pairs;
    int rw, cw;                // row-waves x column-waves of a workgroup (rw * cw == 8)
    int rtiles;                // row super-tiles of 256 * rw rows
    int C, S, cslices;         // columns per stage (multiple of 16), stages per workgroup, column slices = cdiv(m, C * S)
    int pair_base, q_clouds;   // q_clouds > 0: cloud c is the pair (P cloud (pair_base+c)/q_clouds, Q cloud (pair_base+c)%q_clouds)
    float *rowpart_d;          // [pairs][clouds][cslices * cw][n]  row minima per column slice (absent when there is one slice)
    int *rowpart_i;
    float *colpart_d;          // [pairs][clouds][rtiles][m]        column minima per row super-tile (rtiles > 1 only)
    int *colpart_i;
    unsigned long long *row64; // [pairs][clouds][n] packed row minima folded with 64-bit atomic minima (or null: the partials above)
    const int *need[2];        // per pair: null = every cloud; else int[8 * clouds], cloud c is computed only if one of
                               // its 8 flags is set (a workgroup of the paired grid search, chamfer_grid.hip, gave up)
    GridRider rider;           // the attack loop: the paired grid search of (adv, source) as extra workgroups of the scan launch
    JacRider jac;              // ... and the encoder's pool Jacobian (encoder_jac.h): needed by the NEXT step's backward only
    LossRider loss;            // ... and the loop's loss + gradient workgroups (loss_cgrad.h), which wait for their cloud's scan and search
};

constexpr int CS_THREADS = 512;               // 8 waves
constexpr int CS_WAVES = 8;
constexpr int CS_R = 4;                       // rows per lane
constexpr int CS_WROWS = kWave * CS_R;        // 256 rows per wave
constexpr int CS_CHUNK = 8;                   // columns per arg-min chunk
constexpr int CS_ROUND = 16;                  // columns per transpose round
constexpr int CS_CMAX = 256;                  // columns per workgroup at most
constexpr int CS_CMIN = 64;
constexpr int CS_MAX_STAGES = 4;
constexpr int CS_TSTRIDE = 68;                // floats per column in the transpose buffer (64 lanes + pad)
constexpr int CS_SEG = CS_WROWS + 4;          // floats per wave segment of the rows staged for the index search: 16-B aligned, staggers the banks

__device__ __forceinline__ float sqdist_s(float tx, float ty, float tz, float qx, float qy, float qz) {
    const float dx = tx - qx, dy = ty - qy, dz = tz - qz;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    return (xx + yy) + zz;
}

// stage planes | quarter minima [8 waves][CS_CMAX][4] | transpose buffers (later: the workgroup's rows, three planes of 8 segments)
constexpr size_t CS_LDS_FLOATS = 3 * CS_CMAX + 4 * CS_WAVES * CS_CMAX + CS_WAVES * CS_ROUND * CS_TSTRIDE;
constexpr size_t CS_LDS_BYTES = sizeof(float) * CS_LDS_FLOATS;
static_assert(3 * CS_WAVES * CS_SEG <= CS_WAVES * CS_ROUND * CS_TSTRIDE, "the staged rows reuse the transpose buffers");

// lowest row among {64 r + 16 q + i : r in {2 h, 2 h + 1}} of the wave segment staged at (sx, sy, sz) whose distance to the
// column equals v, or INT_MAX
__device__ __forceinline__ int sym_find_half(const float *sx, const float *sy, const float *sz, int q, int h, float qx, float qy, float qz, float v) {
    int found = INT_MAX;
#pragma unroll
    for (int rr = 1; rr >= 0; --rr)                       // descending: the last hit kept is the lowest row
#pragma unroll
        for (int j4 = 3; j4 >= 0; --j4) {
            const int o = 64 * (2 * h + rr) + 16 * q + 4 * j4;
            const float4 xa = *reinterpret_cast<const float4 *>(sx + o);
            const float4 ya = *reinterpret_cast<const float4 *>(sy + o);
            const float4 za = *reinterpret_cast<const float4 *>(sz + o);
            const float d3 = sqdist_s(qx, qy, qz, xa.w, ya.w, za.w), d2 = sqdist_s(qx, qy, qz, xa.z, ya.z, za.z);
            const float d1 = sqdist_s(qx, qy, qz, xa.y, ya.y, za.y), d0 = sqdist_s(qx, qy, qz, xa.x, ya.x, za.x);
            found = d3 == v ? o + 3 : found;
            found = d2 == v ? o + 2 : found;
            found = d1 == v ? o + 1 : found;
            found = d0 == v ? o : found;
        }
    return found;
}

__global__ __launch_bounds__(CS_THREADS, 4) void chamfer_sym_kernel(ChamferSymArgs a) {
    constexpr int R = CS_R;
    extern __shared__ __attribute__((aligned(16))) float stage[];
    if (grid_rider_block<GR_MAX_N>(a.rider)) return;
    if (jac_rider_block(a.jac, stage)) return;
    if (loss_rider_block(a.loss, reinterpret_cast<unsigned *>(stage))) return;
    GA_STAMP(0, 0);
    const int lin = blockIdx.x - a.rider.blocks;           // (the search's workgroups come first); XCD-aware mapping, see chamfer_scan_kernel
    const int xcd = lin & 7, slot = lin >> 3;
    const int per = a.rtiles * a.cslices;                  // workgroups per (pair, cloud) group
    const int group = (slot / per) * 8 + xcd, sub = slot % per;
    if (group >= a.clouds * a.pairs) return;
    const int rt = sub % a.rtiles, cs = sub / a.rtiles;    // row super-tile, column slice
    const int pi = group / a.clouds, c = group % a.clouds;
    if (!sym_needed(a.need[pi], c)) {                      // (a gated-off cloud still counts in: the loss riders expect every workgroup)
        if (a.loss.blocks) loss_rider_arrive(a.loss.done, c);
        return;
    }
    const ChamferPair pr = a.pr[pi];
    const int n = a.n, m = a.m;
    const int cp = a.q_clouds > 0 ? (a.pair_base + c) / a.q_clouds : c;
    const int cq = a.q_clouds > 0 ? (a.pair_base + c) % a.q_clouds : c;
    const float *P = pr.p + (size_t)cp * n * 3;
    const float *Q = pr.q + (size_t)cq * m * 3;

    // all of it in the DYNAMIC region (CS_LDS_BYTES, or a rider's need if larger): a launch that hosts the grid search's
    // workgroups is charged max(scan, search) of LDS per workgroup, not the sum
    float *sx = stage, *sy = stage + CS_CMAX, *sz = stage + 2 * CS_CMAX;
    unsigned *colq = reinterpret_cast<unsigned *>(stage + 3 * CS_CMAX);                   // [rw][CS_CMAX][4]
    float *tbase = stage + 3 * CS_CMAX + 4 * CS_WAVES * CS_CMAX;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rwi = wave % a.rw, cwi = wave / a.rw;        // this wave's row block inside the workgroup, its share of the slice
    const int q0 = (rt * a.rw + rwi) * CS_WROWS;
    float px[R], py[R], pz[R], best[R];
    int bestk[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int j = q0 + r * kWave + lane;
        j = j < n ? j : n - 1;                             // padding rows repeat the last row: same distances,
        px[r] = P[3 * j]; py[r] = P[3 * j + 1]; pz[r] = P[3 * j + 2];   // so the column minima are unaffected
        best[r] = INFINITY; bestk[r] = -1;
    }
    float *tb = tbase + wave * (CS_ROUND * CS_TSTRIDE);
    unsigned *cq4 = colq + (size_t)rwi * CS_CMAX * 4;
    float *rx = tbase, *ry = tbase + CS_WAVES * CS_SEG, *rz = tbase + 2 * CS_WAVES * CS_SEG;   // the rows, staged for the index search
    int found[R];
#pragma unroll
    for (int r = 0; r < R; ++r) found[r] = INT_MAX;
    // A workgroup walks a.S column stages of a.C columns each with the same rows in registers (large clouds: the row loads, the
    // row partials and the launch's workgroup count shrink by S; the headline shape has S = 1).  Columns staged for stage st + 1
    // are REQUESTED before stage st's index search, so that their global round trip hides behind it.
    float nx = INFINITY, ny = INFINITY, nz = INFINITY;     // this thread's column of the next stage (threads < C)
    auto request = [&](int st) {
        nx = ny = nz = INFINITY;
        const int k = (cs * a.S + st) * a.C + (int)threadIdx.x;
        if (st < a.S && (int)threadIdx.x < a.C && k < m) { nx = Q[3 * (size_t)k]; ny = Q[3 * (size_t)k + 1]; nz = Q[3 * (size_t)k + 2]; }
    };
    request(0);
    for (int st = 0; st < a.S; ++st) {
        const int cbeg = (cs * a.S + st) * a.C;
        if (cbeg >= m) break;                              // (uniform)
        const int cnt = min(a.C, m - cbeg);
        const int cntp = (cnt + CS_ROUND - 1) / CS_ROUND * CS_ROUND;
        if ((int)threadIdx.x < cntp) { sx[threadIdx.x] = nx; sy[threadIdx.x] = ny; sz[threadIdx.x] = nz; }   // (padding columns: +inf)
        __syncthreads();
        const int nrounds = cntp / CS_ROUND;
        const int rbeg = nrounds * cwi / a.cw, rend = nrounds * (cwi + 1) / a.cw;
        float prev[R];
#pragma unroll
        for (int r = 0; r < R; ++r) prev[r] = best[r];
        if (st == 0 && rbeg < rend) {
#pragma unroll
            for (int r = 0; r < R; ++r) bestk[r] = rbeg * CS_ROUND;
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
                    if (cm < best[r]) { best[r] = cm; bestk[r] = k0 + hf * CS_CHUNK; }
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
                // {64 r + 16 quarter + i} of this wave: kept in LDS (consecutive lanes, consecutive words) for the index search below
                cq4[(k0 + col) * 4 + quarter] = mb;
            }
            __builtin_amdgcn_wave_barrier();              // the buffer is rewritten by the next round
        }
        GA_STAMP(0, 1);
        // row minima: first index attaining the minimum inside the winning chunk, for the rows whose minimum moved in THIS stage
        // (its columns are still staged) -- branch-free: the chunk's eight columns come back as two ds_read_b128 per plane (chunks
        // start at multiples of 8 inside the stage; columns beyond the slice are staged as +inf and can never equal a finite
        // minimum), and the hits are taken in DESCENDING order so that the last one kept is the lowest index
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if ((st == 0 && rbeg < rend) || best[r] < prev[r]) {   // (first stage: also rows whose minimum stayed +inf / NaN-tainted)
                const int kb = bestk[r];
                const float4 xa = *reinterpret_cast<const float4 *>(&sx[kb]), xb = *reinterpret_cast<const float4 *>(&sx[kb + 4]);
                const float4 ya = *reinterpret_cast<const float4 *>(&sy[kb]), yb = *reinterpret_cast<const float4 *>(&sy[kb + 4]);
                const float4 za = *reinterpret_cast<const float4 *>(&sz[kb]), zb = *reinterpret_cast<const float4 *>(&sz[kb + 4]);
                const float tx[CS_CHUNK] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
                const float ty[CS_CHUNK] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
                const float tz[CS_CHUNK] = {za.x, za.y, za.z, za.w, zb.x, zb.y, zb.z, zb.w};
                int f = kb;
#pragma unroll
                for (int u = CS_CHUNK - 1; u >= 0; --u)
                    f = sqdist_s(tx[u], ty[u], tz[u], px[r], py[r], pz[r]) == best[r] ? kb + u : f;
                found[r] = f + cbeg;
            }
        }
        request(st + 1);                                   // the next stage's columns: in flight during the index search below
        __syncthreads();                                   // quarter minima complete; the transpose buffers are free
        // the workgroup's rows into LDS, one padded segment per row-wave
        if (cwi == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                rx[rwi * CS_SEG + r * kWave + lane] = px[r];
                ry[rwi * CS_SEG + r * kWave + lane] = py[r];
                rz[rwi * CS_SEG + r * kWave + lane] = pz[r];
            }
        }
        __syncthreads();
        GA_STAMP(0, 2);
        // column minima: two threads per column.  Minimum over the (wave, quarter) entries and the LOWEST wave attaining it
        // (strict compare); several quarters of that wave attaining it (exact ties between rows 16 apart or more) are rare:
        // those columns walk their further quarters in a loop the other lanes sit out.
        {
            const int col = threadIdx.x >> 1, h = threadIdx.x & 1;
            const int cc = col < cnt ? col : 0;
            const float4 *cq = reinterpret_cast<const float4 *>(colq);
            float4 cpv[CS_WAVES];                                // (all eight reads in flight; absent row-waves repeat wave 0)
#pragma unroll
            for (int w = 0; w < CS_WAVES; ++w) cpv[w] = cq[(w < a.rw ? w : 0) * CS_CMAX + cc];
            float4 win = cpv[0];
            float v = fminf(fminf(win.x, win.y), fminf(win.z, win.w));
            int bt = 0;
#pragma unroll
            for (int w = 1; w < CS_WAVES; ++w) {
                const float tv = fminf(fminf(cpv[w].x, cpv[w].y), fminf(cpv[w].z, cpv[w].w));
                if (w < a.rw && tv < v) { v = tv; bt = w; win = cpv[w]; }
            }
            const float qx = sx[cc], qy = sy[cc], qz = sz[cc];
            const float *wx = rx + bt * CS_SEG, *wy = ry + bt * CS_SEG, *wz = rz + bt * CS_SEG;
            const float qv[4] = {win.x, win.y, win.z, win.w};
            int q1 = 3;                                           // lowest quarter attaining the minimum
#pragma unroll
            for (int q = 2; q >= 0; --q) q1 = qv[q] == v ? q : q1;
            int fnd = sym_find_half(wx, wy, wz, q1, h, qx, qy, qz, v);
            for (int q = q1 + 1; q < 4; ++q)                      // exact ties across quarters: rare
                if (qv[q] == v) fnd = min(fnd, sym_find_half(wx, wy, wz, q, h, qx, qy, qz, v));
            fnd = min(fnd, __builtin_amdgcn_update_dpp(INT_MAX, fnd, 0xB1, 0xf, 0xf, false));   // the pair's other half (quad_perm [1,0,3,2])
            if (fnd == INT_MAX) fnd = 0;                          // only if v is NaN-tainted (out of contract)
            if (col < cnt) {
                const int k = cbeg + col;
                const int row = (rt * a.rw + bt) * CS_WROWS + fnd;
                if (a.rtiles == 1) {
                    if (h == 0) pr.dist2[(size_t)c * m + k] = v; else pr.idx2[(size_t)c * m + k] = row;
                } else {
                    const size_t o = (((size_t)pi * a.clouds + c) * a.rtiles + rt) * m + k;
                    if (h == 0) a.colpart_d[o] = v; else a.colpart_i[o] = row;
                }
            }
        }
        __syncthreads();                                   // the stage planes, the quarter minima and the staged rows are free again
    }
    // row minima of this workgroup's columns: final (one slice), a packed word folded at the memory side, or a partial per slice
    const int rslices = a.cslices * a.cw;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int j = q0 + r * kWave + lane;
        if (j < n) {
            if (rslices == 1) {
                pr.dist1[(size_t)c * n + j] = best[r];
                pr.idx1[(size_t)c * n + j] = found[r];
            } else if (a.row64) {   // narrow slices: one packed word per row, folded at the memory side (no return value: fire and forget)
                atomicMin(&a.row64[((size_t)pi * a.clouds + c) * n + j], sym_pack(best[r], found[r]));
            } else {   // lexicographic (distance, index) minimum over the slices: chamfer_sym_merge_kernel or the loop's loss launch
                const size_t o = (((size_t)pi * a.clouds + c) * rslices + (cs * a.cw + cwi)) * n + j;
                a.rowpart_d[o] = best[r];
                a.rowpart_i[o] = found[r];
            }
        }
    }
    if (a.loss.blocks) loss_rider_arrive(a.loss.done, c);
    GA_STAMP(0, 7);
}

// The matrix-pipe-screened form of the same scan (chamfer_mx.h): same grid mapping, same riders, same three ways out for the row
// minima (final / packed word folded by a 64-bit atomic minimum / one partial per column slice) and two for the column minima.
template <int NCT>
__global__ __launch_bounds__(MX_THREADS, 2) void chamfer_mx_kernel(ChamferSymArgs a) {
    extern __shared__ __attribute__((aligned(16))) float stage[];
    if (grid_rider_block<GR_MAX_N>(a.rider)) return;
    if (jac_rider_block<true>(a.jac, stage)) return;       // (this kernel's register budget admits the Jacobian's faster form)
    GA_STAMP(0, 0);
    const int lin = blockIdx.x - a.rider.blocks;
    const int xcd = lin & 7, slot = lin >> 3;
    const int per = a.rtiles * a.cslices;
    const int group = (slot / per) * 8 + xcd, sub = slot % per;
    if (group >= a.clouds * a.pairs) return;
    const int rt = sub % a.rtiles, cs = sub / a.rtiles;
    const int pi = group / a.clouds, c = group % a.clouds;
    if (!sym_needed(a.need[pi], c)) return;
    const ChamferPair pr = a.pr[pi];
    const int n = a.n, m = a.m;
    const int cp = a.q_clouds > 0 ? (a.pair_base + c) / a.q_clouds : c;
    const int cq = a.q_clouds > 0 ? (a.pair_base + c) % a.q_clouds : c;
    MxView v;
    v.P = pr.p + (size_t)cp * n * 3; v.Q = pr.q + (size_t)cq * m * 3;
    v.n = n; v.m = m; v.rt = rt; v.cs = cs; v.C = a.C; v.S = a.S;
    const int rslices = a.cslices;
    // The results leave through buffer stores: a uniform base (this workgroup's row of the output or of the partials: scalar
    // registers) and a 32-bit lane offset.  As plain pointer stores the four per-lane 64-bit addresses were hoisted out of the
    // stage loop and, at 256 registers, spilled -- the only scratch segment among the loop's kernels.
    float *row_d = rslices == 1 ? pr.dist1 + (size_t)c * n : a.rowpart_d + (((size_t)pi * a.clouds + c) * rslices + cs) * n;
    int *row_i = rslices == 1 ? pr.idx1 + (size_t)c * n : a.rowpart_i + (((size_t)pi * a.clouds + c) * rslices + cs) * n;
    float *col_d = a.rtiles == 1 ? pr.dist2 + (size_t)c * m : a.colpart_d + (((size_t)pi * a.clouds + c) * a.rtiles + rt) * m;
    int *col_i = a.rtiles == 1 ? pr.idx2 + (size_t)c * m : a.colpart_i + (((size_t)pi * a.clouds + c) * a.rtiles + rt) * m;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(row_d, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(row_i, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t cd = __builtin_amdgcn_make_buffer_rsrc(col_d, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ci = __builtin_amdgcn_make_buffer_rsrc(col_i, 0, 0x7fffffff, 0x00020000);
    const bool row_atomic = rslices != 1 && a.row64 != nullptr;
    // (j / k made opaque where they are used: the byte offsets are loop-invariant per lane and would be hoisted -- and spilled -- too)
    auto out_row = [&](int j, float d, int i) {
        asm volatile("" : "+v"(j));
        if (row_atomic) atomicMin(&a.row64[((size_t)pi * a.clouds + c) * n + j], sym_pack(d, i));
        else {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d), rd, (unsigned)j * 4u, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)i, ri, (unsigned)j * 4u, 0, 0);
        }
    };
    auto out_col = [&](int k, float d, int i) {
        asm volatile("" : "+v"(k));
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d), cd, (unsigned)k * 4u, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32((unsigned)i, ci, (unsigned)k * 4u, 0, 0);
    };
#ifdef GA_STAMPS
    unsigned long long clk0_, clk1_;                               // diagnostic build: the shader clock over the scan (s_memtime counts core cycles)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk0_)::"memory");
#endif
    mx_scan_block<NCT>(v, reinterpret_cast<unsigned *>(stage), out_row, out_col);
#ifdef GA_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk1_)::"memory");
    if (threadIdx.x == 0 && blockIdx.x < GA_STAMP_BLOCKS) ga_stamps[(2 * GA_STAMP_BLOCKS + blockIdx.x) * 8 + 1] = clk1_ - clk0_;
    GA_STAMP(0, 6);
    __syncthreads();                                               // (diagnostic build: stamp 7 = the workgroup's last wave)
#endif
    GA_STAMP(0, 7);
}

// The operator form's second launch: lexicographic (distance, index) minimum over the row partials of the column slices
// and, for clouds of several row super-tiles, over the column partials.  grid = (cdiv(max(n, m), 256), clouds * pairs).
__global__ __launch_bounds__(256) void chamfer_sym_merge_kernel(ChamferSymArgs a) {
    const int group = blockIdx.y;
    const int pi = group / a.clouds, c = group % a.clouds;
    if (!sym_needed(a.need[pi], c)) return;
    const ChamferPair pr = a.pr[pi];
    const int n = a.n, m = a.m, e = blockIdx.x * 256 + threadIdx.x;
    const int rslices = a.cslices * a.cw;
    if (rslices > 1 && e < n) {
        const size_t o = ((size_t)pi * a.clouds + c) * rslices * n + e;
        float d; int i;
        sym_merge_slices(a.rowpart_d + o, a.rowpart_i + o, rslices, n, d, i);
        pr.dist1[(size_t)c * n + e] = d;
        pr.idx1[(size_t)c * n + e] = i;
    }
    if (a.rtiles > 1 && e < m) {
        const size_t o = ((size_t)pi * a.clouds + c) * a.rtiles * m + e;
        float d; int i;
        sym_merge_slices(a.colpart_d + o, a.colpart_i + o, a.rtiles, m, d, i);
        pr.dist2[(size_t)c * m + e] = d;
        pr.idx2[(size_t)c * m + e] = i;
    }
}

// Launch shape of the symmetric scan: how the 8 waves of a workgroup are laid over rows and columns, and how many
// columns a workgroup takes so that the grid fills the chip (256 / 128 / 64; `groups` = live (pair, cloud) groups).
struct SymShape { int rw, cw, rtiles, C, S, cslices; bool mx; };
// process-wide switch (geoadv_set_chamfer_screen): 0 = the unscreened scan everywhere -- the parity tests' second opinion and the
// A/B tools.  Scratch sizes never depend on it (they cover both kernels), so it may change between calls.
static std::atomic<int> g_mx_on{1};
static bool mx_enabled() { return g_mx_on.load() != 0; }
// loop: the attack loop's launch (riders may share it).  The screened kernel holds ONE workgroup per CU (256 registers a wave,
// 100 KB of LDS): the loop's riders -- latency-bound workgroups that hide under the unscreened scan at two workgroups per CU --
// would queue behind it, and its own serial phases (operands, column finish, uncertified queries: ~9 of ~24 us per 2048 x 256
// tile) are only amortised when a CU gets several tiles.  Measured (tools/debug/mx_check.py, us per operator call, alternating in
// one process): 32 x 2048^2 33.5 against 37.8 unscreened, 64 x 2048^2 55.0 / 60.7, 128 x 2048^2 87.0 / 108.2, 32 x 8192^2
// 272 / 383; with 128-column slices (16 x 2048^2) 29.2 / 27.0; inside the B = 32 all-pairs loop with the Jacobian rider behind it
// (two tiles per CU, tools/debug/mx_loop_ab.py) 156.7 us per iteration against 151.4.  So: operators from one 256-column tile
// per CU on, the loop from four.
static SymShape sym_shape(long groups, int n, int m, bool loop, bool screen) {
    SymShape s;
    s.mx = false;
    if (screen && n > 1024 && (long)cdiv(n, MX_ROWS) * cdiv(m, MX_CMAX) * groups >= (loop ? 4 : 1) * (long)kCUs) {
        // the matrix-pipe-screened kernel (chamfer_mx.h): eight row-waves always, a slice of 256 / 128 / 64 columns per stage so that
        // the grid covers the chip (same rule as below), several stages per workgroup while two workgroups per CU remain (one is
        // resident at a time: 256 registers a wave)
        s.mx = true; s.rw = MX_WAVES; s.cw = 1;
        s.rtiles = cdiv(n, MX_ROWS);
        s.C = MX_CMAX;
        while (s.C > 64 && (long)s.rtiles * cdiv(m, s.C) * groups < kCUs &&
               (s.C > 128 || (long)s.rtiles * cdiv(m, s.C / 2) * groups <= kCUs)) s.C /= 2;
        // column stages per workgroup: ONE workgroup is resident per CU, so the launch takes rounds(S) x (operands + S x stage) with
        // rounds = ceil(workgroups / CUs) -- ~6 + 18 S us per workgroup at 256-column stages.  More stages amortise the
        // operands and save whole rounds (512 tiles: two rounds of one stage 48, one round of two stages 42; 1600 tiles: 168 / 156 /
        // 150 at 2 / 4 / 8), as long as the last round is not mostly idle -- which the same product prices.
        s.S = 1;
        {
            const int ctiles = cdiv(m, s.C);
            long best = -1;
            for (int S = 1; S <= MX_MAX_STAGES && S <= std::max(1, ctiles); S *= 2) {
                const long wgs = (long)s.rtiles * cdiv(ctiles, S) * groups;
                const long cost = ((wgs + kCUs - 1) / kCUs) * (6 + 18L * S);
                if (best < 0 || cost < best) { best = cost; s.S = S; }
            }
        }
        s.cslices = cdiv(m, s.C * s.S);
        return s;
    }
    s.rw = n > 1024 ? 8 : n > 512 ? 4 : n > 256 ? 2 : 1;
    s.cw = CS_WAVES / s.rw;
    s.rtiles = cdiv(n, CS_WROWS * s.rw);
    const int cmin = std::max(CS_CMIN, 2 * CS_ROUND * s.cw);      // every column-wave keeps at least two rounds
    // columns per workgroup: halved while the grid does not cover the CUs (measured: slicing only pays then) -- but not below 128
    // columns into a grid that no longer fits ONE workgroup per CU: 10 ... 14 clouds got 320 ... 448 workgroups of 64 columns, a
    // second layer on a quarter of the CUs of a VALU-bound kernel (scan 28.4 us at B = 10 and 12 against 24.0 at B = 16; with 128
    // columns 0.0917 -> 0.0877, 0.0971 -> 0.0894, 0.0976 -> 0.0915 ms per iteration at B = 10 / 12 / 14).  Wider slices than that
    // keep the old rule: 320 workgroups of 128 columns beat 160 of 256 (B = 20: 0.1178 against 0.1268).
    s.C = CS_CMAX;
    while (s.C > cmin && (long)s.rtiles * cdiv(m, s.C) * groups < kCUs &&
           (s.C > 128 || (long)s.rtiles * cdiv(m, s.C / 2) * groups <= kCUs)) s.C /= 2;
    // large launches: several column stages per workgroup (same rows in registers, next stage's columns requested a stage ahead) as
    // long as four workgroups per CU remain.  Only where ONE column-wave walks the slice: with cw > 1 every stage is re-split over
    // the column-waves, a wave's partial then covers columns that are not one ascending range, and the merge's tie rule (lowest
    // slice = lowest index, chamfer_sym.h) would no longer hold (tests/test_gpu_chamfer_shapes.py duplicates cases).
    s.S = 1;
    while (s.cw == 1 && s.S < CS_MAX_STAGES && (long)s.rtiles * cdiv(m, s.C * s.S * 2) * groups >= 4 * kCUs) s.S *= 2;
    s.cslices = cdiv(m, s.C * s.S);
    return s;
}
static size_t sym_group_floats(const SymShape &s, int n, int m) {
    return 2 * (size_t)s.cslices * s.cw * n + (s.rtiles > 1 ? 2 * (size_t)s.rtiles * m : 0);
}

// Scratch for `pairs` problems of `b` clouds each, in floats: an upper bound over the shapes the launcher may choose for
// any number of live groups <= pairs * b (fewer live groups = narrower slices = more row partials per group).
size_t chamfer_sym_workspace_floats(int pairs, int b, int n, int m) {
    size_t per = 0;
    for (int v = 0; v < 4; ++v) {                            // (either launch rule, screened or not: the switch may change later)
        for (long g = 1; g <= (long)pairs * b; g *= 2) per = std::max(per, sym_group_floats(sym_shape(g, n, m, (v & 1) != 0, (v & 2) != 0), n, m));
        per = std::max(per, sym_group_floats(sym_shape((long)pairs * b, n, m, (v & 1) != 0, (v & 2) != 0), n, m));
    }
    return (size_t)pairs * b * per + 64;
}

// Will launch_chamfer_sym_loop fold the row minima into the caller's packed words (SymPartials::row64) at this shape?  (The caller
// fills them with all ones beforehand only then.)  live_groups: clouds x problems that are not gated off.
bool chamfer_sym_packs_rows(long live_groups, int n, int m) {
    const SymShape s = sym_shape(live_groups, n, m, true, mx_enabled());
    return s.rtiles == 1 && s.cslices * s.cw > 8;
}

// pairs: up to 2 problems with identical (n, m).  Requires n >= 1, m >= 1.
int launch_chamfer_sym_ex(const ChamferPair *pairs, int np, int b, int n, int m, float *workspace, int pair_base,
                          int q_clouds, const int *need1, hipStream_t stream, const GridArgs *rider = nullptr,
                          const JacRider *jac = nullptr, SymPartials *defer = nullptr, bool loop = false, LossRider *loss = nullptr);
int launch_chamfer_sym(const ChamferPair *pairs, int np, int b, int n, int m, float *workspace, hipStream_t stream) {
    return launch_chamfer_sym_ex(pairs, np, b, n, m, workspace, 0, 0, nullptr, stream);
}
// need1: per-cloud flags (int[8 * b], 16-byte aligned) restricting the SECOND pair to the clouds that still need the all-pairs kernel
// rider (rider->n <= GR_MAX_N): the paired grid search as 8 * b extra workgroups of the scan launch; jac (or null): the pool
// Jacobian's 8 * b workgroups as well (jac->first_block / blocks are set here); defer (or null): see SymPartials
int launch_chamfer_sym_loop(const ChamferPair *pairs, int np, int b, int n, int m, float *workspace, const int *need1,
                            const GridArgs *rider, const JacRider *jac, SymPartials *defer, hipStream_t stream, LossRider *loss) {
    return launch_chamfer_sym_ex(pairs, np, b, n, m, workspace, 0, 0, need1, stream, rider, jac, defer, true, loss);
}
bool chamfer_sym_hosts_loss(long live_groups, int b, int n, int m) {
    const SymShape s = sym_shape(live_groups, n, m, true, mx_enabled());
    return !s.mx && s.rtiles == 1 && b % 8 == 0;
}

int launch_chamfer_sym_ex(const ChamferPair *pairs, int np, int b, int n, int m, float *workspace, int pair_base,
                          int q_clouds, const int *need1, hipStream_t stream, const GridArgs *rider, const JacRider *jac,
                          SymPartials *defer, bool loop, LossRider *loss) {
    if (loss) loss->blocks = 0;
    unsigned long long *row64 = defer ? defer->row64 : nullptr;
    if (defer) { defer->slices = 1; defer->rowpart_d = nullptr; defer->rowpart_i = nullptr; defer->clouds = b; defer->deferred = false; defer->row64 = nullptr; }
    if (b <= 0 || np <= 0) return GEOADV_OK;
    ChamferSymArgs a;
    a.need[0] = nullptr; a.need[1] = need1;
    a.pair_base = pair_base; a.q_clouds = q_clouds;
    for (int i = 0; i < np; ++i) a.pr[i] = pairs[i];
    a.n = n; a.m = m; a.clouds = b; a.pairs = np;
    // (a second pair gated by `need1` usually has no work at all -- the grid search answers it -- so it does not count)
    const int np_live = need1 ? 1 : np;
    const SymShape s = sym_shape((long)b * np_live, n, m, loop, mx_enabled());
    a.rw = s.rw; a.cw = s.cw; a.rtiles = s.rtiles; a.C = s.C; a.S = s.S; a.cslices = s.cslices;
    const int rslices = s.cslices * s.cw;
    const size_t groups = (size_t)np * b;
    a.rowpart_d = workspace;
    a.rowpart_i = reinterpret_cast<int *>(workspace + groups * rslices * n);
    a.colpart_d = workspace + 2 * groups * rslices * n;
    a.colpart_i = reinterpret_cast<int *>(a.colpart_d + groups * s.rtiles * m);
    // more than 8 row partials per row (narrow slices: small batches) and a caller that reads packed words: atomic form
    a.row64 = (defer && row64 && s.rtiles == 1 && rslices > 8) ? row64 : nullptr;
    static DeviceOnce attr;
    if (int rc = attr.run([]() -> int {
            const int need = (int)std::max(std::max(std::max(CS_LDS_BYTES, MX_LDS_BYTES), chamfer_grid_lds_bytes(GR_MAX_N)), JAC_LDS_BYTES);
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chamfer_sym_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, need));
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chamfer_mx_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, need));
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chamfer_mx_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, need));
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chamfer_mx_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, need));
            return GEOADV_OK;
        })) return rc;
    unsigned grid = (unsigned)(s.rtiles * s.cslices * 8 * cdiv(b * np, 8));
    a.rider.blocks = 0; a.rider.first_block = 0; a.rider.clouds = 0; a.rider.done = nullptr;
    a.loss.blocks = 0; a.loss.first_block = 0; a.loss.done = nullptr;
    size_t scan_lds = s.mx ? MX_LDS_BYTES : CS_LDS_BYTES;
    // ... which pays in a narrow band only (tools/debug/loss_in_scan_ab.py, us per iteration riding / own launch, trajectories
    // bit-identical everywhere): with the search's workgroups in the launch and two or more scan workgroups per CU, the riders of
    // clouds whose slices ran first work under the later ones -- B = 64: 229.2 / 241.0, B = 96: 318.5 / 325.4 (B = 128: 428.2 /
    // 427.2).  With one scan workgroup per CU every cloud's slowest slice ends with the launch: the riders start where a launch of
    // their own would, and the producers' drained stores + the extra residents cost more than the boundary returns (B = 32: 134.1
    // / 129.7, B = 16: 94.1 / 88.3; timeline in DESIGN 6); without the search (all-pairs) the scan's own workgroups hold every
    // slot to the end (B = 32 ... 128: +- 0.6).
    const bool host_loss = loss && !s.mx && s.rtiles == 1 && b % 8 == 0 && q_clouds == 0 &&
                           (loss->force || (rider != nullptr && (long)s.rtiles * s.cslices * b * np_live >= 2 * (long)kCUs));
    if (rider) {
        a.rider.g = *rider;
        a.rider.first_block = 0;                           // dispatched first: its latency-bound workgroups start at once
        a.rider.blocks = grid_rider_blocks(b);
        a.rider.clouds = b;
        if (host_loss) a.rider.done = loss->done;
        grid += (unsigned)a.rider.blocks;
        scan_lds = std::max(scan_lds, chamfer_grid_lds_bytes(rider->n));
    }
    a.jac.blocks = 0; a.jac.first_block = 0; a.jac.raise_prio = 0;
    if (jac) {
        // LAST in the grid: every workgroup of the launch is charged the scan's 70 KB of LDS, so a CU holds two -- the search
        // and the scan from the start; the Jacobian's workgroups take the search's places as those finish (~13 us into a 30 us
        // scan).  Ahead of the scan they delayed it by their whole run time (51 instead of 33 us).
        a.jac = *jac;
        a.jac.first_block = (int)grid;
        a.jac.blocks = b * (128 / JAC_ROWS);
        a.jac.raise_prio = a.jac.blocks * 2 <= kCUs ? 1 : 0;     // fewer riders than half the CUs (B <= 16): see JacRider
        grid += (unsigned)a.jac.blocks;
        scan_lds = std::max(scan_lds, JAC_LDS_BYTES);
    }
    // how the row minima will leave this launch (what `defer` reports below), known before the launch: the loss riders read them
    SymPartials plan{nullptr, nullptr, 1, b, false, nullptr};
    bool merge_launch = false;
    if (rslices == 1 && s.rtiles == 1) {}                               // final in dist1 / idx1
    else if (a.row64) { plan.deferred = true; plan.slices = 0; plan.row64 = a.row64; }
    else if (defer && s.rtiles == 1 && rslices <= 8) { plan.deferred = true; plan.slices = rslices; plan.rowpart_d = a.rowpart_d; plan.rowpart_i = a.rowpart_i; }
    else merge_launch = true;
    if (host_loss && !merge_launch) {
        // behind everything else: a cloud's riders can only start once its scan and search workgroups are done.  Every workgroup
        // of the scan (np problems, gated off or not) and of the search counts itself into done[cloud] once per call.
        loss->first_block = (int)grid;
        loss->blocks = loss_rider_blocks(b, loss->rows);
        loss->clouds = b;
        loss->target += (unsigned)(np * s.rtiles * s.cslices + (rider ? 2 * GR_QSPLIT : 0));
        if (loss->patch) loss->patch(*loss, plan, loss->ctx);
        a.loss = *loss;
        grid += (unsigned)a.loss.blocks;
        scan_lds = std::max(scan_lds, loss_cgrad_lds_bytes(n));
    } else if (a.rider.done) a.rider.done = nullptr;
    if (!s.mx) chamfer_sym_kernel<<<grid, CS_THREADS, scan_lds, stream>>>(a);
    else if (s.C == 256) chamfer_mx_kernel<8><<<grid, MX_THREADS, scan_lds, stream>>>(a);
    else if (s.C == 128) chamfer_mx_kernel<4><<<grid, MX_THREADS, scan_lds, stream>>>(a);
    else chamfer_mx_kernel<2><<<grid, MX_THREADS, scan_lds, stream>>>(a);
    GA_LAUNCH_CHECK();
    if (rslices == 1 && s.rtiles == 1) return GEOADV_OK;  // both sides left the scan final
    if (a.row64) {                                         // the caller's next launch unpacks the folded words on its way in
        defer->deferred = true; defer->slices = 0; defer->row64 = a.row64;
        return GEOADV_OK;
    }
    // the caller's next launch merges the row partials on its way in -- up to 8 slices: a cloud's partials are read by ONE
    // workgroup there (64 KB of distances at 8 slices, ~100 GB/s per CU); the narrow slices of small batches keep the merge launch
    if (defer && s.rtiles == 1 && rslices <= 8) {
        defer->slices = rslices; defer->rowpart_d = a.rowpart_d; defer->rowpart_i = a.rowpart_i; defer->deferred = true;
        return GEOADV_OK;
    }
    chamfer_sym_merge_kernel<<<dim3(cdiv(std::max(n, m), 256), b * np), 256, 0, stream>>>(a);
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

// Scratch of geoadv_chamfer_matrix for `cnt` pairs processed together, in floats: the four nn_distance outputs of every pair +
// the symmetric scan's partials at the launch shape `cnt` groups get (fewer groups = narrower column slices = more row partials each).
static size_t matrix_floats(size_t cnt, int n, int m) {
    return cnt * 2 * ((size_t)n + m) + cnt * std::max(sym_group_floats(sym_shape((long)cnt, n, m, false, false), n, m),
                                                      sym_group_floats(sym_shape((long)cnt, n, m, false, true), n, m)) + 64;
}

extern "C" size_t geoadv_chamfer_matrix_workspace_floats(int na, int nb, int n, int m) {
    if (na <= 0 || nb <= 0 || n <= 0 || m <= 0) return 256;
    const size_t pairs = (size_t)na * nb;
    const size_t chunk = pairs < 8192 ? pairs : 8192;
    return std::max(matrix_floats(chunk, n, m), matrix_floats(1, n, m)) + 256;
}

extern "C" int geoadv_chamfer_matrix(int na, int nb, int n, int m, const float *A, const float *B, float *out,
                                     float *workspace, size_t workspace_floats, void *stream) {
    GA_REQUIRE(na >= 0 && nb >= 0 && n >= 1 && m >= 1, "chamfer_matrix: bad dimensions (na=%d nb=%d n=%d m=%d)", na, nb, n, m);
    if (na == 0 || nb == 0) return GEOADV_OK;
    GA_REQUIRE(A && B && out && workspace, "chamfer_matrix: null pointer");
    GA_REQUIRE(workspace_floats >= matrix_floats(1, n, m) + 256, "chamfer_matrix: workspace too small (%zu floats, need >= %zu)",
               workspace_floats, matrix_floats(1, n, m) + 256);
    const size_t avail = workspace_floats - 256;
    hipStream_t st = as_stream(stream);
    const size_t pairs = (size_t)na * nb;
    for (size_t base = 0; base < pairs;) {
        size_t cnt_ = std::min<size_t>(pairs - base, 32768);
        while (cnt_ > 1 && matrix_floats(cnt_, n, m) > avail) cnt_ = cnt_ * 7 / 8;      // as many pairs per launch as the workspace holds
        const int cnt = (int)cnt_;
        float *d1 = workspace, *d2 = d1 + (size_t)cnt * n;
        int *i1 = reinterpret_cast<int *>(d2 + (size_t)cnt * m), *i2 = i1 + (size_t)cnt * n;
        float *ws = reinterpret_cast<float *>(i2 + (size_t)cnt * m);
        const ChamferPair pr{A, B, d1, i1, d2, i2};
        GA_REQUIRE(base <= 0x7fffffff, "chamfer_matrix: too many pairs");
        if (int rc = launch_chamfer_sym_ex(&pr, 1, cnt, n, m, ws, (int)base, nb, nullptr, st)) return rc;
        chamfer_pair_mean_kernel<<<cnt, 256, 0, st>>>(n, m, d1, d2, out + base);
        GA_LAUNCH_CHECK();
        base += cnt_;
    }
    return GEOADV_OK;
}

// nn_distance through the symmetric scan as an operator: same outputs as geoadv_nn_distance, bit for bit.
// The operator launches with exactly b live groups, so its scratch is the partials of THAT shape (the bound over every possible
// live-group count, chamfer_sym_workspace_floats, is what an attack handle needs: its need flags change the count from call to call).
static size_t nn_sym_floats(int b, int n, int m) {
    return (size_t)b * std::max(sym_group_floats(sym_shape((long)b, n, m, false, false), n, m), sym_group_floats(sym_shape((long)b, n, m, false, true), n, m)) + 64;
}
extern "C" int geoadv_set_chamfer_screen(int on) { return g_mx_on.exchange(on ? 1 : 0); }
// 1 if geoadv_nn_distance_sym answers this shape with the matrix-pipe-screened kernel (chamfer_mx.h), 0 if with the unscreened scan
extern "C" int geoadv_nn_distance_sym_is_screened(int b, int n, int m) {
    return (b > 0 && n > 0 && m > 0 && sym_shape((long)b, n, m, false, mx_enabled()).mx) ? 1 : 0;
}
extern "C" size_t geoadv_nn_distance_sym_workspace_floats(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0) return 64;
    return nn_sym_floats(b, n, m);
}
extern "C" int geoadv_nn_distance_sym(int b, int n, const float *xyz1, int m, const float *xyz2, float *dist1, int *idx1,
                                      float *dist2, int *idx2, float *workspace, size_t workspace_floats, void *stream) {
    GA_REQUIRE(b >= 0 && n >= 1 && m >= 1, "nn_distance_sym: bad dimensions (b=%d n=%d m=%d)", b, n, m);
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(xyz1 && xyz2 && dist1 && idx1 && dist2 && idx2 && workspace, "nn_distance_sym: null pointer");
    GA_REQUIRE(workspace_floats >= nn_sym_floats(b, n, m), "nn_distance_sym: workspace too small (%zu floats, need %zu)",
               workspace_floats, nn_sym_floats(b, n, m));
    const ChamferPair pr{xyz1, xyz2, dist1, idx1, dist2, idx2};
    if (int rc = launch_chamfer_sym(&pr, 1, b, n, m, workspace, as_stream(stream))) return rc;
    return launch_nn_nonfinite_fix(b, n, xyz1, m, xyz2, dist1, idx1, dist2, idx2, as_stream(stream));
}
GA_STAMPS_GETTER(geoadv_debug_stamps_chamfer_sym)
