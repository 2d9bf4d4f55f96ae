// Device side of the paired exact grid search (see chamfer_grid.hip for the method); a header so that its workgroups can
// ride in another kernel's launch (decoder.hip: the FC2 forward leaves most of the chip idle).
#pragma once
#include "common.h"
#include <limits.h>
#include <math.h>

#pragma clang fp contract(off)

namespace geoadv {

constexpr int GR_THREADS = 512;
constexpr int GR_G = 16;                        // cells per axis
constexpr int GR_CELLS = GR_G * GR_G * GR_G;
constexpr int GR_MAX_N = 4096;                  // points per cloud of the default instantiation (two workgroups per CU up to n = 2048)
constexpr int GR_MAX_N_BIG = 8192;              // the large one: 152 KB of the CU's 160 KB of LDS at n = 8192, one workgroup per CU
constexpr int GR_MAX_SPAN = 4;                  // cells per axis a lane walks on its own
constexpr int GR_QSPLIT = 4;                    // workgroups per (cloud, direction): each sorts all targets, answers 1/4 of the queries
constexpr int GR_MEAN_CELLS = 2;                // give the cloud back if its near balls touch more cells than this on average: the
                                                // walk is latency-bound and divergent (~1 us of launch time per mean candidate,
                                                // measured), the all-pairs kernel prices the whole cloud at ~0.7 us ...
constexpr int GR_FAR_DIV = 64;                  // ... or if more than 1/64 of the queries are far
constexpr int GR_RETRY = 16;                    // a workgroup that gave up looks again every 16th call
constexpr int GR_RECHECK = 4;                   // one in good standing has its verdict re-examined every 4th call

__device__ __forceinline__ float gr_sqdist(float tx, float ty, float tz, float qx, float qy, float qz) {
    const float dx = tx - qx, dy = ty - qy, dz = tz - qz;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    return (xx + yy) + zz;
}

// cell index along one axis of a grid with origin lo and 1 / cell size ih: monotone in v (every operation is), clamped
__device__ __forceinline__ int gr_cell1(float v, float lo, float ih) {
    const int c = (int)floorf((v - lo) * ih);
    return c < 0 ? 0 : (c > GR_G - 1 ? GR_G - 1 : c);
}

struct GridArgs {
    const float *P, *Q;           // [b][n][3]
    float *d1; int *i1;           // P -> Q
    float *d2; int *i2;           // Q -> P
    int n;
    int *need;                    // null, or int[8 * b]: (cloud, direction, query slice) -> 1 if that workgroup gave up
                                  // (poor pairing: the caller runs the all-pairs kernel for the cloud), 0 if it wrote its outputs
    const int *need_prev;         // null: the verdict takes effect in THIS call (the caller's all-pairs launch comes after this
                                  // one and reads `need`).  Else int[8 * b], the verdicts of the PREVIOUS call: this search
                                  // shares its launch with the all-pairs kernel, which therefore acts on need_prev -- a
                                  // workgroup answers its queries iff need_prev says 0 (whatever it thinks of the pairing now:
                                  // the search is exact either way, only slower on a poor pairing) and writes its new verdict
                                  // to `need` for the next call.
    int call;                     // running call number (the caller's iteration): paces the retries of workgroups that gave up
    const float *box;             // null, or [b][6]: min xyz, max xyz of every Q cloud (constant over the attack: computed once)
};

// LDS: sorted targets float4 (x, y, z, index bits) [n], cell_start u16 [GR_CELLS + 1], scratch u32 [GR_CELLS] (counts,
// later the queue of far queries)
template <int MAXN>
__device__ __forceinline__ void grid_nn_block(const GridArgs &a, const int cloud, const int dir, const int slice) {
    static_assert(MAXN % GR_THREADS == 0 && MAXN <= 65535 && MAXN / GR_QSPLIT <= GR_CELLS, "u16 cell offsets; the far queue reuses the counts");
    extern __shared__ __attribute__((aligned(16))) unsigned char gr_lds[];
    float4 *sorted = reinterpret_cast<float4 *>(gr_lds);
    unsigned *counts = reinterpret_cast<unsigned *>(sorted + a.n);
    unsigned short *cell_start = reinterpret_cast<unsigned short *>(counts + GR_CELLS);
    // (static LDS of a multiple of 16 bytes -- 32 + 192 + 16 -- so that the dynamic region behind it, which the kernels
    // hosting this block read with ds_read_b128, stays 16-byte aligned: cdna_hip_programming.md, Guideline 17)
    __shared__ unsigned wave_tot[GR_THREADS / 64];
    __shared__ float bb[GR_THREADS / 64][6];
    __shared__ int gr_cnt[4];
    int &n_far = gr_cnt[0], &far_cnt = gr_cnt[1], &cand_cnt = gr_cnt[2];

    const int n = a.n, t = threadIdx.x;
    const int need_off = (2 * cloud + dir) * GR_QSPLIT + slice;
    int *my_need = a.need ? a.need + need_off : nullptr;
    // a workgroup that gave up keeps its flag and leaves at once, except on every GR_RETRY-th call (the attack's points keep
    // moving: a cloud that was hopeless may have become easy and vice versa)
    const int gave_up_before = my_need ? (a.need_prev ? a.need_prev[need_off] : *my_need) : 0;   // (requested together with the point loads below)
    const int jbeg = (int)((long)n * slice / GR_QSPLIT), jend = (int)((long)n * (slice + 1) / GR_QSPLIT);   // this workgroup's queries
    const float *A = (dir ? a.Q : a.P) + (size_t)cloud * n * 3;      // queries
    const float *T = (dir ? a.P : a.Q) + (size_t)cloud * n * 3;      // targets
    float *dist = (dir ? a.d2 : a.d1) + (size_t)cloud * n;
    int *idx = (dir ? a.i2 : a.i1) + (size_t)cloud * n;

    // ---- the grid is fitted to the bounding box of the SECOND cloud (the attack's clean source cloud) in both directions, so
    // the cell size follows the scale of the shape and is not stretched by the few points an attack throws far out ----
    for (int c = t; c < GR_CELLS; c += GR_THREADS) counts[c] = 0;
    if (t == 0) { n_far = 0; far_cnt = 0; cand_cnt = 0; }
    constexpr int PER = MAXN / GR_THREADS;                             // points per thread at most (8 or 16)
    float tx[PER], ty[PER], tz[PER];
    const float *Bx = a.Q + (size_t)cloud * n * 3;
    const float *Ox = (dir ? a.Q : a.P) + (size_t)cloud * n * 3;       // the cloud that is NOT the target set
    float ox[PER], oy[PER], oz[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = t + k * GR_THREADS;
        if (i < n) {
            tx[k] = T[3 * i]; ty[k] = T[3 * i + 1]; tz[k] = T[3 * i + 2];
            if (my_need) { ox[k] = Ox[3 * i]; oy[k] = Ox[3 * i + 1]; oz[k] = Ox[3 * i + 2]; }
        }
    }
    if (gave_up_before != 0 && (a.call % GR_RETRY) != 0) {
        if (a.need_prev && t == 0) *my_need = gave_up_before;         // (two flag arrays: carry the verdict over)
        return;
    }
    float lo[3], ih[3];
    if (a.box) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lo[c] = a.box[cloud * 6 + c];
            ih[c] = (float)GR_G / fmaxf(a.box[cloud * 6 + 3 + c] - lo[c], 1e-6f);
        }
        __syncthreads();
    } else {
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = t; i < n; i += GR_THREADS) {
            const float bx = Bx[3 * i], by = Bx[3 * i + 1], bz = Bx[3 * i + 2];
            mn[0] = fminf(mn[0], bx); mn[1] = fminf(mn[1], by); mn[2] = fminf(mn[2], bz);
            mx[0] = fmaxf(mx[0], bx); mx[1] = fmaxf(mx[1], by); mx[2] = fmaxf(mx[2], bz);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int c = 0; c < 3; ++c) { mn[c] = fminf(mn[c], __shfl_xor(mn[c], off)); mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], off)); }
        if ((t & 63) == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { bb[t >> 6][c] = mn[c]; bb[t >> 6][3 + c] = mx[c]; }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float l = bb[0][c], h = bb[0][3 + c];
#pragma unroll
            for (int w = 1; w < GR_THREADS / 64; ++w) { l = fminf(l, bb[w][c]); h = fmaxf(h, bb[w][3 + c]); }
            lo[c] = l;
            ih[c] = (float)GR_G / fmaxf(h - l, 1e-6f);                 // identical in every thread: the grid is one grid
        }
    }
    GA_STAMP(1, 1);
    // ---- is the pairing good enough?  Decided for the WHOLE cloud from the pairs (P_j, Q_j) alone -- ball radius r_j around
    // Q_j in this grid -- so that all eight workgroups of the cloud reach the same verdict (a cloud answered half by this
    // search and then again by the all-pairs kernel would pay twice) ----
    // (a cloud in good standing is re-examined every GR_RECHECK-th call only: the verdict changes slowly, the examination costs
    // a fifth of the whole search)
    if (my_need && (gave_up_before != 0 || (a.call % GR_RECHECK) == 0)) {
        int lf = 0, lc = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = t + k * GR_THREADS;
            if (i < n) {
                const float qx = dir ? ox[k] : tx[k], qy = dir ? oy[k] : ty[k], qz = dir ? oz[k] : tz[k];   // Q_i
                const float r = sqrtf(gr_sqdist(tx[k], ty[k], tz[k], ox[k], oy[k], oz[k])) * 1.0001f + 1e-5f;
                const int sx = gr_cell1(qx + r, lo[0], ih[0]) - gr_cell1(qx - r, lo[0], ih[0]) + 1;
                const int sy = gr_cell1(qy + r, lo[1], ih[1]) - gr_cell1(qy - r, lo[1], ih[1]) + 1;
                const int sz = gr_cell1(qz + r, lo[2], ih[2]) - gr_cell1(qz - r, lo[2], ih[2]) + 1;
                if (sx > GR_MAX_SPAN || sy > GR_MAX_SPAN || sz > GR_MAX_SPAN) ++lf; else lc += sx * sy * sz;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { lf += __shfl_xor(lf, off); lc += __shfl_xor(lc, off); }
        if ((t & 63) == 0) { atomicAdd(&far_cnt, lf); atomicAdd(&cand_cnt, lc); }
        __syncthreads();
        const bool give_up = far_cnt * GR_FAR_DIV > n || cand_cnt > GR_MEAN_CELLS * n;
        if (t == 0) *my_need = give_up ? 1 : 0;
        if (a.need_prev ? gave_up_before != 0 : give_up) return;       // shared launch: the all-pairs kernel follows the OLD verdict
    } else if (a.need_prev && my_need && t == 0) {
        *my_need = 0;                                                  // in good standing, not re-examined this call
    }
    GA_STAMP(1, 2);
    // ---- counting sort of the targets by cell ----
    int tcell[PER], trank[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = t + k * GR_THREADS;
        if (i < n) {
            tcell[k] = (gr_cell1(tz[k], lo[2], ih[2]) * GR_G + gr_cell1(ty[k], lo[1], ih[1])) * GR_G + gr_cell1(tx[k], lo[0], ih[0]);
            trank[k] = (int)atomicAdd(&counts[tcell[k]], 1u);
        }
    }
    __syncthreads();
    {   // exclusive scan of counts[GR_CELLS] -> cell_start; thread t owns cells [8t, 8t + 8)
        constexpr int CPT = GR_CELLS / GR_THREADS;
        unsigned local[CPT], sum = 0;
#pragma unroll
        for (int k = 0; k < CPT; ++k) { local[k] = counts[t * CPT + k]; sum += local[k]; }
        unsigned incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned v = __shfl_up(incl, off);
            if ((t & 63) >= off) incl += v;
        }
        if ((t & 63) == 63) wave_tot[t >> 6] = incl;
        __syncthreads();
        unsigned base = incl - sum;
        for (int w = 0; w < (t >> 6); ++w) base += wave_tot[w];
#pragma unroll
        for (int k = 0; k < CPT; ++k) { cell_start[t * CPT + k] = (unsigned short)base; base += local[k]; }
        if (t == GR_THREADS - 1) cell_start[GR_CELLS] = (unsigned short)base;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = t + k * GR_THREADS;
        if (i < n) sorted[cell_start[tcell[k]] + trank[k]] = make_float4(tx[k], ty[k], tz[k], __int_as_float(i));
    }
    __syncthreads();
    GA_STAMP(1, 3);
    // the counts are dead: queue of far queries -- index and coordinates (16 B each, so that the scan below does not start
    // with a global round trip) where the slice's queries fit the 16 KB, the index alone for the large instantiation
    constexpr bool QXYZ = (MAXN / GR_QSPLIT) * 4 <= GR_CELLS;
    unsigned *far_queue = counts;
    float4 *far_queue4 = reinterpret_cast<float4 *>(counts);

    // ---- queries of this workgroup's slice: bound from the paired target, cells the ball touches, candidates in them ----
    constexpr int QPT = (MAXN / GR_QSPLIT + GR_THREADS - 1) / GR_THREADS;              // queries per thread (2 or 4)
    float qx[QPT], qy[QPT], qz[QPT], qbest[QPT];
    int qspan[QPT];                                    // x0 | x1 << 4 | y0 << 8 | y1 << 12 | z0 << 16 | z1 << 20, or -1 = far
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
        const int j = jbeg + t + k * GR_THREADS;
        qspan[k] = 0;
        if (j < jend) {
            qx[k] = A[3 * j]; qy[k] = A[3 * j + 1]; qz[k] = A[3 * j + 2];
            qbest[k] = gr_sqdist(T[3 * j], T[3 * j + 1], T[3 * j + 2], qx[k], qy[k], qz[k]);   // the paired target
            const float r = sqrtf(qbest[k]) * 1.0001f + 1e-5f;
            const int x0 = gr_cell1(qx[k] - r, lo[0], ih[0]), x1 = gr_cell1(qx[k] + r, lo[0], ih[0]);
            const int y0 = gr_cell1(qy[k] - r, lo[1], ih[1]), y1 = gr_cell1(qy[k] + r, lo[1], ih[1]);
            const int z0 = gr_cell1(qz[k] - r, lo[2], ih[2]), z1 = gr_cell1(qz[k] + r, lo[2], ih[2]);
            if (x1 - x0 >= GR_MAX_SPAN || y1 - y0 >= GR_MAX_SPAN || z1 - z0 >= GR_MAX_SPAN) qspan[k] = -1;
            else qspan[k] = x0 | x1 << 4 | y0 << 8 | y1 << 12 | z0 << 16 | z1 << 20;
        }
    }
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
        const int j = jbeg + t + k * GR_THREADS;
        if (j >= jend) continue;
        if (qspan[k] < 0) {
            const int pos = atomicAdd(&n_far, 1);                        // (order of the queue does not matter)
            if (QXYZ) far_queue4[pos] = make_float4(qx[k], qy[k], qz[k], __int_as_float(j));
            else far_queue[pos] = (unsigned)j;
            continue;
        }
        const int x0 = qspan[k] & 15, x1 = (qspan[k] >> 4) & 15, y0 = (qspan[k] >> 8) & 15, y1 = (qspan[k] >> 12) & 15;
        const int z0 = (qspan[k] >> 16) & 15, z1 = (qspan[k] >> 20) & 15;
        float best = qbest[k];
        int bestk = j;
        for (int cz = z0; cz <= z1; ++cz)
            for (int cy = y0; cy <= y1; ++cy) {
                const int row = (cz * GR_G + cy) * GR_G;
                const int s = cell_start[row + x0], e = cell_start[row + x1 + 1];   // cells x0..x1 are contiguous
                for (int u = s; u < e; ++u) {
                    const float4 p = sorted[u];
                    const float d = gr_sqdist(p.x, p.y, p.z, qx[k], qy[k], qz[k]);
                    const int kk = __float_as_int(p.w);
                    if (d < best || (d == best && kk < bestk)) { best = d; bestk = kk; }
                }
            }
        dist[j] = best;
        idx[j] = bestk;
    }
    __syncthreads();
    GA_STAMP(1, 4);
    // ---- far queries: one wave per query, all targets ----
    const int lane = t & 63, wave = t >> 6;
    const int nf = n_far;
    for (int f = wave; f < nf; f += GR_THREADS / 64) {
        int j;
        float fx, fy, fz;
        if (QXYZ) { const float4 q = far_queue4[f]; fx = q.x; fy = q.y; fz = q.z; j = __float_as_int(q.w); }
        else { j = (int)far_queue[f]; fx = A[3 * j]; fy = A[3 * j + 1]; fz = A[3 * j + 2]; }
        float best = INFINITY;
        int bestk = INT_MAX;
        // eight independent streams per lane keep eight LDS reads in flight (a single dependent chain pays the LDS
        // latency 32 times per query); merged lexicographically afterwards
        float bd[8];
        int bk[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) { bd[v] = INFINITY; bk[v] = INT_MAX; }
        int u = lane;
        for (; u + 7 * 64 < n; u += 8 * 64) {
            float4 p[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) p[v] = sorted[u + 64 * v];
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const float d = gr_sqdist(p[v].x, p[v].y, p[v].z, fx, fy, fz);
                const int k = __float_as_int(p[v].w);
                if (d < bd[v] || (d == bd[v] && k < bk[v])) { bd[v] = d; bk[v] = k; }
            }
        }
        for (; u < n; u += 64) {
            const float4 p = sorted[u];
            const float d = gr_sqdist(p.x, p.y, p.z, fx, fy, fz);
            const int k = __float_as_int(p.w);
            if (d < bd[0] || (d == bd[0] && k < bk[0])) { bd[0] = d; bk[0] = k; }
        }
#pragma unroll
        for (int v = 0; v < 8; ++v)
            if (bd[v] < best || (bd[v] == best && bk[v] < bestk)) { best = bd[v]; bestk = bk[v]; }
        wave_lexmin(best, bestk);
        if (lane == 0) { dist[j] = best; idx[j] = bestk; }
    }
}

// The search's workgroups as extra blocks of another kernel's launch (blocks first_block .. of a 1-D grid, GR_THREADS threads
// or more -- surplus waves leave): blocks == 0 = none.  Block order: XCD-aware -- workgroups are dealt round-robin over the 8 XCDs,
// so block g of the rider serves cloud (g / 64) * 8 + g % 8 and its (direction, slice) unit (g / 8) % 8: the eight workgroups of a
// cloud have equal g % 8, i.e. ONE XCD, whose L2 then fetches the cloud's 49 KB once (dealt cloud-major, all eight XCDs fetched every
// cloud: 12.6 of the scan launch's 16.7 MB of counted reads at B = 32).  blocks = 64 * ceil(clouds / 8); units of absent clouds leave.
struct GridRider { GridArgs g; int first_block, blocks, clouds; unsigned *done; };   // done: null, or per-cloud arrival counters (loss_cgrad.h: LossRider)
inline int grid_rider_blocks(int clouds) { return 8 * 2 * GR_QSPLIT * ((clouds + 7) / 8); }
template <int MAXN>
__device__ __forceinline__ bool grid_rider_block(const GridRider &r) {   // true: this workgroup belonged to the rider and is done
    if (r.blocks == 0 || (int)blockIdx.x < r.first_block || (int)blockIdx.x >= r.first_block + r.blocks) return false;
    if (threadIdx.x < GR_THREADS) {
        __builtin_amdgcn_s_setprio(3);                     // short and latency-bound beside a VALU-dense host kernel: never starve it
        const int g = blockIdx.x - r.first_block;
        static_assert(2 * GR_QSPLIT == 8, "eight units per cloud: one per slot of an XCD's turn");
        const int cloud = (g >> 6) * 8 + (g & 7), unit = (g >> 3) & 7;
        if (cloud < r.clouds) {
            GA_STAMP(1, 0);
            grid_nn_block<MAXN>(r.g, cloud, unit / GR_QSPLIT, unit % GR_QSPLIT);
            GA_STAMP(1, 7);
            if (r.done) {                                  // (the search leaves its block uniformly: every thread arrives here)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(r.done + cloud, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    return true;
}

inline size_t chamfer_grid_lds_bytes(int n) { return sizeof(float4) * (size_t)n + sizeof(unsigned) * GR_CELLS + sizeof(unsigned short) * (GR_CELLS + 2); }

}  // namespace geoadv
