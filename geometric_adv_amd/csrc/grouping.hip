// Grouping ops (external/grouping) for gfx950: QueryBallPoint, SelectionSort, GroupPoint(+Grad)
// and the fused k-NN the defense uses (tf_grouping.py:48-75, defender/get_knn_dists_per_point.py:78-81).
//
// The reference materialises a dense (b,m,n) distance matrix plus two tiled (b,m,n,3) operands
// (1.7 GB at b=100, n=2048) and then lets ONE THREAD per row run k passes of "find the first
// minimum of [s,n), swap it into s" (tf_grouping_g.cu:83-123).  Here a wave owns a row: the row
// lives in LDS (values + indices), distances are computed straight into it, and every pass is a
// wave-wide lexicographic (value, position) arg-min followed by the same swap -- so the result,
// including the reference's peculiar order among equal distances, is identical, and nothing of
// size n*m ever touches HBM in the fused form.
#include <atomic>
#include "common.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>

#pragma clang fp contract(off)

namespace geoadv {

constexpr int ROW_MAX_N = 16384;          // a row (float + int per entry) must fit in LDS: 128 KB

// k passes of the reference's partial selection sort on an LDS row.  val/idx: [n].  One wave.
__device__ __forceinline__ void wave_selection_sort(float *val, int *idx, int n, int k) {
    const int lane = threadIdx.x & 63;
    for (int s = 0; s < k && s < n; ++s) {
        // first minimum of positions [s, n): start with position s, replace only on strict '<'
        // => lexicographic min of (value, position)
        float bv = INFINITY;
        int bp = 0x7fffffff;
        {   // four independent streams per lane (a single chain pays the LDS latency once per entry: 32 times per pass at n = 2048),
            // merged lexicographically -- the minimum of (value, position) does not depend on the order it is taken in
            float sv[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
            int sq[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
            int t = s + lane;
            for (; t + 3 * 64 < n; t += 4 * 64) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = val[t + 64 * u];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (v[u] < sv[u]) { sv[u] = v[u]; sq[u] = t + 64 * u; }     // ascending positions within a stream: first min kept
            }
            for (; t < n; t += 64) {
                const float v = val[t];
                if (v < sv[0]) { sv[0] = v; sq[0] = t; }                       // (later positions than anything stream 0 has seen)
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (sv[u] < bv || (sv[u] == bv && sq[u] < bp)) { bv = sv[u]; bp = sq[u]; }
        }
        // a lane whose values are all NaN / that saw nothing keeps (inf, maxint)
        wave_lexmin(bv, bp);                      // (DPP + readlane: six ds_bpermute round trips per pass were half of a pass)
        // (bv, bp) = the first position of [s, n) attaining the minimum, exactly what the reference's
        // scan "min = s; if (p[t] < p[min]) min = t" finds; nothing comparable (all NaN) keeps s.
        const int mn = bp == 0x7fffffff ? s : bp;
        if (mn != s) {
            if (lane == 0) {
                const float tv = val[mn]; val[mn] = val[s]; val[s] = tv;
                const int ti = idx[mn]; idx[mn] = idx[s]; idx[s] = ti;
            }
        }
        __syncthreads();                          // block = one wave: orders the swap before the next pass
    }
}

// ------------------------------------------------------------------------------------------
// SelectionSort op: dist (b,m,n) -> outi (b,m,n), out (b,m,n).  grid = rows, block = 64.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void selection_sort_kernel(int n, int k, size_t rows, const float *dist, int *outi,
                                                            float *out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *val = lds;
    int *idx = reinterpret_cast<int *>(lds + n);
    for (size_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const float *src = dist + row * n;
        for (int t = threadIdx.x; t < n; t += 64) { val[t] = src[t]; idx[t] = t; }
        __syncthreads();
        wave_selection_sort(val, idx, n, k);
        for (int t = threadIdx.x; t < n; t += 64) { out[row * n + t] = val[t]; outi[row * n + t] = idx[t]; }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Fused knn_point: for query q of cloud c, row[p] = sum_c (xyz1[p,c]-xyz2[q,c])^2 (left to right,
// tf_grouping.py:68), then the selection sort; val/idx (b,m,k).  grid = (query groups, b), block = 64.
// MODE 0: write val/idx.  MODE 1 (defender): drop column 0 and write the euclidean distances to the
// remaining k-1 neighbours, recomputed from the gathered points as the reference graph does
// (get_knn_dists_per_point.py:79-81: grouped - centre, sqrt(reduce_sum(deltas**2))).
// ------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void knn_row(float *val, int *idx, int n, int m, int k, int c, int q, const float *xyz1,
                                        const float *xyz2, float *val_out, int *idx_out) {
    const float *data = xyz1 + (size_t)c * n * 3;
    const float *qry = xyz2 + (size_t)c * m * 3;
    const float qx = qry[3 * q], qy = qry[3 * q + 1], qz = qry[3 * q + 2];
    // (eight points' loads in flight per lane: one at a time, the 32 round trips of a 2048-point row were 16 of the redo kernel's 22 us)
    int t = threadIdx.x;
    for (; t + 7 * 64 < n; t += 8 * 64) {
        float px[8], py[8], pz[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { px[u] = data[3 * (t + 64 * u)]; py[u] = data[3 * (t + 64 * u) + 1]; pz[u] = data[3 * (t + 64 * u) + 2]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float dx = px[u] - qx, dy = py[u] - qy, dz = pz[u] - qz;
            val[t + 64 * u] = (dx * dx + dy * dy) + dz * dz;
            idx[t + 64 * u] = t + 64 * u;
        }
    }
    for (; t < n; t += 64) {
        const float dx = data[3 * t] - qx, dy = data[3 * t + 1] - qy, dz = data[3 * t + 2] - qz;
        val[t] = (dx * dx + dy * dy) + dz * dz;
        idx[t] = t;
    }
    __syncthreads();
    wave_selection_sort(val, idx, n, k);
    if (MODE == 0) {
        for (int s = threadIdx.x; s < k; s += 64) {
            val_out[((size_t)c * m + q) * k + s] = val[s];
            idx_out[((size_t)c * m + q) * k + s] = idx[s];
        }
    } else {
        for (int s = threadIdx.x; s + 1 < k; s += 64) {
            const int nb = idx[s + 1];
            const float dx = data[3 * nb] - qx, dy = data[3 * nb + 1] - qy, dz = data[3 * nb + 2] - qz;
            val_out[((size_t)c * m + q) * (k - 1) + s] = sqrtf((dx * dx + dy * dy) + dz * dz);
        }
    }
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(64) void knn_kernel(int n, int m, int k, int qper, const float *xyz1, const float *xyz2,
                                                 float *val_out, int *idx_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *val = lds;
    int *idx = reinterpret_cast<int *>(lds + n);
    const int q_end = min(m, (int)(blockIdx.x + 1) * qper);
    for (int q = blockIdx.x * qper; q < q_end; ++q) knn_row<MODE>(val, idx, n, m, k, blockIdx.y, q, xyz1, xyz2, val_out, idx_out);
}

// The queries the fast kernel below handed back (redo[0] = their number, then c * m + q each).
template <int MODE>
__global__ __launch_bounds__(64) void knn_redo_kernel(int n, int m, int k, const float *xyz1, const float *xyz2, float *val_out,
                                                      int *idx_out, const int *redo) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *val = lds;
    int *idx = reinterpret_cast<int *>(lds + n);
    const int count = redo[0];
    for (int e = blockIdx.x; e < count; e += gridDim.x) {
        const int cq = redo[1 + e];
        knn_row<MODE>(val, idx, n, m, k, cq / m, cq % m, xyz1, xyz2, val_out, idx_out);
    }
}

// ------------------------------------------------------------------------------------------
// Fast k-NN: one THREAD per query keeps its S smallest (distance, index) pairs sorted in registers while the dataset
// streams through LDS (broadcast reads, four points per ds_read_b128).  The values and their order are the
// reference's whatever it does with ties (the sorted k smallest distances are what they are); the INDICES are the
// reference's whenever no two of the k + 1 smallest distances are equal -- then the selection is unique.  A query with
// such a tie (duplicated points), or one that meets a NaN / infinite distance, is put on a list and redone by the
// selection-sort kernel above, which reproduces the reference's swap order.  Visiting the dataset in ascending index
// with strict '<' keeps equal distances in index order, so the tie test only has to look at neighbours in the list.
//   MODE 0: val/idx of the k nearest (S >= k + 1: one extra slot for the tie test).
//   MODE 1: sqrt of sorted distances 1..k-1 (the defender's graph; independent of the order among ties, S >= k).
//
// DEFERRED insertion (round 4).  A lane meets ~k ln(n/k) list updates along the scan, the 64 lanes of a wave meet them at
// different points: with the update inline (rounds 1-3) two steps of three ran the whole S-slot shift for one or two live
// lanes -- 2.7 x the cost of the distances themselves (0.63 ms at 256 x 2048 x 2048, k = 8).  Now a lane only APPENDS a
// candidate (distance below its threshold `thr`, which is the list's last entry as of the last drain) to its own LDS queue --
// one masked ds_write -- and the queues are DRAINED into the lists together: every KF_DRAIN points, and whenever some lane's
// queue is nearly full.  A drain step runs the shift once for up to 64 lanes that all have work.  The result is the inline
// form's bit for bit: a lane's queue holds its candidates in ascending index, the drain applies the same strict '<' to each,
// and a candidate admitted by a stale threshold that the current list no longer admits is dropped by that test.
// `!(d >= thr)` admits NaN distances too (they mark the query for the redo kernel at the drain); the LDS pad is +inf
// coordinates: distance +inf, never admitted.
// ------------------------------------------------------------------------------------------
constexpr int KF_THREADS = 256;
constexpr int KF_TILE = 1024;
#ifndef KF_QCAP_V
#define KF_QCAP_V 16
#endif
constexpr int KF_QCAP = KF_QCAP_V;         // queue slots per lane
constexpr int KF_DRAIN = 128;              // scheduled drain period (points)

template <int MODE, int S>
__global__ __launch_bounds__(KF_THREADS) void knn_fast_kernel(int n, int m, int k, const float *xyz1, const float *xyz2,
                                                              float *val_out, int *idx_out, int *redo) {
    __shared__ __attribute__((aligned(16))) float sx[KF_TILE], sy[KF_TILE], sz[KF_TILE];
    __shared__ float qd[KF_QCAP * KF_THREADS];                       // slot-major: lane-consecutive addresses, no bank conflicts
    __shared__ int qi[MODE == 0 ? KF_QCAP * KF_THREADS : 1];
    const int c = blockIdx.y;
    const float *data = xyz1 + (size_t)c * n * 3;
    const int q = blockIdx.x * KF_THREADS + threadIdx.x;
    const bool live = q < m;
    const float *qp = xyz2 + ((size_t)c * m + (live ? q : m - 1)) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    float v[S];
    int ix[S];
#pragma unroll
    for (int i = 0; i < S; ++i) { v[i] = INFINITY; ix[i] = -1; }
    bool odd = false;                                      // met a NaN distance
    float thr = INFINITY;                                  // v[S - 1] as of the last drain
    unsigned qw = 4u * threadIdx.x;                        // BYTE offset of this lane's next queue entry: slot-major, (slot * KF_THREADS + lane) * 4
                                                           // (a running offset: a push is one ds_write and one add)

    auto drain = [&]() {
        for (unsigned jo = 4u * threadIdx.x; __any(jo < qw); jo += 4u * KF_THREADS) {
            const bool has = jo < qw;
            const float d = has ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(qd) + jo) : INFINITY;
            int id = 0;
            if (MODE == 0) id = has ? *reinterpret_cast<const int *>(reinterpret_cast<const char *>(qi) + jo) : 0;
            odd |= d != d;
            if (MODE == 1) {                               // values only: sorted insertion is a median per slot, v[i] <- med3(v[i - 1], d, v[i])
#pragma unroll                                             // from the top down (the OLD v[i - 1]): S instructions where a compare-exchange chain
                for (int i = S - 1; i > 0; --i) v[i] = __builtin_amdgcn_fmed3f(v[i - 1], d, v[i]);      // takes 2 S (a NaN only ever marks
                v[0] = fminf(v[0], d);                                                                    // the query `odd`: its list is not used)
            } else {                                       // (value, index): the candidate goes in front of the first entry it is STRICTLY below
                bool below[S];                             // (an equal entry -- earlier index -- stays ahead of it); monotone along the list
#pragma unroll
                for (int i = 0; i < S; ++i) below[i] = d < v[i];
#pragma unroll
                for (int i = S - 1; i > 0; --i) {
                    v[i] = __builtin_amdgcn_fmed3f(v[i - 1], d, v[i]);
                    ix[i] = below[i - 1] ? ix[i - 1] : (below[i] ? id : ix[i]);
                }
                v[0] = fminf(v[0], d);
                ix[0] = below[0] ? id : ix[0];
            }
        }
        qw = 4u * threadIdx.x;
        thr = v[S - 1];
    };

    for (int t0 = 0; t0 < n; t0 += KF_TILE) {
        const int cnt = min(KF_TILE, n - t0);
        __syncthreads();
        for (int e = threadIdx.x; e < KF_TILE; e += KF_THREADS) {
            const bool in = e < cnt;                       // the pad is never admitted: its distance is +inf
            sx[e] = in ? data[3 * (size_t)(t0 + e)] : INFINITY; sy[e] = in ? data[3 * (size_t)(t0 + e) + 1] : INFINITY;
            sz[e] = in ? data[3 * (size_t)(t0 + e) + 2] : INFINITY;
        }
        __syncthreads();
        for (int e1 = 0; e1 < cnt; e1 += KF_DRAIN) {
            const int e1_end = min(cnt, e1 + KF_DRAIN);
            for (int e0 = e1; e0 < e1_end; e0 += 4) {
                const float4 xa = *reinterpret_cast<const float4 *>(&sx[e0]);
                const float4 ya = *reinterpret_cast<const float4 *>(&sy[e0]);
                const float4 za = *reinterpret_cast<const float4 *>(&sz[e0]);
                const float tx[4] = {xa.x, xa.y, xa.z, xa.w}, ty[4] = {ya.x, ya.y, ya.z, ya.w}, tz[4] = {za.x, za.y, za.z, za.w};
                float d[4];
                bool adm[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float dx = tx[u] - qx, dy = ty[u] - qy, dz = tz[u] - qz;
                    d[u] = (dx * dx + dy * dy) + dz * dz;                            // tf_grouping.py:68, left to right
                    adm[u] = !(d[u] >= thr);
                }
                if (__any(adm[0] | adm[1] | adm[2] | adm[3])) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (adm[u]) {
                            *reinterpret_cast<float *>(reinterpret_cast<char *>(qd) + qw) = d[u];
                            if (MODE == 0) *reinterpret_cast<int *>(reinterpret_cast<char *>(qi) + qw) = t0 + e0 + u;
                            qw += 4u * KF_THREADS;
                        }
                    if (__any(qw >= 4u * (KF_QCAP - 3) * KF_THREADS)) drain();
                }
            }
            drain();
        }
    }
    if (!live) return;
    if (MODE == 0) {
        bool again = odd || ix[k - 1] < 0;                 // fewer than k finite distances
#pragma unroll
        for (int i = 0; i + 1 < S; ++i)
            if (i < k && v[i] == v[i + 1] && ix[i + 1] >= 0) again = true;         // a tie among the k + 1 smallest
        if (again) { redo[1 + atomicAdd(redo, 1)] = c * m + q; return; }
#pragma unroll
        for (int i = 0; i < S; ++i)
            if (i < k) { val_out[((size_t)c * m + q) * k + i] = v[i]; idx_out[((size_t)c * m + q) * k + i] = ix[i]; }
    } else {
        if (odd || !(v[k - 1] < INFINITY)) { redo[1 + atomicAdd(redo, 1)] = c * m + q; return; }
#pragma unroll
        for (int i = 1; i < S; ++i)
            if (i < k) val_out[((size_t)c * m + q) * (k - 1) + i - 1] = sqrtf(v[i]);
    }
}

// ------------------------------------------------------------------------------------------
// Grid k-NN (round 4): the same per-lane lists and queues as knn_fast_kernel, but a wave only visits the cells of a uniform
// grid that can hold neighbours of ITS 64 queries.  Exact: a lane is finished when the last entry of its list is strictly
// below the squared distance to the nearest face of the box of cells visited so far (faces on the grid's boundary do not
// count: nothing lies beyond them), with a slack for the rounding of cell assignment; until every lane is finished the box
// grows by one shell of cells, in the worst case to the whole grid (= the all-points scan, plus the build).
//   build (one workgroup per cloud): bounding box, G cells per non-degenerate axis, counting sort of the dataset into
//     row-major cells (x fastest: a run of cells along x is a contiguous range of the sorted array) -> sorted[b][n] float4
//     (x, y, z, original index) + cell_start; the QUERIES are ordered along a Hilbert curve through their cells (qorder), so 64
//     consecutive ones sit in a compact block of cells.  A cloud with a non-finite coordinate gets the 1 x 1 x 1 grid: every
//     query then meets every point, as the all-points kernel would (NaN distances send the query to the redo kernel).
//   search: every workgroup stages the whole sorted cloud in LDS (<= 4096 points, (x, y, z, index) per point: one ds_read_b128 a
//     candidate).  A task = 64 consecutive queries of the Hilbert order, one per lane, in two stages:
//     (1) lane-private: the lane walks the 3 x 3 x 3 cells around ITS query -- nine runs of the sorted cloud, nearest rows first,
//         four candidates a step, gathered at per-lane addresses -- and is done if the face-distance test on its own block closes
//         its list (at 2048 uniform points in 8^3 cells: ~110 candidates a lane, 99.97 % of the lanes);
//     (2) the lanes it leaves over (a sparse corner, an outlier) are listed per workgroup and, when the tasks are done, searched
//         64 at a time by ALL the workgroup's waves: they are scattered, their common box is the grid, so every wave scans an
//         eighth of the cloud for them (broadcast reads) and wave 0 merges the eight lists.
//     The wave-uniform shell walk (the round's first form: the wave's common box of cells, shell by shell, row by row, broadcast
//     reads; 400 candidates a lane at the same size because the box is the UNION of 64 neighbourhoods) stays as
//     geoadv_knn_grid_mode(3) and for lists too long for the merge (k > 15).  Measured at 256 x 2048, k = 8 (rocprofv3): 174 us
//     (1) + (2), 191 us shells only, 366 us all points; before the cloud went into LDS as float4 (SoA planes, and 16 KB reserved for cell
//     offsets of which 2 are used: one workgroup per CU instead of two) the shell walk took 220 us.
//     What bounds (1) now is the queues' drain, not the walk: ~115 drain steps a task against 42 walk steps -- a drain runs as long
//     as its fullest queue, and 64 lanes fill at different rates.  Each lane therefore tightens its own threshold while it pushes
//     (once S values are queued, their maximum bounds the S-th smallest); without that: 140 steps.
//     (First form of all, measured: scalar loads straight from the sorted array, s_load_dwordx8 feeding the VALU as SGPR operands
//     -- no faster than the all-points scan, every group of four points waited for its own load.)
// The order in which a lane meets its candidates is no longer the index order; the results do not depend on it (the k + 1
// smallest distinct distances are what they are, ties among them go to knn_redo_kernel as before).
// ------------------------------------------------------------------------------------------
#ifndef KNN_GRID_MIN_N
#define KNN_GRID_MIN_N 512                 // smaller datasets: the all-points kernel
#endif
#ifndef KNN_GRID_G_SMALL
#define KNN_GRID_G_SMALL 4                 // n <  KNN_GRID_N_MID
#define KNN_GRID_N_MID 1024
#define KNN_GRID_G_MID 8                   // n <  KNN_GRID_N_BIG
#define KNN_GRID_N_BIG 3072
#define KNN_GRID_G_BIG 16
#endif
constexpr int KG_MAX_G = 16, KG_MAX_CELLS = KG_MAX_G * KG_MAX_G * KG_MAX_G;
constexpr int KG_BUILD_THREADS = 1024;
constexpr int KG_MAX_N = 4096;             // the whole sorted cloud sits in the search kernel's LDS (64 KB at 4096 points)

struct KnnGrid {
    float lo[3], ih[3], h[3], eps[3];       // origin, 1 / cell size, cell size, slack on a face distance (rounding of the cell assignment)
    int g[3], cells;
    int bad;                                // the cloud holds a non-finite coordinate (1 x 1 x 1 grid; the keyed search sends its queries to the redo list)
};

__device__ __forceinline__ int kg_cell1(float v, float lo, float ih, int g) {
    const float t = fminf(fmaxf((v - lo) * ih, 0.f), (float)(g - 1));       // NaN -> 0; monotone in v
    return (int)t;
}

// Hilbert index (Skilling's transpose form, 4 bits per axis) of a query's cell on a 16-per-axis grid over the dataset's box
// (finer than the cell grid where that has fewer cells: the order only has to make 64 consecutive queries compact).  A Morton
// order was measured first: its runs jump across the cloud wherever a high bit flips, the boxes of those waves cover most of
// the grid, and a workgroup waits for its slowest wave (in-kernel stamps: median 75 us per workgroup, worst 189).
__device__ __forceinline__ unsigned kg_hilbert(const KnnGrid &g, float x, float y, float z) {
    const float s0 = g.g[0] > 1 ? 16.f / (float)g.g[0] : 0.f, s1 = g.g[1] > 1 ? 16.f / (float)g.g[1] : 0.f, s2 = g.g[2] > 1 ? 16.f / (float)g.g[2] : 0.f;
    unsigned X[3] = {(unsigned)kg_cell1(x, g.lo[0], g.ih[0] * s0, 16), (unsigned)kg_cell1(y, g.lo[1], g.ih[1] * s1, 16),
                     (unsigned)kg_cell1(z, g.lo[2], g.ih[2] * s2, 16)};
#pragma unroll
    for (unsigned Q = 8; Q > 1; Q >>= 1) {
        const unsigned P = Q - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { const unsigned t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    X[1] ^= X[0]; X[2] ^= X[1];
    unsigned t = 0;
#pragma unroll
    for (unsigned Q = 8; Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    unsigned h = 0;
#pragma unroll
    for (int b = 3; b >= 0; --b)
#pragma unroll
        for (int i = 0; i < 3; ++i) h = (h << 1) | ((X[i] >> b) & 1u);
    return h;
}

// in-place exclusive scan of a[0 .. KG_MAX_CELLS) (LDS), 4 entries per thread; returns nothing (a[i] = sum of a[0..i))
__device__ __forceinline__ void kg_scan(int *a, int *wsum) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int v[4], s = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { v[u] = a[4 * t + u]; s += v[u]; }
    int inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int before = inc - s;
    for (int w = 0; w < wave; ++w) before += wsum[w];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[4 * t + u] = before; before += v[u]; }
    __syncthreads();
}

// min / max over the 64 lanes, result uniform (as an SGPR value)
__device__ __forceinline__ int kg_wave_min(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int kg_wave_max(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    return __builtin_amdgcn_readfirstlane(v);
}

__global__ __launch_bounds__(KG_BUILD_THREADS) void knn_grid_build_kernel(int n, int m, int G, const float *xyz1, const float *xyz2,
                                                                          float4 *sorted, int *cell_start, int *qorder, KnnGrid *info,
                                                                          int *task_order, int lpt, int *redo) {
    __shared__ int cnt[KG_MAX_CELLS], qcnt[KG_MAX_CELLS];
    extern __shared__ __attribute__((aligned(16))) float kb_lds[];         // stage [KG_MAX_N] float4, qstage [KG_MAX_N] int
    float4 *stage = reinterpret_cast<float4 *>(kb_lds);
    int *qstage = reinterpret_cast<int *>(kb_lds + 4 * KG_MAX_N);
    const bool qlds = m <= KG_MAX_N;
    __shared__ float red[7][KG_BUILD_THREADS / 64];
    __shared__ int wsum[KG_BUILD_THREADS / 64];
    __shared__ KnnGrid g;
    const int c = blockIdx.x, t = threadIdx.x;
    if (c == 0 && t == 0) redo[0] = 0;                                     // the redo list's counter (the search and redo launches follow)
    const float *data = xyz1 + (size_t)c * n * 3;
    const float *qry = xyz2 + (size_t)c * m * 3;
    // The cloud (n <= KG_MAX_N = 4 x 1024) and the first 4096 queries are read ONCE and kept in registers through the three passes
    // (box, count, scatter): a pass over global memory is a round trip of ~3 us on this one-workgroup-per-cloud kernel, and there
    // were three of them (22 us for what is a few thousand LDS atomics).
    constexpr int PT = KG_MAX_N / KG_BUILD_THREADS;
    float dx[PT], dy[PT], dz[PT], ux[PT], uy[PT], uz[PT];
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const int i = t + u * KG_BUILD_THREADS;
        const size_t e = 3 * (size_t)(i < n ? i : n - 1);
        dx[u] = data[e]; dy[u] = data[e + 1]; dz[u] = data[e + 2];
        const size_t f = 3 * (size_t)(i < m ? i : m - 1);
        ux[u] = qry[f]; uy[u] = qry[f + 1]; uz[u] = qry[f + 2];
    }
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, bad = 0.f;
#pragma unroll
    for (int u = 0; u < PT; ++u) {                                       // (clamped copies of the last point change nothing)
        const float v3[3] = {dx[u], dy[u], dz[u]};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], v3[a]); hi[a] = fmaxf(hi[a], v3[a]);
            bad = fabsf(v3[a]) < INFINITY ? bad : 1.f;                  // NaN or +-inf
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], off)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off)); }
        bad = fmaxf(bad, __shfl_xor(bad, off));
    }
    if ((t & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[a][t >> 6] = lo[a]; red[3 + a][t >> 6] = hi[a]; }
        red[6][t >> 6] = bad;
    }
    for (int i = t; i < KG_MAX_CELLS; i += KG_BUILD_THREADS) { cnt[i] = 0; qcnt[i] = 0; }
    __syncthreads();
    if (t == 0) {
        float b = 0.f;
        for (int w = 0; w < KG_BUILD_THREADS / 64; ++w) b = fmaxf(b, red[6][w]);
        for (int a = 0; a < 3; ++a) {
            float l = red[a][0], h = red[3 + a][0];
            for (int w = 1; w < KG_BUILD_THREADS / 64; ++w) { l = fminf(l, red[a][w]); h = fmaxf(h, red[3 + a][w]); }
            const float ext = h - l;
            const int ga = (b == 0.f && ext > 0.f && ext < INFINITY) ? G : 1;
            g.g[a] = ga; g.lo[a] = b == 0.f ? l : 0.f;
            g.h[a] = ga > 1 ? ext / (float)ga : 0.f;
            g.ih[a] = ga > 1 ? (float)ga / ext : 0.f;
            g.eps[a] = 8.f * 1.2e-7f * fmaxf(fabsf(l), fabsf(h));
        }
        g.cells = g.g[0] * g.g[1] * g.g[2];
        g.bad = b != 0.f ? 1 : 0;
        info[c] = g;
    }
    __syncthreads();
    const int gx = g.g[0], gy = g.g[1], gz = g.g[2];
    int dcell[PT];
    unsigned qh[PT];
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const int i = t + u * KG_BUILD_THREADS;
        dcell[u] = (kg_cell1(dz[u], g.lo[2], g.ih[2], gz) * gy + kg_cell1(dy[u], g.lo[1], g.ih[1], gy)) * gx + kg_cell1(dx[u], g.lo[0], g.ih[0], gx);
        if (i < n) atomicAdd(&cnt[dcell[u]], 1);
        qh[u] = kg_hilbert(g, ux[u], uy[u], uz[u]);
        if (i < m) atomicAdd(&qcnt[qh[u]], 1);
    }
    for (int j = t + PT * KG_BUILD_THREADS; j < m; j += KG_BUILD_THREADS)         // (more than 4096 queries: the rest from memory)
        atomicAdd(&qcnt[kg_hilbert(g, qry[3 * (size_t)j], qry[3 * (size_t)j + 1], qry[3 * (size_t)j + 2])], 1);
    __syncthreads();
    kg_scan(cnt, wsum);
    kg_scan(qcnt, wsum);
    int *cs = cell_start + (size_t)c * (KG_MAX_CELLS + 1);
    for (int i = t; i < KG_MAX_CELLS; i += KG_BUILD_THREADS) cs[i] = cnt[i];         // (cells past g.cells hold n: empty)
    if (t == 0) cs[KG_MAX_CELLS] = n;
    __syncthreads();
    // the two permutations are formed in LDS and leave in order: scattered 16-byte stores straight to memory were 6 of the kernel's 21 us
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const int i = t + u * KG_BUILD_THREADS;
        if (i < n) stage[atomicAdd(&cnt[dcell[u]], 1)] = make_float4(dx[u], dy[u], dz[u], __int_as_float(i));
        if (i < m) {
            const int pos = atomicAdd(&qcnt[qh[u]], 1);
            if (qlds) qstage[pos] = i; else qorder[(size_t)c * m + pos] = i;
        }
    }
    for (int j = t + PT * KG_BUILD_THREADS; j < m; j += KG_BUILD_THREADS) {       // (m > KG_MAX_N: the order does not fit the stage)
        const unsigned mc = kg_hilbert(g, qry[3 * (size_t)j], qry[3 * (size_t)j + 1], qry[3 * (size_t)j + 2]);
        qorder[(size_t)c * m + atomicAdd(&qcnt[mc], 1)] = j;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const int i = t + u * KG_BUILD_THREADS;
        if (i < n) sorted[(size_t)c * n + i] = stage[i];
        if (qlds && i < m) qorder[(size_t)c * m + i] = qstage[i];
    }
    // The order in which the search hands out its tasks (64 consecutive queries of qorder each): the ones with the largest
    // boxes of cells first -- a task costs between a few cells and the whole grid, and a workgroup's last tasks should be cheap
    // ones (longest-processing-time-first; the estimate is the box of the task's cells grown by two cells a side).
    const int tasks = (m + 63) / 64;
    int *tord = task_order + (size_t)c * tasks;
    if (tasks > KG_BUILD_THREADS || !lpt) {                     // (the lane-private search's tasks cost the same: no order to find)
        for (int i = t; i < tasks; i += KG_BUILD_THREADS) tord[i] = i;
        return;
    }
    __threadfence_block();
    __syncthreads();
    int *tkey = cnt;                                           // (the cell counters are no longer needed)
    for (int task = t >> 6; task < tasks; task += KG_BUILD_THREADS / 64) {
        const int slot = task * 64 + (t & 63);
        const int j = qorder[(size_t)c * m + (slot < m ? slot : m - 1)];
        const float x = qry[3 * (size_t)j], y = qry[3 * (size_t)j + 1], z = qry[3 * (size_t)j + 2];
        const int cx = kg_cell1(x, g.lo[0], g.ih[0], gx), cy = kg_cell1(y, g.lo[1], g.ih[1], gy), cz = kg_cell1(z, g.lo[2], g.ih[2], gz);
        const int ex = min(gx - 1, kg_wave_max(cx) + 2) - max(0, kg_wave_min(cx) - 2) + 1;
        const int ey = min(gy - 1, kg_wave_max(cy) + 2) - max(0, kg_wave_min(cy) - 2) + 1;
        const int ez = min(gz - 1, kg_wave_max(cz) + 2) - max(0, kg_wave_min(cz) - 2) + 1;
        if ((t & 63) == 0) tkey[task] = ex * ey * ez;
    }
    __syncthreads();
    if (t < tasks) {
        const int key = tkey[t];
        int rank = 0;
        for (int j = 0; j < tasks; ++j) rank += (tkey[j] > key || (tkey[j] == key && j < t)) ? 1 : 0;
        tord[rank] = t;
    }
}

// LDS of the search kernel (dynamic): the sorted cloud as float4 (x, y, z, original index) [n4], n4 = n rounded up to a multiple of 4
// plus one group of padding; the candidate queues (distance [, index]) of THREADS lanes; the cell offsets of the cloud's grid.
__host__ __device__ inline int kg_n4(int n) { return ((n + 3) & ~3) + 4; }
// queue slots per lane in the grid search: (value, index) queues of 16 slots leave room for ONE 512-thread workgroup per CU at
// 2048 points, and two waves per SIMD do not hide the lane-private walk's gathers; ten slots (lists of up to ten entries: k <= 9)
// let two workgroups in
template <int MODE, int S> constexpr int kg_qcap() { return (MODE == 0 && S <= 10) ? 10 : KF_QCAP; }
template <int MODE, int S> __host__ __device__ inline size_t kg_lds_bytes(int n, int threads, int cells = KG_MAX_CELLS) {
    return (size_t)kg_n4(n) * 16 + (size_t)kg_qcap<MODE, S>() * threads * 4 * (MODE == 0 ? 2 : 1) + sizeof(int) * (cells + 4);
}
#ifndef KG_THREADS_V
#define KG_THREADS_V 512
#endif
#ifndef KG_TASKS_PER_WAVE
#define KG_TASKS_PER_WAVE 2
#endif
constexpr int KG_THREADS = KG_THREADS_V;
constexpr int KG_LEFT_CAP = 512;          // queries a workgroup can hold back for the wave-uniform search (more: the redo list)
__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }
#ifdef KG_DIAG                             // diagnostic build only (tools/debug/build_variants.sh): per-wave work counters
__device__ unsigned long long kg_diag[16];
#define KG_COUNT(I, V) do { if ((threadIdx.x & 63) == 0) atomicAdd(&kg_diag[I], (unsigned long long)(V)); } while (0)
#else
#define KG_COUNT(I, V)
#endif

// One workgroup = the whole sorted cloud in LDS + THREADS / 64 waves that pull TASKS (64 consecutive queries of the Hilbert
// order) from a counter in LDS until the workgroup's share [task0, task1) of the cloud's tasks is done: a task costs between
// a few cells and the whole grid, so a fixed assignment leaves a workgroup waiting for its unluckiest wave.
template <int MODE, int S, int THREADS>
__global__ __launch_bounds__(THREADS, (S <= 10 ? 4 : 2)) void knn_grid_kernel(int n, int m, int k, int split, const float4 *__restrict__ sorted,
                                                           const int *__restrict__ cell_start, const int *__restrict__ qorder,
                                                           const KnnGrid *__restrict__ info, const int *__restrict__ task_order,
                                                           const float *__restrict__ xyz2, float *val_out, int *idx_out, int *redo,
                                                           int lane_first) {
    extern __shared__ __attribute__((aligned(16))) float kg_lds[];
    __shared__ int next_task, left_n;
    __shared__ int left_q[KG_LEFT_CAP];
    GA_STAMP(0, 0);
    const int n4 = kg_n4(n);
    float4 *sp = reinterpret_cast<float4 *>(kg_lds);       // the sorted cloud, (x, y, z, original index) per point: ONE ds_read_b128 per candidate
                                                           // whether the address is the wave's (broadcast) or the lane's own (gather)
    float *qd = kg_lds + 4 * (size_t)n4;
    constexpr int QC = kg_qcap<MODE, S>();
    // MODE 2 = knn_point on KEYED lists: a candidate is ONE word, the bits of its squared distance with the low 12 mantissa bits
    // replaced by its position in the sorted cloud (n <= 4096).  Distances are >= +0, so keys order like (truncated distance,
    // position) as unsigned integers, and the whole machinery of the values-only search (one queue word, one med3 per list slot) runs
    // on them.  When a list is finished, the <= S positions it names are re-read, their EXACT distances recomputed and sorted; every
    // candidate the list does not name has a key >= the list's last, i.e. a true distance >= that key's truncated value T, so the
    // k smallest are exact whenever the k-th exact distance is below T (else: the redo list, like ties; ~1 % of the queries of a
    // uniform cloud at k = 9).  Same outputs as the (value, index) lists of MODE 0, at the values-only search's price.
    constexpr bool KEYED = MODE == 2;
    constexpr unsigned KMASK = 0xFFFu, KEMPTY = 0xFFFFFFFFu;
    const float LIST_INIT = KEYED ? __uint_as_float(KEMPTY) : INFINITY;                   // an empty list slot / an open threshold
    // a candidate as the lists hold it (its key, or its distance), and "strictly below the threshold" on that representation
    auto cand_of = [&](const float d, const int pos) { return KEYED ? __uint_as_float((__float_as_uint(d) & ~KMASK) | (unsigned)pos) : d; };
    auto below_thr = [&](const float cnd, const float thr) { return KEYED ? __float_as_uint(cnd) < __float_as_uint(thr) : !(cnd >= thr); };
    // an upper bound of the true distance of a list entry (a key's truncated bits filled up; an empty slot compares false)
    auto upper_of = [&](const float e) { return KEYED ? __uint_as_float(__float_as_uint(e) | KMASK) : e; };
    int *qi = reinterpret_cast<int *>(qd + QC * THREADS);
    int *cs_l = qi + (MODE == 0 ? QC * THREADS : 0);   // the cell offsets too: every shell of every task looks rows up in them
    const int c = blockIdx.y;
    const float4 *pts = sorted + (size_t)c * n;
    const int *cs_g = cell_start + (size_t)c * (KG_MAX_CELLS + 1);
    const KnnGrid g = info[c];
    const int tasks = cdiv_dev(m, 64);
    // this workgroup's tasks: places blockIdx.x, blockIdx.x + split, ... of the cloud's task order (heaviest first, dealt round the
    // cloud's workgroups), taken in that order through the counter
    const int *tord = task_order + (size_t)c * tasks;
    const int task0 = 0, task1 = (tasks - (int)blockIdx.x + split - 1) / split;       // (counted in places of this workgroup)
    if (threadIdx.x == 0) { next_task = task0 + THREADS / 64; left_n = 0; }   // (every wave starts with the place of its own number)
    {   // the cloud: all requests of a thread first, then the LDS writes (one round trip instead of one per 1024 points)
        constexpr int PER = (KG_MAX_N + 4 + THREADS - 1) / THREADS;
        float4 pt[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = threadIdx.x + u * THREADS;
            pt[u] = e < n ? pts[e] : make_float4(INFINITY, INFINITY, INFINITY, 0.f);      // the pad is never admitted (and it is masked)
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = threadIdx.x + u * THREADS;
            if (e < n4) sp[e] = pt[u];
        }
    }
    const int gx = g.g[0], gy = g.g[1], gz = g.g[2];
    for (int e = threadIdx.x; e <= gx * gy * gz; e += THREADS) cs_l[e] = cs_g[e];
    const int *cs = cs_l;
    const int lane = threadIdx.x & 63;
    GA_STAMP(0, 1);
    __syncthreads();                                        // the only workgroup barrier: from here on the waves go their own ways
    GA_STAMP(0, 2);

    // (a task's queries requested one task AHEAD -- index, then coordinates, two dependent round trips that otherwise open every
    // task -- was measured on the shell walk: 280 against 244 us, the task taken ahead is one the counter can no longer balance)
    struct Query { int q; float x, y, z; };
    auto load_query = [&](int place) {
        const int slot = tord[blockIdx.x + place * split] * 64 + lane;
        Query r;
        r.q = qorder[(size_t)c * m + (slot < m ? slot : m - 1)];
        const float *qp = xyz2 + ((size_t)c * m + r.q) * 3;
        r.x = qp[0]; r.y = qp[1]; r.z = qp[2];
        return r;
    };
    // what a lane's finished list leaves behind (a tie among the k + 1 smallest, a NaN, fewer than k finite distances: the redo list)
    auto emit = [&](const int q, const float (&v)[S], const int (&ix)[S], const bool odd, const float qx, const float qy, const float qz) {
        if (KEYED) {
            // exact (distance, original index) of the positions the keys name, sorted by distance (insertion network; ties and the
            // order among them do not matter: a tie inside the first k + 1 sends the query to the redo list)
            float e[S];
            int id[S];
            bool again = odd;
#pragma unroll
            for (int i = 0; i < S; ++i) {
                const unsigned key = __float_as_uint(v[i]);
                const float4 t = sp[key == KEMPTY ? 0 : (int)(key & KMASK)];
                const float dx = t.x - qx, dy = t.y - qy, dz = t.z - qz;
                const float d = (dx * dx + dy * dy) + dz * dz;                       // tf_grouping.py:68, left to right
                e[i] = key == KEMPTY ? INFINITY : d;
                id[i] = key == KEMPTY ? -1 : __float_as_int(t.w);
                again |= d != d;
            }
#pragma unroll
            for (int i = 1; i < S; ++i)
#pragma unroll
                for (int j = i; j > 0; --j) {
                    const bool sw = e[j] < e[j - 1];
                    const float lo2 = sw ? e[j] : e[j - 1], hi2 = sw ? e[j - 1] : e[j];
                    const int loi = sw ? id[j] : id[j - 1], hii = sw ? id[j - 1] : id[j];
                    e[j - 1] = lo2; e[j] = hi2; id[j - 1] = loi; id[j] = hii;
                }
            const unsigned last = __float_as_uint(v[S - 1]);                         // the list's largest key (all ones: the list is not full --
            const float T = __uint_as_float(last & ~KMASK);                          // it then names EVERY candidate met, and nothing bounds from outside)
            float ek = INFINITY;
#pragma unroll
            for (int i = 0; i < S; ++i) if (i == k - 1) ek = e[i];
            again |= last != KEMPTY && !(ek < T);
#pragma unroll
            for (int i = 0; i < S; ++i) if (i == k - 1) again |= id[i] < 0;
#pragma unroll
            for (int i = 0; i + 1 < S; ++i)
                if (i < k && e[i] == e[i + 1] && id[i + 1] >= 0) again = true;
            if (again) redo[1 + atomicAdd(redo, 1)] = c * m + q;
            else {
#pragma unroll
                for (int i = 0; i < S; ++i)
                    if (i < k) { val_out[((size_t)c * m + q) * k + i] = e[i]; idx_out[((size_t)c * m + q) * k + i] = id[i]; }
            }
        } else if (MODE == 0) {
            bool again = odd || ix[k - 1] < 0;
#pragma unroll
            for (int i = 0; i + 1 < S; ++i)
                if (i < k && v[i] == v[i + 1] && ix[i + 1] >= 0) again = true;
            if (again) redo[1 + atomicAdd(redo, 1)] = c * m + q;
            else {
#pragma unroll
                for (int i = 0; i < S; ++i)
                    if (i < k) { val_out[((size_t)c * m + q) * k + i] = v[i]; idx_out[((size_t)c * m + q) * k + i] = ix[i]; }
            }
        } else {
            if (odd || !(v[k - 1] < INFINITY)) redo[1 + atomicAdd(redo, 1)] = c * m + q;
            else {
#pragma unroll
                for (int i = 1; i < S; ++i)
                    if (i < k) val_out[((size_t)c * m + q) * (k - 1) + i - 1] = sqrtf(v[i]);
            }
        }
    };
    // the queues' drain into the lists (see knn_fast_kernel)
    auto drain_into = [&](float (&v)[S], int (&ix)[S], bool &odd, float &thr, unsigned &qw) {
        KG_COUNT(5, 1);
        for (unsigned jo = 4u * threadIdx.x; __any(jo < qw); jo += 4u * THREADS) {
            KG_COUNT(2, 1);
            const bool has = jo < qw;
            const float d = has ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(qd) + jo) : INFINITY;
            int id = 0;
            if (MODE == 0) id = has ? *reinterpret_cast<const int *>(reinterpret_cast<const char *>(qi) + jo) : 0;
            if (!KEYED) odd |= d != d;
            if (KEYED) {                                   // keys: the same medians on unsigned integers (v_med3_u32)
                const unsigned x = has ? __float_as_uint(d) : KEMPTY;
#pragma unroll
                for (int i = S - 1; i > 0; --i) {
                    const unsigned a_ = __float_as_uint(v[i - 1]), b_ = __float_as_uint(v[i]);
                    v[i] = __uint_as_float(max(min(a_, x), min(max(a_, x), b_)));
                }
                v[0] = __uint_as_float(min(__float_as_uint(v[0]), x));
            } else if (MODE == 1) {                        // values only: sorted insertion is a median per slot, v[i] <- med3(v[i - 1], d, v[i])
#pragma unroll                                             // from the top down (the OLD v[i - 1]); half the instructions of a compare-exchange chain
                for (int i = S - 1; i > 0; --i) v[i] = __builtin_amdgcn_fmed3f(v[i - 1], d, v[i]);
                v[0] = fminf(v[0], d);                     // (a NaN only ever marks the query `odd`: its list is not used)
            } else {                                       // (value, index): the candidate goes in front of the first entry it is STRICTLY below
                bool below[S];                             // (an equal entry stays ahead of it); `below` is monotone along the sorted list
#pragma unroll
                for (int i = 0; i < S; ++i) below[i] = d < v[i];
#pragma unroll
                for (int i = S - 1; i > 0; --i) {
                    v[i] = __builtin_amdgcn_fmed3f(v[i - 1], d, v[i]);
                    ix[i] = below[i - 1] ? ix[i - 1] : (below[i] ? id : ix[i]);
                }
                v[0] = fminf(v[0], d);
                ix[0] = below[0] ? id : ix[0];
            }
        }
        qw = 4u * threadIdx.x;
        thr = v[S - 1];
    };

    // ---- wave-uniform search: the wave's box of cells, shell by shell, until every (live) lane's list is closed ----
    auto uniform_search = [&](const int q, const float qx, const float qy, const float qz, const bool live) {
        const int ccx = kg_cell1(qx, g.lo[0], g.ih[0], gx), ccy = kg_cell1(qy, g.lo[1], g.ih[1], gy), ccz = kg_cell1(qz, g.lo[2], g.ih[2], gz);
        // the wave's box of cells
        const int bx0 = kg_wave_min(ccx), bx1 = kg_wave_max(ccx), by0 = kg_wave_min(ccy), by1 = kg_wave_max(ccy);
        const int bz0 = kg_wave_min(ccz), bz1 = kg_wave_max(ccz);
        float v[S];
        int ix[S];
#pragma unroll
        for (int i = 0; i < S; ++i) { v[i] = LIST_INIT; ix[i] = -1; }
        bool odd = false;
        float thr = LIST_INIT;
        unsigned qw = 4u * threadIdx.x;
        // the points [lo, hi) of the sorted cloud: wave-uniform bounds, broadcast LDS reads, four points a step (the ones past hi
        // belong to other rows: masked, or a point would enter a list twice)
        auto run = [&](int lo, int hi) {
            KG_COUNT(0, hi - lo);
            for (int p = lo; p < hi; p += 4) {
                KG_COUNT(3, 1);
                float4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = sp[p + u];                        // (reads past hi stay inside the padded array and are masked)
                float d[4];
                bool adm[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float dx = t[u].x - qx, dy = t[u].y - qy, dz = t[u].z - qz;
                    d[u] = cand_of((dx * dx + dy * dy) + dz * dz, p + u);            // tf_grouping.py:68, left to right
                    adm[u] = (p + u < hi) && below_thr(d[u], thr);
                }
                if (__any(adm[0] | adm[1] | adm[2] | adm[3])) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (adm[u]) {
                            *reinterpret_cast<float *>(reinterpret_cast<char *>(qd) + qw) = d[u];
                            if (MODE == 0) *reinterpret_cast<int *>(reinterpret_cast<char *>(qi) + qw) = __float_as_int(t[u].w);
                            qw += 4u * THREADS;
                        }
                    if (__any(qw >= 4u * (QC - 3) * THREADS)) drain_into(v, ix, odd, thr, qw);
                }
            }
        };

        int px0 = 1, px1 = 0, py0 = 1, py1 = 0, pz0 = 1, pz1 = 0;              // the box visited so far (empty)
        bool fin = !live;
        for (int s = 0;; ++s) {
            const int x0 = max(0, bx0 - s), x1 = min(gx - 1, bx1 + s), y0 = max(0, by0 - s), y1 = min(gy - 1, by1 + s);
            const int z0 = max(0, bz0 - s), z1 = min(gz - 1, bz1 + s);
            // rows of the shell: lane r of a batch looks up the bounds of row r (all rows of a batch in one round trip)
            const int ny = y1 - y0 + 1, rows = ny * (z1 - z0 + 1);
            for (int r0 = 0; r0 < rows; r0 += 64) {
                const int r = r0 + lane;
                int a_lo = 0, a_hi = 0, b_lo = 0, b_hi = 0;
                if (r < rows) {
                    const int z = z0 + r / ny, y = y0 + r % ny;
                    const int *row = cs + (z * gy + y) * gx;
                    const bool seen = z >= pz0 && z <= pz1 && y >= py0 && y <= py1;     // the middle of this row was walked in an earlier shell
                    if (!seen) { a_lo = row[x0]; a_hi = row[x1 + 1]; }
                    else {
                        if (x0 < px0) { a_lo = row[x0]; a_hi = row[px0]; }
                        if (x1 > px1) { b_lo = row[px1 + 1]; b_hi = row[x1 + 1]; }
                    }
                }
                const int cnt_rows = min(64, rows - r0);
                for (int i = 0; i < cnt_rows; ++i) {
                    const int alo = __builtin_amdgcn_readlane(a_lo, i), ahi = __builtin_amdgcn_readlane(a_hi, i);
                    const int blo = __builtin_amdgcn_readlane(b_lo, i), bhi = __builtin_amdgcn_readlane(b_hi, i);
                    run(alo, ahi);
                    run(blo, bhi);
                }
            }
            drain_into(v, ix, odd, thr, qw);
            // finished lanes: every point not met yet lies beyond a face of the box, i.e. at least `mg` away
            float mg = INFINITY;
            if (x0 > 0) mg = fminf(mg, (qx - (g.lo[0] + (float)x0 * g.h[0])) * 0.999f - g.eps[0]);
            if (x1 < gx - 1) mg = fminf(mg, ((g.lo[0] + (float)(x1 + 1) * g.h[0]) - qx) * 0.999f - g.eps[0]);
            if (y0 > 0) mg = fminf(mg, (qy - (g.lo[1] + (float)y0 * g.h[1])) * 0.999f - g.eps[1]);
            if (y1 < gy - 1) mg = fminf(mg, ((g.lo[1] + (float)(y1 + 1) * g.h[1]) - qy) * 0.999f - g.eps[1]);
            if (z0 > 0) mg = fminf(mg, (qz - (g.lo[2] + (float)z0 * g.h[2])) * 0.999f - g.eps[2]);
            if (z1 < gz - 1) mg = fminf(mg, ((g.lo[2] + (float)(z1 + 1) * g.h[2]) - qz) * 0.999f - g.eps[2]);
            fin = fin || (mg > 0.f && upper_of(v[S - 1]) < mg * mg);
            const bool whole = x0 == 0 && y0 == 0 && z0 == 0 && x1 == gx - 1 && y1 == gy - 1 && z1 == gz - 1;
            KG_COUNT(1, 1); KG_COUNT(6, rows);
            if (whole || !__any(!fin)) { KG_COUNT(4, 1); KG_COUNT(7, whole ? 1 : 0); break; }
            px0 = x0; px1 = x1; py0 = y0; py1 = y1; pz0 = z0; pz1 = z1;
        }
        if (live) emit(q, v, ix, odd, qx, qy, qz);
    };

    // ---- lane-private search (round 4, second form): every lane walks the 3 x 3 x 3 cells around ITS query -- nine runs of the
    // sorted cloud, gathered from LDS at per-lane addresses -- and is finished if that closes its list (the same face-distance test,
    // on the lane's own block).  The 64 queries of a task share a neighbourhood, but the UNION of their neighbourhoods, which the
    // wave-uniform search walks for all of them, is several times one lane's 27 cells (at 2048 points in 8^3 cells: ~110 candidates
    // a lane against 500-800 for the wave).  Lanes it does not finish (a sparse corner, an outlier) go to a list in LDS and are
    // searched the wave-uniform way, 64 at a time, when the workgroup's tasks are done.  Returns true where the lane is done.
    auto lane_search = [&](const int q, const float qx, const float qy, const float qz, const bool live) -> bool {
        const int ccx = kg_cell1(qx, g.lo[0], g.ih[0], gx), ccy = kg_cell1(qy, g.lo[1], g.ih[1], gy), ccz = kg_cell1(qz, g.lo[2], g.ih[2], gz);
        const int x0 = max(ccx - 1, 0), x1 = min(ccx + 1, gx - 1), y0 = max(ccy - 1, 0), y1 = min(ccy + 1, gy - 1);
        const int z0 = max(ccz - 1, 0), z1 = min(ccz + 1, gz - 1);
        int rlo[9], rhi[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int z = ccz + r / 3 - 1, y = ccy + r % 3 - 1;
            const bool in = live && z >= 0 && z < gz && y >= 0 && y < gy;
            const int row = in ? (z * gy + y) * gx : 0;
            rlo[r] = in ? cs[row + x0] : 0;
            rhi[r] = in ? cs[row + x1 + 1] : 0;
        }
        float v[S];
        int ix[S];
#pragma unroll
        for (int i = 0; i < S; ++i) { v[i] = LIST_INIT; ix[i] = -1; }
        bool odd = false;
        float thr = LIST_INIT;
        unsigned qw = 4u * threadIdx.x;
        // Between two drains `thr` is stale, and a stale threshold admits most of what comes (the 9th of the first 16 candidates
        // lets every second one through): the queues filled 13 times a task and the drains cost twice the walk.  So a lane
        // tightens its own threshold as it pushes: once S values below `thr` are queued, the S-th smallest seen so far is at most
        // their maximum -- that becomes `thr` (the drain then sets it exactly).  Rows nearest first (own row, the four that
        // share a face with it, the corners): the first S pushes are already close.
        float pmax = 0.f;
        int pcnt = 0;
        constexpr int ORDER[9] = {4, 1, 3, 5, 7, 0, 2, 6, 8};
#pragma unroll
        for (int ro = 0; ro < 9; ++ro) {
            const int r = ORDER[ro];
            int p = rlo[r];
            const int hi = rhi[r];
            while (__any(p < hi)) {                          // four candidates a step: the step's gathers are in flight together
                KG_COUNT(9, 1);
                float4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = sp[min(p + u, n4 - 1)];
                float d[4];
                bool adm[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float dx = t[u].x - qx, dy = t[u].y - qy, dz = t[u].z - qz;
                    d[u] = cand_of((dx * dx + dy * dy) + dz * dz, p + u);            // tf_grouping.py:68, left to right
                    adm[u] = (p + u < hi) && below_thr(d[u], thr);
                }
                if (__any(adm[0] | adm[1] | adm[2] | adm[3])) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (adm[u]) {
                            *reinterpret_cast<float *>(reinterpret_cast<char *>(qd) + qw) = d[u];
                            if (MODE == 0) *reinterpret_cast<int *>(reinterpret_cast<char *>(qi) + qw) = __float_as_int(t[u].w);
                            qw += 4u * THREADS;
                            pmax = KEYED ? __uint_as_float(max(__float_as_uint(pmax), __float_as_uint(d[u]))) : fmaxf(pmax, d[u]);
                            pcnt += 1;
                        }
                    if (pcnt >= S) { thr = KEYED ? __uint_as_float(min(__float_as_uint(thr), __float_as_uint(pmax))) : fminf(thr, pmax); pmax = 0.f; pcnt = 0; }
                    if (__any(qw >= 4u * (QC - 3) * THREADS)) { drain_into(v, ix, odd, thr, qw); pmax = 0.f; pcnt = 0; }
                }
                p += 4;
            }
        }
        drain_into(v, ix, odd, thr, qw);
        float mg = INFINITY;
        if (x0 > 0) mg = fminf(mg, (qx - (g.lo[0] + (float)x0 * g.h[0])) * 0.999f - g.eps[0]);
        if (x1 < gx - 1) mg = fminf(mg, ((g.lo[0] + (float)(x1 + 1) * g.h[0]) - qx) * 0.999f - g.eps[0]);
        if (y0 > 0) mg = fminf(mg, (qy - (g.lo[1] + (float)y0 * g.h[1])) * 0.999f - g.eps[1]);
        if (y1 < gy - 1) mg = fminf(mg, ((g.lo[1] + (float)(y1 + 1) * g.h[1]) - qy) * 0.999f - g.eps[1]);
        if (z0 > 0) mg = fminf(mg, (qz - (g.lo[2] + (float)z0 * g.h[2])) * 0.999f - g.eps[2]);
        if (z1 < gz - 1) mg = fminf(mg, ((g.lo[2] + (float)(z1 + 1) * g.h[2]) - qz) * 0.999f - g.eps[2]);
        const bool whole = x0 == 0 && y0 == 0 && z0 == 0 && x1 == gx - 1 && y1 == gy - 1 && z1 == gz - 1;
        const bool fin = whole || (mg > 0.f && upper_of(v[S - 1]) < mg * mg);
        KG_COUNT(8, 1); KG_COUNT(10, __popcll(__ballot(live && !fin)));
        if (live && fin) emit(q, v, ix, odd, qx, qy, qz);
        return fin || !live;
    };

    int task = task0 + (threadIdx.x >> 6);
    Query cur = load_query(task < task1 ? task : task0);
    while (task < task1) {
        const bool live = tord[blockIdx.x + task * split] * 64 + lane < m;
        if (KEYED && g.bad) {                               // a non-finite coordinate in the cloud: NaN distances must reach the reference-order
            if (live) redo[1 + atomicAdd(redo, 1)] = c * m + cur.q;   // kernel, and a key would hide them behind the threshold
        } else if (lane_first) {
            const bool done = lane_search(cur.q, cur.x, cur.y, cur.z, live);
            const unsigned long long nb = __ballot(!done);
            if (nb) {                                       // the lanes left over: into the workgroup's list (overflow: the redo list, exact as well)
                int base = 0;
                if (lane == 0) base = atomicAdd(&left_n, __popcll(nb));
                base = __builtin_amdgcn_readfirstlane(base);
                const int pos = base + __popcll(nb & ((1ull << lane) - 1ull));
                if (!done) {
                    if (pos < KG_LEFT_CAP) left_q[pos] = cur.q;
                    else redo[1 + atomicAdd(redo, 1)] = c * m + cur.q;
                }
            }
        } else uniform_search(cur.q, cur.x, cur.y, cur.z, live);
        int nt = 0;
        if (lane == 0) nt = atomicAdd(&next_task, 1);
        nt = __builtin_amdgcn_readfirstlane(nt);
        cur = load_query(nt < task1 ? nt : task);
        task = nt;
    }
    if (lane_first) {
        // The leftovers, 64 at a time.  They are scattered over the cloud, so their common box of cells is the grid: an all-points scan.
        // Left to ONE wave that is ~100 K instructions behind which the other waves idle (measured: it doubled the kernel's time for a
        // handful of leftover queries per workgroup); so every wave scans an EIGHTH of the cloud for the same 64 queries (broadcast
        // reads, the lists and queues as everywhere), leaves its list in its queue slots, and wave 0 merges the eight lists.
        __syncthreads();                                    // every wave's leftovers are listed
        const int left = min(left_n, KG_LEFT_CAP);
        constexpr bool COOP = S <= QC;                 // (a list must fit the lane's queue slots; longer ones: one wave, the shell walk)
        const int wave = threadIdx.x >> 6;
        for (int chunk = 0; chunk * 64 < left; ++chunk) {
            const int slot = chunk * 64 + lane;
            const bool lv = slot < left;
            const int q = left_q[lv ? slot : chunk * 64];   // (a dead lane repeats the chunk's first query)
            const float *qp = xyz2 + ((size_t)c * m + q) * 3;
            const float qx = qp[0], qy = qp[1], qz = qp[2];
            KG_COUNT(11, 1);
            if (!COOP) {
                if (chunk % (THREADS / 64) == wave) uniform_search(q, qx, qy, qz, lv);
                continue;
            }
            float v[S];
            int ix[S];
#pragma unroll
            for (int i = 0; i < S; ++i) { v[i] = LIST_INIT; ix[i] = -1; }
            bool odd = false;
            float thr = LIST_INIT;
            unsigned qw = 4u * threadIdx.x;
            const int per = (((n + THREADS / 64 - 1) / (THREADS / 64)) + 3) & ~3;
            const int lo = min(n, wave * per), hi = min(n, lo + per);
            for (int p = lo; p < hi; p += 4) {
                float4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = sp[p + u];
                float d[4];
                bool adm[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float dx = t[u].x - qx, dy = t[u].y - qy, dz = t[u].z - qz;
                    d[u] = cand_of((dx * dx + dy * dy) + dz * dz, p + u);            // tf_grouping.py:68, left to right
                    adm[u] = (p + u < hi) && below_thr(d[u], thr);
                }
                if (__any(adm[0] | adm[1] | adm[2] | adm[3])) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (adm[u]) {
                            *reinterpret_cast<float *>(reinterpret_cast<char *>(qd) + qw) = d[u];
                            if (MODE == 0) *reinterpret_cast<int *>(reinterpret_cast<char *>(qi) + qw) = __float_as_int(t[u].w);
                            qw += 4u * THREADS;
                        }
                    if (__any(qw >= 4u * (QC - 3) * THREADS)) drain_into(v, ix, odd, thr, qw);
                }
            }
            drain_into(v, ix, odd, thr, qw);
            if (wave > 0) {                                 // publish: list -> the lane's own queue slots, NaN flag -> slot 0's sign of life
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    qd[i * THREADS + threadIdx.x] = (i == 0 && odd) ? NAN : v[i];
                    if (MODE == 0) qi[i * THREADS + threadIdx.x] = ix[i];
                }
            }
            __syncthreads();
            if (wave == 0) {
                for (int w = 1; w < THREADS / 64; ++w) {
#pragma unroll
                    for (int i = 0; i < S; ++i) {
                        const float d = qd[i * THREADS + w * 64 + lane];
                        if (!KEYED) odd |= d != d;
                        if (MODE != 0) {
                            unsigned x = __float_as_uint(d);
#pragma unroll
                            for (int j = 0; j < S; ++j) {
                                const unsigned vj = __float_as_uint(v[j]);
                                const unsigned lo2 = min(vj, x);
                                x = max(vj, x);
                                v[j] = __uint_as_float(lo2);
                            }
                        } else {
                            float x = d;
                            int xi = qi[i * THREADS + w * 64 + lane];
#pragma unroll
                            for (int j = 0; j < S; ++j) {
                                const bool cc = x < v[j];
                                const float lo2 = cc ? x : v[j], hi2 = cc ? v[j] : x;
                                const int loi = cc ? xi : ix[j], hii = cc ? ix[j] : xi;
                                v[j] = lo2; ix[j] = loi; x = hi2; xi = hii;
                            }
                        }
                    }
                }
                if (lv) emit(q, v, ix, odd, qx, qy, qz);
            }
            __syncthreads();                                // the queues are free again
        }
    }
    GA_STAMP(0, 7);
}

// ------------------------------------------------------------------------------------------
// QueryBallPoint (tf_grouping_g.cu:3-36): the FIRST nsample dataset points with
// max(sqrt(d2), 1e-20) < radius, padded with the first hit; pts_cnt = number found.
// One THREAD per query (256 queries per workgroup, the dataset streams through LDS): a candidate costs the nine distance
// instructions and one compare.
// ------------------------------------------------------------------------------------------
// `max(sqrtf(d2), 1e-20f) < radius` is decided WITHOUT the square root: sqrtf is correctly rounded and monotone, so
// the host finds the largest float t2max with sqrtf(t2max) < radius once, and the test is d2 <= t2max (NaN fails both).
constexpr int QB_THREADS = 256;
constexpr int QB_TILE = 1024;
__global__ __launch_bounds__(QB_THREADS) void query_ball_fast_kernel(int n, int m, float t2max, int nsample, const float *xyz1,
                                                                     const float *xyz2, int *idx, int *pts_cnt) {
    __shared__ __attribute__((aligned(16))) float sx[QB_TILE], sy[QB_TILE], sz[QB_TILE];
    const int c = blockIdx.y;
    const float *data = xyz1 + (size_t)c * n * 3;
    const int q = blockIdx.x * QB_THREADS + threadIdx.x;
    const bool live = q < m;
    const float *qp = xyz2 + ((size_t)c * m + (live ? q : m - 1)) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    int *row = idx + ((size_t)c * m + (live ? q : m - 1)) * nsample;
    int cnt = live ? 0 : nsample, first = -1;               // (a dead lane counts as finished)
    for (int t0 = 0; t0 < n; t0 += QB_TILE) {
        if (__syncthreads_and(cnt >= nsample)) break;        // every query of the workgroup has its nsample hits
        const int tn = min(QB_TILE, n - t0);
        for (int e = threadIdx.x; e < QB_TILE; e += QB_THREADS) {
            const bool in = e < tn;                          // the pad never hits: NaN compares false
            sx[e] = in ? data[3 * (size_t)(t0 + e)] : NAN; sy[e] = in ? data[3 * (size_t)(t0 + e) + 1] : NAN;
            sz[e] = in ? data[3 * (size_t)(t0 + e) + 2] : NAN;
        }
        __syncthreads();
        for (int e0 = 0; e0 < tn; e0 += 4) {
            const float4 xa = *reinterpret_cast<const float4 *>(&sx[e0]);
            const float4 ya = *reinterpret_cast<const float4 *>(&sy[e0]);
            const float4 za = *reinterpret_cast<const float4 *>(&sz[e0]);
            const float tx[4] = {xa.x, xa.y, xa.z, xa.w}, ty[4] = {ya.x, ya.y, ya.z, ya.w}, tz[4] = {za.x, za.y, za.z, za.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float dx = qx - tx[u], dy = qy - ty[u], dz = qz - tz[u];
                const float d2 = (dx * dx + dy * dy) + dz * dz;
                if (d2 <= t2max && cnt < nsample) {
                    if (first < 0) first = t0 + e0 + u;
                    row[cnt++] = t0 + e0 + u;
                }
            }
        }
    }
    if (!live) return;
    if (first >= 0)
        for (int l = cnt; l < nsample; ++l) row[l] = first;  // pad with the first hit
    if (pts_cnt) pts_cnt[(size_t)c * m + q] = cnt;
}

// GroupPoint gather (tf_grouping_g.cu:40-57): out[b,j,k,:] = points[b, idx[b,j,k], :]
__global__ void group_point_kernel(int n, int cch, size_t per_cloud, size_t total, const float *points, const int *idx,
                                   float *out) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;    // over (b, j, k, channel)
    if (e >= total) return;
    const size_t entry = e / cch;
    const int ch = (int)(e % cch);
    const size_t cloud = entry / per_cloud;
    out[e] = points[(cloud * n + idx[entry]) * cch + ch];
}

// GroupPointGrad (tf_grouping_g.cu:61-78) without float atomics and in the CPU twin's accumulation order
// (test/query_ball_point.cpp:70-84: entries ascending), in time linear in the number of entries: a stable LSD radix sort
// of the entries by destination point, six bits per pass, then one in-order sum per (point, channel).
// One pass = count, scan, place.  A WAVE owns one segment of the (current order of the) entries and all 64 digit values,
// lane = digit: it walks its segment 64 entries at a time; six ballots of the digit's bits give every lane, without LDS,
// both the entries whose digit is the lane's own (count / cursor advance) and, as an entry, its peers with the same digit
// in the group -- its slot is cursor[digit] + (peers in lower lanes).  Counts are laid out digit-major, segment-minor, so
// one exclusive scan yields every (digit, segment) cursor and the order inside a digit stays the entry order: stable.
// No atomics anywhere: the result does not depend on scheduling.  Invalid destinations sort behind the last point and
// are never summed.  Round 1 let every (point, channel) thread walk every entry: O(n c m nsample) per cloud.
constexpr int GPG_WAVES = 4, GPG_AHEAD = 4;

template <bool PLACE>
__global__ __launch_bounds__(64 * GPG_WAVES) void gpg_pass_kernel(int n, int entries, int shift, int segs, int seg_len,
                                                                  const int *keys_in, const int *perm_in, int *table,
                                                                  int *keys_out, int *perm_out) {
    const int c = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int seg = blockIdx.x * GPG_WAVES + (threadIdx.x >> 6);
    if (seg >= segs) return;                                 // (no workgroup barrier below: waves are independent)
    const int *kin = keys_in + (size_t)c * entries;
    const int *pin = perm_in ? perm_in + (size_t)c * entries : nullptr;       // first pass: the identity
    int *slot = table + (size_t)c * (64 * segs + 1) + lane * segs + seg;
    const int lo = seg * seg_len, hi = min(entries, lo + seg_len);
    int cursor = PLACE ? *slot : 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int i0 = lo; i0 < hi; i0 += 64 * GPG_AHEAD) {
        int key[GPG_AHEAD], src[GPG_AHEAD];
#pragma unroll
        for (int u = 0; u < GPG_AHEAD; ++u) {                // the loads of several groups in flight together
            const int i = i0 + u * 64 + lane;
            key[u] = 0; src[u] = i;
            if (i < hi) {
                key[u] = kin[i];
                if (PLACE && pin) src[u] = pin[i];
            }
        }
#pragma unroll
        for (int u = 0; u < GPG_AHEAD; ++u) {
            const int i = i0 + u * 64 + lane;
            const bool in = i < hi;
            int k = key[u];
            if (!pin) k = (k < 0 || k >= n) ? n : k;         // first pass: raw indices; invalid ones behind the last point
            const int digit = (k >> shift) & 63;
            const unsigned long long any = __ballot(in);
            if (any == 0) break;
            unsigned long long to_me = any, peers = any;
#pragma unroll
            for (int bit = 0; bit < 6; ++bit) {
                const unsigned long long set = __ballot(in && ((digit >> bit) & 1));
                to_me &= ((lane >> bit) & 1) ? set : ~set;
                if (PLACE) peers &= ((digit >> bit) & 1) ? set : ~set;
            }
            if (PLACE) {
                const int base = __shfl(cursor, digit);      // (every lane takes part in the permute)
                if (in) {
                    const size_t o = (size_t)c * entries + base + __popcll(peers & below);
                    keys_out[o] = k;
                    perm_out[o] = src[u];
                }
            }
            cursor += __popcll(to_me);
        }
    }
    if (!PLACE) *slot = cursor;
}

// in place: cnt[c][0..n) -> start[c][0..n], start[c][n] = total
__global__ __launch_bounds__(1024) void gpg_scan_kernel(int n, int *cnt) {
    __shared__ int wsum[16];
    __shared__ int carry;
    int *row = cnt + (size_t)blockIdx.x * (n + 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int p0 = 0; p0 < n; p0 += 1024) {
        const int p = p0 + threadIdx.x;
        const int v = p < n ? row[p] : 0;
        int inc = v;                                        // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (p < n) row[p] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) row[n] = carry;
}

// start[c][p] = first position of point p in the sorted destinations (p = 0 .. n; start[n] = number of valid entries)
__global__ __launch_bounds__(256) void gpg_bounds_kernel(int n, int entries, const int *sorted_keys, int *start) {
    const int c = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p > n) return;
    const int *k = sorted_keys + (size_t)c * entries;
    int lo = 0, hi = entries;                               // lower bound of p
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (k[mid] < p) lo = mid + 1; else hi = mid;
    }
    start[(size_t)c * (n + 1) + p] = lo;
}

__global__ __launch_bounds__(256) void gpg_sum_kernel(int n, int cch, int entries, const float *grad_out, const int *start,
                                                      const int *perm, float *grad_points) {
    const int c = blockIdx.y;
    const int pc = blockIdx.x * 256 + threadIdx.x;     // (point, channel) pair of this cloud
    if (pc >= n * cch) return;
    const int p = pc / cch, ch = pc % cch;
    const int lo = start[(size_t)c * (n + 1) + p], hi = start[(size_t)c * (n + 1) + p + 1];
    const int *pm = perm + (size_t)c * entries;
    const float *go = grad_out + (size_t)c * entries * cch;
    float acc = 0.f;
    int i = lo;
    for (; i + 4 <= hi; i += 4) {                      // four rows in flight; the sum stays in entry order
        const int e0 = pm[i], e1 = pm[i + 1], e2 = pm[i + 2], e3 = pm[i + 3];
        const float v0 = go[(size_t)e0 * cch + ch], v1 = go[(size_t)e1 * cch + ch], v2 = go[(size_t)e2 * cch + ch], v3 = go[(size_t)e3 * cch + ch];
        acc += v0; acc += v1; acc += v2; acc += v3;
    }
    for (; i < hi; ++i) acc += go[(size_t)pm[i] * cch + ch];
    grad_points[((size_t)c * n + p) * cch + ch] = acc;
}

}  // namespace geoadv

using namespace geoadv;

static int row_lds_attr() {
    static DeviceOnce once;
    return once.run([]() -> int {
        const int cap = ROW_MAX_N * 8;
        GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(selection_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_redo_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_redo_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        return GEOADV_OK;
    });
}

extern "C" int geoadv_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out, void *stream) {
    GA_REQUIRE(b >= 0 && n >= 0 && m >= 0, "selection_sort: negative dimension");
    GA_REQUIRE(k > 0, "SelectionSort expects positive k");                         // tf_grouping.cpp:112-113
    GA_REQUIRE(n <= ROW_MAX_N, "selection_sort: rows longer than %d are not supported (n=%d)", ROW_MAX_N, n);
    const size_t rows = (size_t)b * m;
    if (rows == 0 || n == 0) return GEOADV_OK;
    GA_REQUIRE(dist && outi && out, "selection_sort: null pointer");
    if (int rc = row_lds_attr()) return rc;
    const unsigned grid = (unsigned)(rows < 65535u * 16u ? rows : 65535u * 16u);
    selection_sort_kernel<<<grid, 64, (size_t)n * 8, as_stream(stream)>>>(n, k, rows, dist, outi, out);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

// Which kernel answers a k-NN call (k <= 16): GEOADV_KNN_AUTO = by size (datasets of >= 512 points: the exact grid search,
// smaller ones: the all-points kernel), _ALL_POINTS, _GRID = the grid search at every size, _GRID_SHELLS = as _GRID without the
// lane-private first pass.  Per call (geoadv_knn_*_ws); the reference-shaped entry points use the process default below.
static std::atomic<int> g_knn_default_mode{GEOADV_KNN_AUTO};   // geoadv_knn_grid_mode: tests and measurements only

static bool knn_uses_grid(int kmode, int n, int m) {
    return n <= KG_MAX_N && (kmode >= GEOADV_KNN_GRID || (kmode == GEOADV_KNN_AUTO && n >= KNN_GRID_MIN_N && m >= 64));
}
static size_t knn_up(size_t v) { return (v + 255) / 256 * 256; }
// caller-owned scratch of the list kernels: the redo list (counter + one entry per query at most) and, for the grid search, the
// sorted dataset (+ 4 entries of padding: the scalar loads read four points at a time), cell offsets, query order, grid
static size_t knn_fast_scratch_bytes(int b, int n, int m) {
    const size_t redo_b = sizeof(int) * ((size_t)b * m + 1), sorted_b = sizeof(float4) * ((size_t)b * n + 4);
    const size_t cs_b = sizeof(int) * (size_t)b * (KG_MAX_CELLS + 1), qo_b = sizeof(int) * (size_t)b * m, info_b = sizeof(KnnGrid) * (size_t)b;
    const size_t to_b = sizeof(int) * (size_t)b * cdiv(m, 64);
    return knn_up(redo_b) + knn_up(sorted_b) + knn_up(cs_b) + knn_up(qo_b) + knn_up(info_b) + knn_up(to_b);
}

// kmode: GEOADV_KNN_*; scratch: knn_fast_scratch_bytes(b, n, m) bytes, 256-byte aligned
// (lane-private 27-cell walk first -- values-only lists (knn_dists) only: with (value, index) lists the walk lost to the shells at
// either occupancy: knn_point(8) at 256 x 2048: 378 against 343 us at one workgroup per CU, 265 against 227 us at two)
template <int MODE, int S>
static int launch_knn_fast(int kmode, int b, int n, int m, int k, const float *xyz1, const float *xyz2, float *val, int *idx, char *scratch,
                           hipStream_t st) {
    const bool grid = knn_uses_grid(kmode, n, m);
    const size_t redo_b = sizeof(int) * ((size_t)b * m + 1), sorted_b = sizeof(float4) * ((size_t)b * n + 4);
    const size_t cs_b = sizeof(int) * (size_t)b * (KG_MAX_CELLS + 1), qo_b = sizeof(int) * (size_t)b * m, info_b = sizeof(KnnGrid) * (size_t)b;
    auto up = knn_up;
    int *redo = reinterpret_cast<int *>(scratch);
    if (!grid) GA_HIP(hipMemsetAsync(redo, 0, sizeof(int), st));          // (the grid's build kernel clears the counter itself: one launch less)
    if (grid) {
        float4 *sorted = reinterpret_cast<float4 *>(scratch + up(redo_b));
        int *cs = reinterpret_cast<int *>(scratch + up(redo_b) + up(sorted_b));
        int *qo = reinterpret_cast<int *>(scratch + up(redo_b) + up(sorted_b) + up(cs_b));
        KnnGrid *info = reinterpret_cast<KnnGrid *>(scratch + up(redo_b) + up(sorted_b) + up(cs_b) + up(qo_b));
        int *tord = reinterpret_cast<int *>(scratch + up(redo_b) + up(sorted_b) + up(cs_b) + up(qo_b) + up(info_b));
        const int G = n < KNN_GRID_N_MID ? KNN_GRID_G_SMALL : (n < KNN_GRID_N_BIG ? KNN_GRID_G_MID : KNN_GRID_G_BIG);
        // knn_point (MODE 0): the keyed lists (kernel MODE 2) on the lane-private walk, unless the caller asks for the shell walk alone
        // (GEOADV_KNN_GRID_SHELLS): that one keeps the (value, index) lists, as the parity tests' second opinion
        const bool keyed = MODE == 0 && kmode != GEOADV_KNN_GRID_SHELLS;
        const int lane_first = ((MODE == 1 || keyed) && kmode != GEOADV_KNN_GRID_SHELLS) ? 1 : 0;
        constexpr size_t KB_LDS = (size_t)KG_MAX_N * 20;
        static DeviceOnce battr;
        if (int rc = battr.run([]() -> int {
                GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_grid_build_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)KB_LDS));
                return GEOADV_OK;
            })) return rc;
        knn_grid_build_kernel<<<b, KG_BUILD_THREADS, KB_LDS, st>>>(n, m, G, xyz1, xyz2, sorted, cs, qo, info, tord, lane_first ? 0 : 1, redo);
        constexpr int KMODE = MODE == 0 ? 2 : MODE;          // the keyed instantiation of this list length
        static DeviceOnce attr;
        if (int rc = attr.run([]() -> int {
                GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_grid_kernel<MODE, S, KG_THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)kg_lds_bytes<MODE, S>(KG_MAX_N, KG_THREADS)));
                GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_grid_kernel<KMODE, S, KG_THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)kg_lds_bytes<KMODE, S>(KG_MAX_N, KG_THREADS)));
                return GEOADV_OK;
            })) return rc;
        // workgroups per cloud: every wave should see KG_TASKS_PER_WAVE tasks or more (the counter balances them), and the launch
        // about two workgroups per CU
        const int tasks = cdiv(m, 64), waves = KG_THREADS / 64;
        const int split = std::max(1, std::min(std::max(1, tasks / (waves * KG_TASKS_PER_WAVE)), cdiv(2 * kCUs, b)));
        if (keyed)
            knn_grid_kernel<KMODE, S, KG_THREADS><<<dim3(split, b), KG_THREADS, kg_lds_bytes<KMODE, S>(n, KG_THREADS, G * G * G), st>>>(n, m, k, split, sorted, cs, qo,
                                                                                                                         info, tord, xyz2, val, idx, redo, lane_first);
        else
            knn_grid_kernel<MODE, S, KG_THREADS><<<dim3(split, b), KG_THREADS, kg_lds_bytes<MODE, S>(n, KG_THREADS, G * G * G), st>>>(n, m, k, split, sorted, cs, qo, info,
                                                                                                                       tord, xyz2, val, idx, redo, lane_first);
    } else {
        knn_fast_kernel<MODE, S><<<dim3(cdiv(m, KF_THREADS), b), KF_THREADS, 0, st>>>(n, m, k, xyz1, xyz2, val, idx, redo);
    }
    knn_redo_kernel<MODE><<<1024, 64, (size_t)n * 8, st>>>(n, m, k, xyz1, xyz2, val, idx, redo);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

#ifdef KG_DIAG
extern "C" int geoadv_debug_knn_occupancy(int n, int *blocks256, int *blocks512, int *blocks_fast) {
    *blocks256 = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks512, knn_grid_kernel<1, 9, KG_THREADS>, KG_THREADS, kg_lds_bytes<1, 9>(n, KG_THREADS)) != hipSuccess) return 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_fast, knn_fast_kernel<1, 9>, 256, 0) != hipSuccess) return 1;
    return 0;
}
extern "C" int geoadv_debug_knn_diag(unsigned long long *host_out, int reset) {
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(geoadv::kg_diag), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(geoadv::kg_diag), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

static int knn_mode_check(const char *op, int mode) {
    GA_REQUIRE(mode >= GEOADV_KNN_AUTO && mode <= GEOADV_KNN_GRID_SHELLS, "%s: kernel selection must be 0 = by size, 1 = all-points kernel only, "
               "2 = grid search at every size, 3 = as 2 with the wave-uniform shell search only (no lane-private first pass)", op);
    return GEOADV_OK;
}
extern "C" int geoadv_knn_grid_mode(int mode) {
    if (int rc = knn_mode_check("knn_grid_mode", mode)) return rc;
    g_knn_default_mode.store(mode);
    return GEOADV_OK;
}

static bool knn_list_kernels(int mode, int b, int m, int k) {   // the register-list kernels (else: the generic one-wave-per-query kernel, no scratch)
    return (mode == 0 ? k + 1 : k) <= 17 && (size_t)b * m < ((size_t)1 << 31);
}

// mode 0: knn_point (k values + indices); mode 1: the defender's distances (k includes the dropped self column)
// kmode: GEOADV_KNN_*; scratch: caller-owned, knn_fast_scratch_bytes (unused by the generic kernel)
static int launch_knn(int mode, int kmode, int b, int n, int m, int k, const float *xyz1, const float *xyz2, float *val, int *idx,
                      char *scratch, hipStream_t st) {
    if (int rc = row_lds_attr()) return rc;
    const int slots = mode == 0 ? k + 1 : k;              // register list of the fast kernel
    if (knn_list_kernels(mode, b, m, k)) {
        if (mode == 0) {
            if (slots <= 3) return launch_knn_fast<0, 3>(kmode, b, n, m, k, xyz1, xyz2, val, idx, scratch, st);
            if (slots <= 5) return launch_knn_fast<0, 5>(kmode, b, n, m, k, xyz1, xyz2, val, idx, scratch, st);
            if (slots <= 9) return launch_knn_fast<0, 9>(kmode, b, n, m, k, xyz1, xyz2, val, idx, scratch, st);
            if (slots <= 10) return launch_knn_fast<0, 10>(kmode, b, n, m, k, xyz1, xyz2, val, idx, scratch, st);     // k = 9: the defender's knn_point call
            if (slots <= 11) return launch_knn_fast<0, 11>(kmode, b, n, m, k, xyz1, xyz2, val, idx, scratch, st);
            if (slots <= 13) return launch_knn_fast<0, 13>(kmode, b, n, m, k, xyz1, xyz2, val, idx, scratch, st);
            return launch_knn_fast<0, 17>(kmode, b, n, m, k, xyz1, xyz2, val, idx, scratch, st);
        }
        if (slots <= 3) return launch_knn_fast<1, 3>(kmode, b, n, m, k, xyz1, xyz2, val, idx, scratch, st);
        if (slots <= 5) return launch_knn_fast<1, 5>(kmode, b, n, m, k, xyz1, xyz2, val, idx, scratch, st);
        if (slots <= 9) return launch_knn_fast<1, 9>(kmode, b, n, m, k, xyz1, xyz2, val, idx, scratch, st);
        if (slots <= 13) return launch_knn_fast<1, 13>(kmode, b, n, m, k, xyz1, xyz2, val, idx, scratch, st);
        return launch_knn_fast<1, 17>(kmode, b, n, m, k, xyz1, xyz2, val, idx, scratch, st);
    }
    // enough workgroups to fill the chip, a few queries each to amortise the launch
    int qper = 1;
    while ((long)cdiv(m, qper) * b > 16384 && qper < 16) qper *= 2;
    dim3 grid(cdiv(m, qper), b);
    if (mode == 0) knn_kernel<0><<<grid, 64, (size_t)n * 8, st>>>(n, m, k, qper, xyz1, xyz2, val, idx);
    else knn_kernel<1><<<grid, 64, (size_t)n * 8, st>>>(n, m, k, qper, xyz1, xyz2, val, idx);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

static int knn_point_check(int b, int n, int m, int k) {
    GA_REQUIRE(b >= 0 && n >= 1 && m >= 0, "knn_point: bad dimensions (b=%d n=%d m=%d)", b, n, m);
    GA_REQUIRE(k >= 1 && k <= n, "knn_point: k=%d must be in [1, n=%d]", k, n);
    GA_REQUIRE(n <= ROW_MAX_N, "knn_point: more than %d dataset points per cloud are not supported (n=%d)", ROW_MAX_N, n);
    GA_REQUIRE(b <= 65535, "knn_point: batch %d exceeds 65535", b);
    return GEOADV_OK;
}
static int knn_dists_check(int b, int n, int k) {
    GA_REQUIRE(b >= 0 && n >= 2, "knn_dists: bad dimensions (b=%d n=%d)", b, n);
    GA_REQUIRE(k >= 1 && k + 1 <= n, "knn_dists: k=%d must be in [1, n-1=%d]", k, n - 1);
    GA_REQUIRE(n <= ROW_MAX_N, "knn_dists: more than %d points per cloud are not supported (n=%d)", ROW_MAX_N, n);
    GA_REQUIRE(b <= 65535, "knn_dists: batch %d exceeds 65535", b);
    return GEOADV_OK;
}

extern "C" size_t geoadv_knn_workspace_bytes(int b, int n, int m, int k) {
    if (b <= 0 || n <= 0 || m <= 0 || k <= 0) return 256;
    return knn_fast_scratch_bytes(b, n, m) + 256;          // (+ 256: the workspace pointer is aligned up here)
}

static char *ws_align(void *p) { return reinterpret_cast<char *>((reinterpret_cast<size_t>(p) + 255) & ~(size_t)255); }

extern "C" int geoadv_knn_point_ws(int kernel, int b, int n, int m, int k, const float *xyz1, const float *xyz2, float *val, int *idx,
                                   void *workspace, size_t workspace_bytes, void *stream) {
    if (int rc = knn_mode_check("knn_point", kernel)) return rc;
    if (int rc = knn_point_check(b, n, m, k)) return rc;
    if (b == 0 || m == 0) return GEOADV_OK;
    GA_REQUIRE(xyz1 && xyz2 && val && idx, "knn_point: null pointer");
    GA_REQUIRE(workspace && workspace_bytes >= geoadv_knn_workspace_bytes(b, n, m, k), "knn_point: workspace too small (%zu bytes, need %zu)",
               workspace_bytes, geoadv_knn_workspace_bytes(b, n, m, k));
    return launch_knn(0, kernel, b, n, m, k, xyz1, xyz2, val, idx, ws_align(workspace), as_stream(stream));
}

extern "C" int geoadv_knn_dists_ws(int kernel, int b, int n, int k, const float *pc, float *out, void *workspace, size_t workspace_bytes,
                                   void *stream) {
    if (int rc = knn_mode_check("knn_dists", kernel)) return rc;
    if (int rc = knn_dists_check(b, n, k)) return rc;
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(pc && out, "knn_dists: null pointer");
    GA_REQUIRE(workspace && workspace_bytes >= geoadv_knn_workspace_bytes(b, n, n, k + 1), "knn_dists: workspace too small (%zu bytes, need %zu)",
               workspace_bytes, geoadv_knn_workspace_bytes(b, n, n, k + 1));
    return launch_knn(1, kernel, b, n, n, k + 1, pc, pc, out, nullptr, ws_align(workspace), as_stream(stream));
}

// The reference-shaped entry points (tf_grouping.py:48-75 has no scratch argument): thin conveniences over the _ws forms with
// stream-ordered scratch (hipMallocAsync / hipFreeAsync on the caller's stream) and the process-default kernel selection.
static int knn_with_own_scratch(int mode, int b, int n, int m, int k, const float *xyz1, const float *xyz2, float *val, int *idx, hipStream_t st) {
    char *scratch = nullptr;
    if (knn_list_kernels(mode, b, m, k)) GA_HIP(hipMallocAsync(reinterpret_cast<void **>(&scratch), knn_fast_scratch_bytes(b, n, m), st));
    const int rc = launch_knn(mode, g_knn_default_mode.load(), b, n, m, k, xyz1, xyz2, val, idx, scratch, st);
    if (scratch) GA_HIP(hipFreeAsync(scratch, st));
    return rc;
}

extern "C" int geoadv_knn_point(int b, int n, int m, int k, const float *xyz1, const float *xyz2, float *val, int *idx,
                                void *stream) {
    if (int rc = knn_point_check(b, n, m, k)) return rc;
    if (b == 0 || m == 0) return GEOADV_OK;
    GA_REQUIRE(xyz1 && xyz2 && val && idx, "knn_point: null pointer");
    return knn_with_own_scratch(0, b, n, m, k, xyz1, xyz2, val, idx, as_stream(stream));
}

extern "C" int geoadv_knn_dists(int b, int n, int k, const float *pc, float *out, void *stream) {
    if (int rc = knn_dists_check(b, n, k)) return rc;
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(pc && out, "knn_dists: null pointer");
    return knn_with_own_scratch(1, b, n, n, k + 1, pc, pc, out, nullptr, as_stream(stream));
}

extern "C" int geoadv_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1,
                                       const float *xyz2, int *idx, int *pts_cnt, void *stream) {
    GA_REQUIRE(b >= 0 && n >= 0 && m >= 0, "query_ball_point: negative dimension");
    GA_REQUIRE(radius > 0.f, "QueryBallPoint expects positive radius");           // tf_grouping.cpp:70-71
    GA_REQUIRE(nsample > 0, "QueryBallPoint expects positive nsample");           // tf_grouping.cpp:73-74
    GA_REQUIRE(b <= 65535, "query_ball_point: batch %d exceeds 65535", b);
    if (b == 0 || m == 0) return GEOADV_OK;
    GA_REQUIRE(xyz1 && xyz2 && idx, "query_ball_point: null pointer");
    // largest squared distance that still passes `max(sqrtf(d2), 1e-20f) < radius` (none if radius <= 1e-20f; every finite one
    // if radius is infinite)
    float t2max = -1.f;
    if (radius > 1e-20f) {
        t2max = radius * radius;
        if (!(t2max < INFINITY)) t2max = FLT_MAX;
        while (t2max > 0.f && !(sqrtf(t2max) < radius)) t2max = nextafterf(t2max, 0.f);
        while (t2max < FLT_MAX && sqrtf(nextafterf(t2max, INFINITY)) < radius) t2max = nextafterf(t2max, INFINITY);
        if (!(sqrtf(t2max) < radius)) t2max = -1.f;         // (radius so small that not even d2 = 0 ... cannot happen above 1e-20)
    }
    query_ball_fast_kernel<<<dim3(cdiv(m, QB_THREADS), b), QB_THREADS, 0, as_stream(stream)>>>(n, m, t2max, nsample, xyz1, xyz2, idx, pts_cnt);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

extern "C" int geoadv_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx,
                                  float *out, void *stream) {
    GA_REQUIRE(b >= 0 && n >= 0 && c >= 0 && m >= 0 && nsample >= 0, "group_point: negative dimension");
    const size_t per_cloud = (size_t)m * nsample, total = (size_t)b * per_cloud * c;
    if (total == 0) return GEOADV_OK;
    GA_REQUIRE(points && idx && out, "group_point: null pointer");
    group_point_kernel<<<(unsigned)((total + 255) / 256), 256, 0, as_stream(stream)>>>(n, c, per_cloud, total, points, idx, out);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

static void gpg_shape(int n, int m, int nsample, int &entries, int &segs, int &seg_len, int &passes) {
    entries = m * nsample;
    // segments of >= 16 groups of 64 entries, at most 64 of them per cloud
    segs = std::max(1, std::min(64, entries / (64 * 16)));
    seg_len = cdiv(cdiv(entries, segs), 64) * 64;
    passes = 1;
    while ((n >> (6 * passes)) != 0) ++passes;              // destinations 0 .. n (n = invalid)
}
// two (keys, perm) buffers, the (digit, segment) table, start[b][n + 1]
static size_t gpg_scratch_ints(int b, int n, int m, int nsample) {
    int entries, segs, seg_len, passes;
    gpg_shape(n, m, nsample, entries, segs, seg_len, passes);
    return 4 * (size_t)b * entries + (size_t)b * (64 * segs + 1) + (size_t)b * (n + 1);
}
static int gpg_check(int b, int n, int c, int m, int nsample) {
    GA_REQUIRE(b >= 0 && n >= 0 && c >= 0 && m >= 0 && nsample >= 0, "group_point_grad: negative dimension");
    GA_REQUIRE(b <= 65535, "group_point_grad: batch %d exceeds 65535", b);
    GA_REQUIRE((size_t)m * nsample < ((size_t)1 << 30), "group_point_grad: m * nsample too large");
    return GEOADV_OK;
}

static int launch_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx, float *grad_points,
                                   int *scratch, hipStream_t st) {
    int entries, segs, seg_len, passes;
    gpg_shape(n, m, nsample, entries, segs, seg_len, passes);
    const size_t per = (size_t)b * entries, table_ints = (size_t)b * (64 * segs + 1);
    int *keys[2] = {scratch, scratch + per}, *perm[2] = {scratch + 2 * per, scratch + 3 * per};
    int *table = scratch + 4 * per, *start = table + table_ints;
    const dim3 grid(cdiv(segs, GPG_WAVES), b);
    const int *kin = idx, *pin = nullptr;
    for (int pass = 0; pass < passes; ++pass) {
        int *kout = keys[pass & 1], *pout = perm[pass & 1];
        gpg_pass_kernel<false><<<grid, 64 * GPG_WAVES, 0, st>>>(n, entries, 6 * pass, segs, seg_len, kin, pin, table, nullptr, nullptr);
        gpg_scan_kernel<<<b, 1024, 0, st>>>(64 * segs, table);
        gpg_pass_kernel<true><<<grid, 64 * GPG_WAVES, 0, st>>>(n, entries, 6 * pass, segs, seg_len, kin, pin, table, kout, pout);
        kin = kout; pin = pout;
    }
    gpg_bounds_kernel<<<dim3(cdiv(n + 1, 256), b), 256, 0, st>>>(n, entries, kin, start);
    gpg_sum_kernel<<<dim3(cdiv(n * c, 256), b), 256, 0, st>>>(n, c, entries, grad_out, start, pin, grad_points);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

extern "C" size_t geoadv_group_point_grad_workspace_bytes(int b, int n, int c, int m, int nsample) {
    (void)c;
    if (b <= 0 || n < 0 || m <= 0 || nsample <= 0) return 256;
    return sizeof(int) * gpg_scratch_ints(b, n, m, nsample) + 256;
}

extern "C" int geoadv_group_point_grad_ws(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                                          float *grad_points, void *workspace, size_t workspace_bytes, void *stream) {
    if (int rc = gpg_check(b, n, c, m, nsample)) return rc;
    if ((size_t)b * n * c == 0) return GEOADV_OK;
    GA_REQUIRE(grad_points && (m * nsample == 0 || (grad_out && idx)), "group_point_grad: null pointer");
    hipStream_t st = as_stream(stream);
    if (m * nsample == 0) {
        GA_HIP(hipMemsetAsync(grad_points, 0, (size_t)b * n * c * sizeof(float), st));        // tf_grouping.cpp:204
        return GEOADV_OK;
    }
    GA_REQUIRE(workspace && workspace_bytes >= geoadv_group_point_grad_workspace_bytes(b, n, c, m, nsample),
               "group_point_grad: workspace too small (%zu bytes, need %zu)", workspace_bytes, geoadv_group_point_grad_workspace_bytes(b, n, c, m, nsample));
    return launch_group_point_grad(b, n, c, m, nsample, grad_out, idx, grad_points, reinterpret_cast<int *>(ws_align(workspace)), st);
}

// reference-shaped (groupPointGradLauncher has no scratch argument): stream-ordered scratch of its own
extern "C" int geoadv_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                                       float *grad_points, void *stream) {
    if (int rc = gpg_check(b, n, c, m, nsample)) return rc;
    if ((size_t)b * n * c == 0) return GEOADV_OK;
    GA_REQUIRE(grad_points && (m * nsample == 0 || (grad_out && idx)), "group_point_grad: null pointer");
    hipStream_t st = as_stream(stream);
    if (m * nsample == 0) {
        GA_HIP(hipMemsetAsync(grad_points, 0, (size_t)b * n * c * sizeof(float), st));        // tf_grouping.cpp:204
        return GEOADV_OK;
    }
    int *scratch = nullptr;
    GA_HIP(hipMallocAsync(reinterpret_cast<void **>(&scratch), sizeof(int) * gpg_scratch_ints(b, n, m, nsample), st));
    const int rc = launch_group_point_grad(b, n, c, m, nsample, grad_out, idx, grad_points, scratch, st);
    GA_HIP(hipFreeAsync(scratch, st));
    return rc;
}
GA_STAMPS_GETTER(geoadv_debug_stamps_grouping)
