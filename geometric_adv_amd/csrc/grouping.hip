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
        for (int t = s + lane; t < n; t += 64) {
            const float v = val[t];
            if (v < bv) { bv = v; bp = t; }     // per lane ascending t: first min kept
        }
        // a lane whose values are all NaN / that saw nothing keeps (inf, maxint)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int op = __shfl_xor(bp, off);
            if (ov < bv || (ov == bv && op < bp)) { bv = ov; bp = op; }
        }
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
    for (int t = threadIdx.x; t < n; t += 64) {
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
constexpr int KF_QCAP = 16;                // queue slots per lane
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
            if (MODE == 1) {                               // values only: a compare-exchange chain on the BIT PATTERNS -- distances are
                unsigned x = __float_as_uint(d);           // sums of squares (>= +0, or +inf), for which unsigned order = float order; no
#pragma unroll                                             // selects, no moves, no canonicalising v_max (a NaN only ever marks the query `odd`)
                for (int i = 0; i < S; ++i) {
                    const unsigned vi = __float_as_uint(v[i]);
                    const unsigned lo = min(vi, x);
                    x = max(vi, x);
                    v[i] = __uint_as_float(lo);
                }
            } else {                                       // (value, index): the candidate sinks past every entry it is strictly below;
                float x = d;                               // an equal entry -- earlier index -- stays ahead of it.  Selects only, no branch.
                int xi = id;
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    const bool c = x < v[i];
                    const float lo = c ? x : v[i], hi = c ? v[i] : x;
                    const int loi = c ? xi : ix[i], hii = c ? ix[i] : xi;
                    v[i] = lo; ix[i] = loi; x = hi; xi = hii;
                }
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

template <int MODE, int S>
static int launch_knn_fast(int b, int n, int m, int k, const float *xyz1, const float *xyz2, float *val, int *idx, hipStream_t st) {
    int *redo = nullptr;                                  // stream-ordered scratch: counter + one entry per query at most
    GA_HIP(hipMallocAsync(reinterpret_cast<void **>(&redo), sizeof(int) * ((size_t)b * m + 1), st));
    GA_HIP(hipMemsetAsync(redo, 0, sizeof(int), st));
    knn_fast_kernel<MODE, S><<<dim3(cdiv(m, KF_THREADS), b), KF_THREADS, 0, st>>>(n, m, k, xyz1, xyz2, val, idx, redo);
    knn_redo_kernel<MODE><<<1024, 64, (size_t)n * 8, st>>>(n, m, k, xyz1, xyz2, val, idx, redo);
    const hipError_t launched = hipGetLastError();
    GA_HIP(hipFreeAsync(redo, st));
    GA_HIP(launched);
    return GEOADV_OK;
}

// mode 0: knn_point (k values + indices); mode 1: the defender's distances (k includes the dropped self column)
static int launch_knn(int mode, int b, int n, int m, int k, const float *xyz1, const float *xyz2, float *val, int *idx,
                      hipStream_t st) {
    if (int rc = row_lds_attr()) return rc;
    const int slots = mode == 0 ? k + 1 : k;              // register list of the fast kernel
    if (slots <= 17 && (size_t)b * m < ((size_t)1 << 31)) {
        if (mode == 0) {
            if (slots <= 3) return launch_knn_fast<0, 3>(b, n, m, k, xyz1, xyz2, val, idx, st);
            if (slots <= 5) return launch_knn_fast<0, 5>(b, n, m, k, xyz1, xyz2, val, idx, st);
            if (slots <= 9) return launch_knn_fast<0, 9>(b, n, m, k, xyz1, xyz2, val, idx, st);
            return launch_knn_fast<0, 17>(b, n, m, k, xyz1, xyz2, val, idx, st);
        }
        if (slots <= 3) return launch_knn_fast<1, 3>(b, n, m, k, xyz1, xyz2, val, idx, st);
        if (slots <= 5) return launch_knn_fast<1, 5>(b, n, m, k, xyz1, xyz2, val, idx, st);
        if (slots <= 9) return launch_knn_fast<1, 9>(b, n, m, k, xyz1, xyz2, val, idx, st);
        return launch_knn_fast<1, 17>(b, n, m, k, xyz1, xyz2, val, idx, st);
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

extern "C" int geoadv_knn_point(int b, int n, int m, int k, const float *xyz1, const float *xyz2, float *val, int *idx,
                                void *stream) {
    GA_REQUIRE(b >= 0 && n >= 1 && m >= 0, "knn_point: bad dimensions (b=%d n=%d m=%d)", b, n, m);
    GA_REQUIRE(k >= 1 && k <= n, "knn_point: k=%d must be in [1, n=%d]", k, n);
    GA_REQUIRE(n <= ROW_MAX_N, "knn_point: more than %d dataset points per cloud are not supported (n=%d)", ROW_MAX_N, n);
    GA_REQUIRE(b <= 65535, "knn_point: batch %d exceeds 65535", b);
    if (b == 0 || m == 0) return GEOADV_OK;
    GA_REQUIRE(xyz1 && xyz2 && val && idx, "knn_point: null pointer");
    return launch_knn(0, b, n, m, k, xyz1, xyz2, val, idx, as_stream(stream));
}

extern "C" int geoadv_knn_dists(int b, int n, int k, const float *pc, float *out, void *stream) {
    GA_REQUIRE(b >= 0 && n >= 2, "knn_dists: bad dimensions (b=%d n=%d)", b, n);
    GA_REQUIRE(k >= 1 && k + 1 <= n, "knn_dists: k=%d must be in [1, n-1=%d]", k, n - 1);
    GA_REQUIRE(n <= ROW_MAX_N, "knn_dists: more than %d points per cloud are not supported (n=%d)", ROW_MAX_N, n);
    GA_REQUIRE(b <= 65535, "knn_dists: batch %d exceeds 65535", b);
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(pc && out, "knn_dists: null pointer");
    return launch_knn(1, b, n, n, k + 1, pc, pc, out, nullptr, as_stream(stream));
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

extern "C" int geoadv_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                                       float *grad_points, void *stream) {
    GA_REQUIRE(b >= 0 && n >= 0 && c >= 0 && m >= 0 && nsample >= 0, "group_point_grad: negative dimension");
    GA_REQUIRE(b <= 65535, "group_point_grad: batch %d exceeds 65535", b);
    if ((size_t)b * n * c == 0) return GEOADV_OK;
    GA_REQUIRE(grad_points && (m * nsample == 0 || (grad_out && idx)), "group_point_grad: null pointer");
    GA_REQUIRE((size_t)m * nsample < ((size_t)1 << 30), "group_point_grad: m * nsample too large");
    hipStream_t st = as_stream(stream);
    const int entries = m * nsample;
    if (entries == 0) {
        GA_HIP(hipMemsetAsync(grad_points, 0, (size_t)b * n * c * sizeof(float), st));        // tf_grouping.cpp:204
        return GEOADV_OK;
    }
    // segments of >= 16 groups of 64 entries, at most 64 of them per cloud
    const int segs = std::max(1, std::min(64, entries / (64 * 16)));
    const int seg_len = cdiv(cdiv(entries, segs), 64) * 64;
    int passes = 1;
    while ((n >> (6 * passes)) != 0) ++passes;              // destinations 0 .. n (n = invalid)
    // stream-ordered scratch: two (keys, perm) buffers, the (digit, segment) table, start[b][n + 1]
    const size_t per = (size_t)b * entries, table_ints = (size_t)b * (64 * segs + 1), start_ints = (size_t)b * (n + 1);
    int *scratch = nullptr;
    GA_HIP(hipMallocAsync(reinterpret_cast<void **>(&scratch), (4 * per + table_ints + start_ints) * sizeof(int), st));
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
    const hipError_t err = hipGetLastError();
    GA_HIP(hipFreeAsync(scratch, st));
    GA_HIP(err);
    return GEOADV_OK;
}
