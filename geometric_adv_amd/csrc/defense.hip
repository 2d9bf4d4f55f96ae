// The two defenses' per-cloud bookkeeping on the device (round 4; until then per-cloud numpy loops on the host, as in the
// reference): src/adversary_utils.py:149-178 (get_outlier_pc_inlier_pc: threshold the per-point kNN score, pack inliers and
// outliers stably, pad with the last packed point, int16 indices) and src/ae_utils.py:12-80 (get_critical_points /
// get_critical_pc_non_critical_pc: the points that attain the encoder's max-pool, most channels first, and their complement).
// One 1024-thread workgroup per cloud; packing = wave ballots + a 16-entry LDS scan per 1024-point chunk, in point order.
// Integer / copy work: results are the reference's bit for bit, with ONE documented exception -- the order of critical points
// that own EQUALLY many channels, which the reference leaves to numpy's default (unstable, build-dependent) argsort
// (ae_utils.py:34: np.argsort(counts)[::-1]); here it is the order a STABLE sort gives that expression: count descending,
// then point index descending.  Nothing downstream depends on it (the padded critical cloud has the same latent code,
// the complement is in point order).
#include "common.h"

namespace geoadv {

constexpr int DF_THREADS = 1024;
constexpr int DF_WAVES = DF_THREADS / 64;

struct CompactLds {
    int wcnt[DF_WAVES];
    int last;               // index of the last kept point so far (-1: none)
};

// Stable packing of the points p in [0, n) with keep(p): emit(p, slot) for slot = 0, 1, ... in point order.  Returns the
// number kept; L.last = the last kept point.  Every thread of the workgroup must call it.
template <class Keep, class Emit>
__device__ __forceinline__ int block_compact(int n, CompactLds &L, Keep keep, Emit emit) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) L.last = -1;
    int base = 0;
    for (int p0 = 0; p0 < n; p0 += DF_THREADS) {
        const int p = p0 + threadIdx.x;
        const bool f = p < n && keep(p);
        const unsigned long long bal = __ballot(f);
        if (lane == 0) L.wcnt[wave] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < DF_WAVES; ++w) {
            const int cw = L.wcnt[w];
            before += w < wave ? cw : 0;
            total += cw;
        }
        if (f) {
            const int r = before + __popcll(bal & ((1ull << lane) - 1ull));
            emit(p, base + r);
            if (r == total - 1) L.last = p;
        }
        base += total;
        __syncthreads();
    }
    return base;
}

__device__ __forceinline__ void copy_point(float *dst, const float *src) { dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; }

// slots [count, n) of a packed cloud: the last packed point again (adversary_utils.py:168-169,175-176; ae_utils.py:71,77),
// zeros when nothing was packed (the reference's arrays start as zeros)
__device__ __forceinline__ void pad_cloud(float *dst, const float *cloud, int count, int n, int last) {
    float px = 0.f, py = 0.f, pz = 0.f;
    if (count > 0) { px = cloud[3 * last]; py = cloud[3 * last + 1]; pz = cloud[3 * last + 2]; }
    for (int s = count + threadIdx.x; s < n; s += DF_THREADS) { dst[3 * s] = px; dst[3 * s + 1] = py; dst[3 * s + 2] = pz; }
}

// score[p] = np.mean(knn[p, :top_k]) in float32: a left-to-right sum (numpy's reduction of fewer than 8 elements) divided
// by the count (run_defense_surface.py:187-191 thresholds the mean of the first two distances)
__device__ __forceinline__ float knn_score(const float *row, int top_k) {
    float s = row[0];
    for (int j = 1; j < top_k; ++j) s += row[j];
    return top_k > 1 ? s / (float)top_k : s;
}

__global__ __launch_bounds__(DF_THREADS) void outlier_filter_kernel(int n, const float *pc, const float *knn, int stride, int top_k,
                                                                    float thresh, float *outlier_pc, short *outlier_idx,
                                                                    short *outlier_num, float *inlier_pc) {
    __shared__ CompactLds L;
    const int c = blockIdx.x;
    const float *cloud = pc + (size_t)c * n * 3;
    const float *rows = knn + (size_t)c * n * stride;
    float *opc = outlier_pc ? outlier_pc + (size_t)c * n * 3 : nullptr;
    short *oix = outlier_idx ? outlier_idx + (size_t)c * n : nullptr;
    float *ipc = inlier_pc + (size_t)c * n * 3;
    // outliers: np.where(d > thresh) (a NaN score is neither an outlier nor an inlier, as in numpy)
    const int no = block_compact(n, L, [&](int p) { return knn_score(rows + (size_t)p * stride, top_k) > thresh; },
                                 [&](int p, int slot) {
                                     if (opc) copy_point(opc + 3 * (size_t)slot, cloud + 3 * (size_t)p);
                                     if (oix) oix[slot] = (short)p;                          // int16, wraps like the numpy store
                                 });
    const int last_o = L.last;
    __syncthreads();
    if (opc) pad_cloud(opc, cloud, no, n, last_o);
    if (oix) for (int s = no + threadIdx.x; s < n; s += DF_THREADS) oix[s] = 0;
    if (outlier_num && threadIdx.x == 0) outlier_num[c] = (short)no;
    const int ni = block_compact(n, L, [&](int p) { return knn_score(rows + (size_t)p * stride, top_k) <= thresh; },
                                 [&](int p, int slot) { copy_point(ipc + 3 * (size_t)slot, cloud + 3 * (size_t)p); });
    const int last_i = L.last;
    __syncthreads();
    pad_cloud(ipc, cloud, ni, n, last_i);
}

constexpr int CR_MAX_C = 1024;            // latent channels
constexpr int CR_MAX_N = 32768;           // points per cloud (the critical-point bitmap lives in LDS)

__global__ __launch_bounds__(DF_THREADS) void critical_split_kernel(int n, int C, const float *pc, const float *max_val, const int *max_idx,
                                                                    float *critical_points, short *critical_idx, short *critical_num,
                                                                    float *critical_pc, float *non_critical_pc) {
    __shared__ CompactLds L;
    __shared__ int cidx[CR_MAX_C], ccount[CR_MAX_C], sidx[CR_MAX_C];
    __shared__ unsigned bitmap[CR_MAX_N / 32];
    __shared__ int num_s;
    const int c = blockIdx.x, t = threadIdx.x;
    const float *cloud = pc + (size_t)c * n * 3;
    for (int w = t; w < (n + 31) / 32; w += DF_THREADS) bitmap[w] = 0u;
    if (t == 0) num_s = 0;
    int idx = -1;
    if (t < C) {
        const int i = max_idx[(size_t)c * C + t];
        if (max_val[(size_t)c * C + t] > 0.0f && i >= 0 && i < n) idx = i;      // ae_utils.py:25: channels that are 0 for the whole cloud drop out
        cidx[t] = idx;
    }
    __syncthreads();
    int cnt = 0;
    if (idx >= 0) {                                                             // np.unique(..., return_counts=True)
        bool first = true;
        for (int u = 0; u < C; ++u) {
            const bool same = cidx[u] == idx;
            cnt += same ? 1 : 0;
            first = first && !(same && u < t);
        }
        cnt = first ? cnt : 0;
        if (first) { atomicOr(&bitmap[idx >> 5], 1u << (idx & 31)); atomicAdd(&num_s, 1); }
    }
    if (t < C) ccount[t] = cnt;
    __syncthreads();
    const int num = num_s;
    if (cnt > 0) {                                                              // np.argsort(counts)[::-1] as a stable sort orders it
        int rank = 0;
        for (int u = 0; u < C; ++u) {
            const int cu = ccount[u];
            rank += (cu > cnt || (cu == cnt && cidx[u] > idx)) ? 1 : 0;
        }
        sidx[rank] = idx;
    }
    __syncthreads();
    if (t < C) {
        const bool in = t < num;
        if (critical_idx) critical_idx[(size_t)c * C + t] = in ? (short)sidx[t] : (short)0;
        if (critical_points) {
            float *d = critical_points + ((size_t)c * C + t) * 3;
            if (in) copy_point(d, cloud + 3 * (size_t)sidx[t]); else { d[0] = 0.f; d[1] = 0.f; d[2] = 0.f; }
        }
    }
    if (critical_num && t == 0) critical_num[c] = (short)num;
    if (critical_pc) {
        float *d = critical_pc + (size_t)c * n * 3;
        for (int s = t; s < n; s += DF_THREADS) {
            if (num > 0) copy_point(d + 3 * (size_t)s, cloud + 3 * (size_t)sidx[s < num ? s : num - 1]);
            else { d[3 * s] = 0.f; d[3 * s + 1] = 0.f; d[3 * s + 2] = 0.f; }
        }
    }
    if (non_critical_pc) {
        float *d = non_critical_pc + (size_t)c * n * 3;
        const int nn = block_compact(n, L, [&](int p) { return ((bitmap[p >> 5] >> (p & 31)) & 1u) == 0u; },
                                     [&](int p, int slot) { copy_point(d + 3 * (size_t)slot, cloud + 3 * (size_t)p); });
        const int last = L.last;
        __syncthreads();
        pad_cloud(d, cloud, nn, n, last);
    }
}

}  // namespace geoadv

using namespace geoadv;

extern "C" int geoadv_outlier_filter(int b, int n, const float *pc, const float *knn_dists, int knn_stride, int top_k, float thresh,
                                     float *outlier_pc, short *outlier_idx, short *outlier_num, float *inlier_pc, void *stream) {
    GA_REQUIRE(b >= 0 && n >= 1, "outlier_filter: bad dimensions (b=%d n=%d)", b, n);
    GA_REQUIRE(top_k >= 1 && top_k <= knn_stride && top_k < 8,
               "outlier_filter: top_k=%d must be in [1, min(knn_stride=%d, 7)] (the mean is numpy's left-to-right float32 sum, which "
               "numpy itself only uses below 8 elements)", top_k, knn_stride);
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(pc && knn_dists && inlier_pc, "outlier_filter: null pointer");
    outlier_filter_kernel<<<b, DF_THREADS, 0, as_stream(stream)>>>(n, pc, knn_dists, knn_stride, top_k, thresh, outlier_pc, outlier_idx,
                                                                    outlier_num, inlier_pc);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

extern "C" int geoadv_critical_split(int b, int n, int c, const float *pc, const float *max_val, const int *max_idx,
                                     float *critical_points, short *critical_idx, short *critical_num, float *critical_pc,
                                     float *non_critical_pc, void *stream) {
    GA_REQUIRE(b >= 0 && n >= 1 && c >= 1, "critical_split: bad dimensions (b=%d n=%d c=%d)", b, n, c);
    GA_REQUIRE(c <= CR_MAX_C, "critical_split: more than %d latent channels are not supported (c=%d)", CR_MAX_C, c);
    GA_REQUIRE(n <= CR_MAX_N, "critical_split: more than %d points per cloud are not supported (n=%d)", CR_MAX_N, n);
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(pc && max_val && max_idx, "critical_split: null pointer");
    critical_split_kernel<<<b, DF_THREADS, 0, as_stream(stream)>>>(n, c, pc, max_val, max_idx, critical_points, critical_idx, critical_num,
                                                                    critical_pc, non_critical_pc);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}
