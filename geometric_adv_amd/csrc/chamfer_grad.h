// Deterministic, CPU-ordered Chamfer gradient for one output side (shared by the NnDistanceGrad
// op in chamfer.hip and by the attack loop in attack.hip).
//
// For the output side "own" every point j receives its own term g_j*(p_j - q_match[j]) and one
// scatter term per point k of the other side with match_oth[k] == j.  The reference CPU op applies
// them in a fixed order (external/structural_losses/tf_nndistance.cpp:130-163): for grad_xyz1 the
// own term first, then the scatter terms in ascending k; for grad_xyz2 the scatter terms
// (ascending j) first, then the own term.  We reproduce that order -- and therefore the bits --
// with three phases per (cloud, side) workgroup:
//   1. bitonic sort of the keys (match_oth[k] << 16 | k) in LDS           (all threads, P/2 pairs)
//   2. the scatter terms g_k*(q_k - p_j) of all sorted positions in parallel -> LDS planes
//   3. one thread per own point: binary search of its segment, then a sequential (CPU-ordered)
//      sum over LDS values -- no global loads on the serial chain, so a point matched by a
//      thousand others costs a few microseconds instead of a hundred.
// Beyond TERMS_MAX_P keys the term planes no longer fit in LDS next to the keys; phase 2 is then
// skipped and phase 3 computes the terms on the fly.
#pragma once
#include "common.h"

namespace geoadv {

constexpr int CG_TERMS_MAX_P = 8192;   // keys 32 KB + 3 planes 96 KB = 128 KB of the 160 KB LDS

struct GradSide {
    int n_own, n_oth;
    const float *own, *oth;              // [n_own][3], [n_oth][3]
    const int *match_own, *match_oth;    // own -> other matches, other -> own matches
    const float *gd_own, *gd_oth;        // per-point upstream gradients, or null: use the scalars
    float gd_own_s, gd_oth_s;
    int jstar;                           // own point whose upstream gradient gets `extra` added (-1: none)
    float extra;
    float *gout;                         // [n_own][3]
};

template <int THREADS>
__device__ __forceinline__ void bitonic_sort_lds(unsigned *keys, int P) {
    const int half = P >> 1;
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < half; t += THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // bit j clear
                const int p = i | j;
                const unsigned a = keys[i], b = keys[p];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { keys[i] = b; keys[p] = a; }
            }
            __syncthreads();
        }
}

template <bool OWN_FIRST, int THREADS>
__device__ void chamfer_grad_side(const GradSide &s, unsigned *lds, int P) {
#pragma clang fp contract(off)
    unsigned *keys = lds;
    const bool planes = P <= CG_TERMS_MAX_P;
    float *tx = reinterpret_cast<float *>(lds + P), *ty = tx + P, *tz = ty + P;
    for (int i = threadIdx.x; i < P; i += THREADS)
        keys[i] = i < s.n_oth ? (((unsigned)s.match_oth[i] << 16) | (unsigned)i) : 0xFFFFFFFFu;
    __syncthreads();
    bitonic_sort_lds<THREADS>(keys, P);
    if (planes) {
        for (int pos = threadIdx.x; pos < s.n_oth; pos += THREADS) {
            const unsigned key = keys[pos];
            const int k = key & 0xFFFF, j = key >> 16;
            const float gk = (s.gd_oth ? s.gd_oth[k] : s.gd_oth_s) * 2;
            tx[pos] = gk * (s.oth[3 * k] - s.own[3 * j]);
            ty[pos] = gk * (s.oth[3 * k + 1] - s.own[3 * j + 1]);
            tz[pos] = gk * (s.oth[3 * k + 2] - s.own[3 * j + 2]);
        }
        __syncthreads();
    }
    for (int j = threadIdx.x; j < s.n_own; j += THREADS) {
        const float px = s.own[3 * j], py = s.own[3 * j + 1], pz = s.own[3 * j + 2];
        const int mj = s.match_own[j];
        float gd = s.gd_own ? s.gd_own[j] : s.gd_own_s;
        if (j == s.jstar) gd = gd + s.extra;
        const float g = gd * 2;
        const float ox = g * (px - s.oth[3 * mj]), oy = g * (py - s.oth[3 * mj + 1]), oz = g * (pz - s.oth[3 * mj + 2]);
        float ax = 0.f, ay = 0.f, az = 0.f;
        if (OWN_FIRST) { ax += ox; ay += oy; az += oz; }
        // segment [lo, hi) of sorted positions whose high half equals j
        const unsigned want = (unsigned)j << 16;
        int lo = 0, hi = P;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (keys[mid] < want) lo = mid + 1; else hi = mid;
        }
        int e = lo;
        hi = P;
        const unsigned wend = want | 0xFFFFu;
        while (e < hi) {
            const int mid = (e + hi) >> 1;
            if (keys[mid] <= wend) e = mid + 1; else hi = mid;
        }
        if (planes) {
            int p = lo;
            for (; p + 4 <= e; p += 4) {        // loads first, then the ordered chain
                const float x0 = tx[p], x1 = tx[p + 1], x2 = tx[p + 2], x3 = tx[p + 3];
                const float y0 = ty[p], y1 = ty[p + 1], y2 = ty[p + 2], y3 = ty[p + 3];
                const float z0 = tz[p], z1 = tz[p + 1], z2 = tz[p + 2], z3 = tz[p + 3];
                ax -= x0; ay -= y0; az -= z0;
                ax -= x1; ay -= y1; az -= z1;
                ax -= x2; ay -= y2; az -= z2;
                ax -= x3; ay -= y3; az -= z3;
            }
            for (; p < e; ++p) { ax -= tx[p]; ay -= ty[p]; az -= tz[p]; }
        } else {
            for (int p = lo; p < e; ++p) {
                const int k = keys[p] & 0xFFFF;
                const float gk = (s.gd_oth ? s.gd_oth[k] : s.gd_oth_s) * 2;
                ax -= gk * (s.oth[3 * k] - px);
                ay -= gk * (s.oth[3 * k + 1] - py);
                az -= gk * (s.oth[3 * k + 2] - pz);
            }
        }
        if (!OWN_FIRST) { ax += ox; ay += oy; az += oz; }
        s.gout[3 * j] = ax; s.gout[3 * j + 1] = ay; s.gout[3 * j + 2] = az;
    }
}

}  // namespace geoadv
