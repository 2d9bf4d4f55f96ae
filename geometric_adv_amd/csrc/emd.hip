// approx-EMD (ApproxMatch / MatchCost / MatchCostGrad) for gfx950.
//
// Replaces approxmatchLauncher / matchcostLauncher / matchcostgradLauncher
// (external/structural_losses/tf_approxmatch.cpp:141-143; kernels tf_approxmatch_g.cu).
// Parity target: the reference CPU op (tf_approxmatch.cpp:23-140): 11 levels j = 8..-2 with
// level = -4^j (0 at j = -2), double bookkeeping, expf of the float-narrowed exponent.
//
// The reference GPU kernel read-modify-writes the (b,m,n) match matrix once per level -- 10 x 2 x 4
// bytes per pair, the dominant HBM traffic.  Here the per-level weights are kept in
// FACTORISED form: with w_j(k,l) = expf(level_j * |p_k - q_l|^2),
//     weight_j(k,l) = w_j(k,l) * fL_j[k] * fR_j[l],
//     fL_j[k] = remL[k] / (1e-9 + sum_l w_j remR[l]),                       (pass A, thread per k)
//     T_l = sum_k w_j fL_j[k];  r = min(remR[l] / (1e-9 + remR[l] T_l), 1);
//     fR_j[l] = remR[l] r;  remR[l] <- max(remR[l] - fR_j[l] T_l, 0)        (pass B, thread per l)
//     remL[k] <- max(remL[k] - fL_j[k] sum_l w_j fR_j[l], 0)                (pass C, thread per k; fused with pass A of
//                                                                            level j+1: one distance, two weights)
// which is the CPU loop (:36-78) with the row/column normalisations pulled out of the pair sums.
// Only 12 (n+m) doubles per cloud live in HBM during the levels; match is written ONCE at the end
// (sum over the 11 levels, accumulated level by level in float like the CPU's `match[k] += weight[k]`).
// All sweeps are exp/VALU bound; every per-point sum runs sequentially in the CPU's order.
#include "common.h"
#include <math.h>

#pragma clang fp contract(off)

namespace geoadv {

constexpr int EMD_LEVELS = 11;                  // j = 8 .. -2 (tf_approxmatch.cpp:31)
constexpr int EMD_TILE = 1024;                  // "other" points staged per LDS tile

static inline double emd_level(int li) {        // li = 0..10  <->  j = 8..-2
    const int j = 8 - li;
    return j == -2 ? 0.0 : -(double)powf(4.0f, (float)j);     // level = -powf(4.0, j) (:33-35)
}

// temp layout per cloud (doubles): remL[n] remR[m] then per level: fL[n] fR[m]
__host__ __device__ inline size_t emd_temp_doubles_per_cloud(int n, int m) { return (size_t)(n + m) * (1 + EMD_LEVELS); }

__device__ __forceinline__ double pair_w(double level, double ox, double oy, double oz, double x2, double y2, double z2) {
    const double d2 = (ox - x2) * (ox - x2) + (oy - y2) * (oy - y2) + (oz - z2) * (oz - z2);
    return (double)expf((float)(level * d2));
}

__global__ void emd_init_kernel(int n, int m, double *temp) {
    const int c = blockIdx.y;
    double *t = temp + (size_t)c * emd_temp_doubles_per_cloud(n, m);
    const int big = n > m ? n : m;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) t[i] = (double)(big / n);                     // factorl = max(n,m)/n, integer division (:25)
    else if (i < n + m) t[i] = (double)(big / m);            // factorr (:26)
}

// PASS 0 = A, 1 = B, 2 = C.  Thread per "own" point; the "other" cloud and its per-point factor are
// staged through LDS and walked in ascending order.
template <int PASS>
__global__ __launch_bounds__(256) void emd_sweep_kernel(int n, int m, int li, double level, const float *xyz1,
                                                        const float *xyz2, double *temp) {
    // the other cloud's coordinates are widened to double ONCE per tile here (the CPU widens them per pair, :38-41: same
    // values), not once per pair in the loop: conversions issue at the same rate as the fp64 arithmetic they feed
    __shared__ double ox[EMD_TILE], oy[EMD_TILE], oz[EMD_TILE];
    __shared__ double of[EMD_TILE];
    const int c = blockIdx.y;
    double *t = temp + (size_t)c * emd_temp_doubles_per_cloud(n, m);
    double *remL = t, *remR = t + n, *fL = t + (size_t)(n + m) * (1 + li), *fR = fL + n;
    const bool own_is_1 = PASS != 1;
    const int n_own = own_is_1 ? n : m, n_oth = own_is_1 ? m : n;
    const float *own = (own_is_1 ? xyz1 : xyz2) + (size_t)c * n_own * 3;
    const float *oth = (own_is_1 ? xyz2 : xyz1) + (size_t)c * n_oth * 3;
    const double *ofac = PASS == 0 ? remR : (PASS == 1 ? fL : fR);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n_own;
    double px = 0, py = 0, pz = 0;
    if (live) { px = own[3 * i]; py = own[3 * i + 1]; pz = own[3 * i + 2]; }
    double acc = PASS == 0 ? 1e-9 : 0.0;            // pass A: the CPU starts its row sum at 1e-9 (:49)
    for (int t0 = 0; t0 < n_oth; t0 += EMD_TILE) {
        const int cnt = min(EMD_TILE, n_oth - t0);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt; e += 256) {
            ox[e] = oth[3 * (size_t)(t0 + e)]; oy[e] = oth[3 * (size_t)(t0 + e) + 1]; oz[e] = oth[3 * (size_t)(t0 + e) + 2];
            of[e] = ofac[t0 + e];
        }
        __syncthreads();
        if (live)
            for (int e = 0; e < cnt; ++e) acc += pair_w(level, px, py, pz, ox[e], oy[e], oz[e]) * of[e];
    }
    if (!live) return;
    if (PASS == 0) {
        fL[i] = remL[i] / acc;
    } else if (PASS == 1) {
        const double rr = remR[i];
        const double ss = 1e-9 + rr * acc;
        double r = rr / ss;
        r = r < 1.0 ? r : 1.0;
        const double f = rr * r;
        fR[i] = f;
        const double left = rr - f * acc;
        remR[i] = left > 0.0 ? left : 0.0;
    } else {
        const double left = remL[i] - fL[i] * acc;
        remL[i] = left > 0.0 ? left : 0.0;
    }
}

// Pass C of level li and pass A of level li + 1 in one walk over the other cloud: both are "thread per k, sum over l", pass A
// does not read what pass C writes for other k, and the pair distance -- a third of the per-pair work -- is computed once
// for the two weights.  The two sums run in the same order as in the separate passes: same bits.
__global__ __launch_bounds__(256) void emd_sweep_ca_kernel(int n, int m, int li, double level_c, double level_a, const float *xyz1,
                                                           const float *xyz2, double *temp) {
    __shared__ double ox[EMD_TILE], oy[EMD_TILE], oz[EMD_TILE];
    __shared__ double ofc[EMD_TILE], ofa[EMD_TILE];
    const int c = blockIdx.y;
    double *t = temp + (size_t)c * emd_temp_doubles_per_cloud(n, m);
    double *remL = t, *remR = t + n, *fL = t + (size_t)(n + m) * (1 + li), *fR = fL + n, *fL_next = fL + (n + m);
    const float *own = xyz1 + (size_t)c * n * 3, *oth = xyz2 + (size_t)c * m * 3;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    double px = 0, py = 0, pz = 0;
    if (live) { px = own[3 * i]; py = own[3 * i + 1]; pz = own[3 * i + 2]; }
    double acc_c = 0.0, acc_a = 1e-9;               // pass A: the CPU starts its row sum at 1e-9 (:49)
    for (int t0 = 0; t0 < m; t0 += EMD_TILE) {
        const int cnt = min(EMD_TILE, m - t0);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt; e += 256) {
            ox[e] = oth[3 * (size_t)(t0 + e)]; oy[e] = oth[3 * (size_t)(t0 + e) + 1]; oz[e] = oth[3 * (size_t)(t0 + e) + 2];
            ofc[e] = fR[t0 + e]; ofa[e] = remR[t0 + e];
        }
        __syncthreads();
        if (live)
            for (int e = 0; e < cnt; ++e) {
                const double d2 = (px - ox[e]) * (px - ox[e]) + (py - oy[e]) * (py - oy[e]) + (pz - oz[e]) * (pz - oz[e]);
                acc_c += (double)expf((float)(level_c * d2)) * ofc[e];
                acc_a += (double)expf((float)(level_a * d2)) * ofa[e];
            }
    }
    if (!live) return;
    const double left = remL[i] - fL[i] * acc_c;
    const double rl = left > 0.0 ? left : 0.0;
    remL[i] = rl;
    fL_next[i] = rl / acc_a;
}

// match[c][l][k] = sum over levels of w_j(k,l) fL_j[k] fR_j[l], accumulated in float level by level.
// grid = (n/256, m/32, b): thread = one k, 32 l's.
constexpr int EMD_LT = 32;
struct EmdLevels { double v[EMD_LEVELS]; };

__global__ __launch_bounds__(256) void emd_match_kernel(int n, int m, EmdLevels lv, const float *xyz1, const float *xyz2,
                                                        const double *temp, float *match) {
    __shared__ float qx[EMD_LT], qy[EMD_LT], qz[EMD_LT];
    __shared__ double fr[EMD_LEVELS][EMD_LT];
    const int c = blockIdx.z;
    const double *t = temp + (size_t)c * emd_temp_doubles_per_cloud(n, m);
    const int l0 = blockIdx.y * EMD_LT;
    const int lcnt = min(EMD_LT, m - l0);
    for (int e = threadIdx.x; e < lcnt * (3 + EMD_LEVELS); e += 256) {
        const int l = e % lcnt, what = e / lcnt;
        if (what < 3) (what == 0 ? qx : what == 1 ? qy : qz)[l] = xyz2[((size_t)c * m + l0 + l) * 3 + what];
        else fr[what - 3][l] = t[(size_t)(n + m) * (1 + (what - 3)) + n + l0 + l];
    }
    __syncthreads();
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const float *p = xyz1 + ((size_t)c * n + k) * 3;
    const double px = p[0], py = p[1], pz = p[2];
    double fl[EMD_LEVELS];
#pragma unroll
    for (int j = 0; j < EMD_LEVELS; ++j) fl[j] = t[(size_t)(n + m) * (1 + j) + k];
    for (int l = 0; l < lcnt; ++l) {
        const double x2 = qx[l], y2 = qy[l], z2 = qz[l];
        const double d2 = (px - x2) * (px - x2) + (py - y2) * (py - y2) + (pz - z2) * (pz - z2);
        float mf = 0.f;
#pragma unroll
        for (int j = 0; j < EMD_LEVELS; ++j) {
            const double w = (double)expf((float)(lv.v[j] * d2)) * fr[j][l] * fl[j];
            mf = (float)((double)mf + w);
        }
        match[((size_t)c * m + l0 + l) * n + k] = mf;
    }
}

// cost[c] = sum_{k,l} sqrtf(|q_l - p_k|^2) * match[l][k]  (float product, double sum; :85-105).
// The match matrix is read once: workgroup = EMD_COST_ROWS rows l of one cloud, lanes across k (contiguous in match);
// partial double sums go to a scratch vector and a second launch folds them in a fixed order.
constexpr int EMD_COST_ROWS = 8;
__global__ __launch_bounds__(256) void emd_cost_partial_kernel(int n, int m, const float *xyz1, const float *xyz2,
                                                               const float *match, double *partial) {
    __shared__ double red[4];
    const int c = blockIdx.y, l0 = blockIdx.x * EMD_COST_ROWS;
    const int rows = min(EMD_COST_ROWS, m - l0);
    const float *p = xyz1 + (size_t)c * n * 3, *q = xyz2 + ((size_t)c * m + l0) * 3;
    const float *mt = match + ((size_t)c * m + l0) * n;
    double acc = 0.0;
    for (int k = threadIdx.x; k < n; k += 256) {
        const float px = p[3 * k], py = p[3 * k + 1], pz = p[3 * k + 2];
        float w[EMD_COST_ROWS];
#pragma unroll
        for (int r = 0; r < EMD_COST_ROWS; ++r) w[r] = r < rows ? mt[(size_t)r * n + k] : 0.f;
#pragma unroll
        for (int r = 0; r < EMD_COST_ROWS; ++r)
            if (r < rows) {
                const float dx = q[3 * r] - px, dy = q[3 * r + 1] - py, dz = q[3 * r + 2] - pz;
                const float d = sqrtf(dx * dx + dy * dy + dz * dz) * w[r];
                acc += (double)d;
            }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)c * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void emd_cost_fold_kernel(int parts, const double *partial, float *cost) {
    __shared__ double red[256];
    const double *pp = partial + (size_t)blockIdx.x * parts;
    double acc = 0.0;
    for (int i = threadIdx.x; i < parts; i += 256) acc += pp[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) cost[blockIdx.x] = (float)red[0];
}

// grad1[k] = -sum_l match[l][k] * (q_l - p_k)/max(|q_l - p_k|, 1e-20), l ascending (the CPU's order, :110-133).
constexpr int EMD_G1_AHEAD = 16;
__global__ __launch_bounds__(256) void emd_grad1_kernel(int n, int m, const float *xyz1, const float *xyz2,
                                                        const float *match, float *grad1) {
    __shared__ float qx[EMD_TILE], qy[EMD_TILE], qz[EMD_TILE];
    const int c = blockIdx.y;
    const float *q = xyz2 + (size_t)c * m * 3, *mt = match + (size_t)c * n * m;
    const int k = blockIdx.x * 256 + threadIdx.x;
    const bool live = k < n;
    float px = 0, py = 0, pz = 0;
    if (live) { const float *p = xyz1 + ((size_t)c * n + k) * 3; px = p[0]; py = p[1]; pz = p[2]; }
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int t0 = 0; t0 < m; t0 += EMD_TILE) {
        const int cnt = min(EMD_TILE, m - t0);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt; e += 256) { qx[e] = q[3 * (t0 + e)]; qy[e] = q[3 * (t0 + e) + 1]; qz[e] = q[3 * (t0 + e) + 2]; }
        __syncthreads();
        if (live)
            for (int e0 = 0; e0 < cnt; e0 += EMD_G1_AHEAD) {     // the loads of a group are in flight together; the sum stays in l order
                float w[EMD_G1_AHEAD];
#pragma unroll
                for (int u = 0; u < EMD_G1_AHEAD; ++u) w[u] = e0 + u < cnt ? mt[(size_t)(t0 + e0 + u) * n + k] : 0.f;
#pragma unroll
                for (int u = 0; u < EMD_G1_AHEAD; ++u)
                    if (e0 + u < cnt) {
                        const int e = e0 + u;
                        const float ox = qx[e] - px, oy = qy[e] - py, oz = qz[e] - pz;
                        float d = sqrtf(ox * ox + oy * oy + oz * oz);
                        d = d < 1e-20f ? 1e-20f : d;
                        gx -= w[u] * (ox / d); gy -= w[u] * (oy / d); gz -= w[u] * (oz / d);
                    }
            }
    }
    if (live) { grad1[((size_t)c * n + k) * 3] = gx; grad1[((size_t)c * n + k) * 3 + 1] = gy; grad1[((size_t)c * n + k) * 3 + 2] = gz; }
}

// grad2[l] = sum_k match[l][k] * (q_l - p_k)/max(|.|, 1e-20).  One wave per l (row of match is
// contiguous in k); lanes take k strided, partial sums combined in a fixed butterfly order.
__global__ __launch_bounds__(64) void emd_grad2_kernel(int n, int m, const float *xyz1, const float *xyz2,
                                                       const float *match, float *grad2) {
    const int c = blockIdx.y, l = blockIdx.x, lane = threadIdx.x;
    const float *p = xyz1 + (size_t)c * n * 3;
    const float *qp = xyz2 + ((size_t)c * m + l) * 3;
    const float *row = match + ((size_t)c * m + l) * n;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int k = lane; k < n; k += 64) {
        const float ox = qx - p[3 * k], oy = qy - p[3 * k + 1], oz = qz - p[3 * k + 2];
        float d = sqrtf(ox * ox + oy * oy + oz * oz);
        d = d < 1e-20f ? 1e-20f : d;
        const float w = row[k];
        sx += w * (ox / d); sy += w * (oy / d); sz += w * (oz / d);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { sx += __shfl_xor(sx, off); sy += __shfl_xor(sy, off); sz += __shfl_xor(sz, off); }
    if (lane == 0) {
        float *g = grad2 + ((size_t)c * m + l) * 3;
        g[0] = sx; g[1] = sy; g[2] = sz;
    }
}

}  // namespace geoadv

using namespace geoadv;

extern "C" size_t geoadv_approx_match_temp_floats(int b, int n, int m) {
    if (b <= 0 || n + m <= 0) return 16;
    return 2 * (size_t)b * emd_temp_doubles_per_cloud(n, m) + 16;
}

static int emd_check(const char *op, int b, int n, int m) {
    GA_REQUIRE(b >= 0 && n >= 1 && m >= 1, "%s: needs b >= 0 and at least one point per cloud (b=%d n=%d m=%d)", op, b, n, m);
    GA_REQUIRE(b <= 65535, "%s: batch %d exceeds 65535", op, b);
    GA_REQUIRE((size_t)n * m <= ((size_t)1 << 31), "%s: n*m too large", op);
    return GEOADV_OK;
}

extern "C" int geoadv_approx_match(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp,
                                   void *stream) {
    if (int rc = emd_check("approx_match", b, n, m)) return rc;
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(xyz1 && xyz2 && match && temp, "approx_match: null pointer");
    hipStream_t st = as_stream(stream);
    double *t = reinterpret_cast<double *>((reinterpret_cast<size_t>(temp) + 7) & ~(size_t)7);
    emd_init_kernel<<<dim3(cdiv(n + m, 256), b), 256, 0, st>>>(n, m, t);
    GA_LAUNCH_CHECK();
    EmdLevels lv;
    for (int li = 0; li < EMD_LEVELS; ++li) lv.v[li] = emd_level(li);
    emd_sweep_kernel<0><<<dim3(cdiv(n, 256), b), 256, 0, st>>>(n, m, 0, lv.v[0], xyz1, xyz2, t);
    for (int li = 0; li < EMD_LEVELS; ++li) {
        emd_sweep_kernel<1><<<dim3(cdiv(m, 256), b), 256, 0, st>>>(n, m, li, lv.v[li], xyz1, xyz2, t);
        if (li + 1 < EMD_LEVELS)                       // pass C of this level with pass A of the next
            emd_sweep_ca_kernel<<<dim3(cdiv(n, 256), b), 256, 0, st>>>(n, m, li, lv.v[li], lv.v[li + 1], xyz1, xyz2, t);
        else
            emd_sweep_kernel<2><<<dim3(cdiv(n, 256), b), 256, 0, st>>>(n, m, li, lv.v[li], xyz1, xyz2, t);
        GA_LAUNCH_CHECK();
    }
    emd_match_kernel<<<dim3(cdiv(n, 256), cdiv(m, EMD_LT), b), 256, 0, st>>>(n, m, lv, xyz1, xyz2, t, match);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

extern "C" int geoadv_match_cost(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match, float *out,
                                 void *stream) {
    if (int rc = emd_check("match_cost", b, n, m)) return rc;
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(xyz1 && xyz2 && match && out, "match_cost: null pointer");
    hipStream_t st = as_stream(stream);
    const int parts = cdiv(m, EMD_COST_ROWS);
    double *partial = nullptr;                        // stream-ordered scratch: concurrent callers never share it
    GA_HIP(hipMallocAsync(reinterpret_cast<void **>(&partial), (size_t)b * parts * sizeof(double), st));
    emd_cost_partial_kernel<<<dim3(parts, b), 256, 0, st>>>(n, m, xyz1, xyz2, match, partial);
    emd_cost_fold_kernel<<<b, 256, 0, st>>>(parts, partial, out);
    const hipError_t launched = hipGetLastError();
    GA_HIP(hipFreeAsync(partial, st));
    GA_HIP(launched);
    return GEOADV_OK;
}

extern "C" int geoadv_match_cost_grad(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                                      float *grad1, float *grad2, void *stream) {
    if (int rc = emd_check("match_cost_grad", b, n, m)) return rc;
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(xyz1 && xyz2 && match && grad1 && grad2, "match_cost_grad: null pointer");
    hipStream_t st = as_stream(stream);
    emd_grad1_kernel<<<dim3(cdiv(n, 256), b), 256, 0, st>>>(n, m, xyz1, xyz2, match, grad1);
    GA_LAUNCH_CHECK();
    emd_grad2_kernel<<<dim3(m, b), 64, 0, st>>>(n, m, xyz1, xyz2, match, grad2);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}
