// Chamfer nearest-neighbour distance (NnDistance / NnDistanceGrad) for gfx950.
//
// Replaces NmDistanceKernelLauncher / NmDistanceGradKernelLauncher
// (external/structural_losses/tf_nndistance.cpp:168,208; kernels tf_nndistance_g.cu:5-157).
// Results reproduce the reference CPU op bit for bit (tf_nndistance.cpp:21-43, 126-163):
//   d = ((dx*dx) + (dy*dy)) + (dz*dz) in fp32 with every operation rounded on its own (this
//   file is compiled with -ffp-contract=off and carries the pragma below), strict '<' so the
//   lowest target index wins ties, and a CPU-ordered (deterministic) gradient accumulation.
//
// Forward layout: one workgroup = 4 waves x 64 lanes; every lane keeps R query points in
// registers; the target cloud is staged into LDS as three SoA planes (x[], y[], z[]) and read
// back with wave-uniform (broadcast) ds_read_b128; wave w scans the w-th quarter of each stage.
// The argmin is tracked per CHUNK of 8 targets (one v_min per pair, one compare per chunk) and
// the winning chunk is re-scanned at the end for the first index whose distance equals the
// minimum -- exact, and ~25 % fewer VALU instructions than compare+select per pair.
// The kernel is VALU-issue bound (8 unfusable fp32 ops + 1 min per pair); HBM traffic is the
// 20*B*(N+M) algorithmic bytes.
#include "common.h"
#include "chamfer_grad.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>

#pragma clang fp contract(off)

namespace geoadv {

struct ChamferScan {
    const float *query;   // [b, nq, 3]
    const float *target;  // [b, nt, 3]
    float *dist;          // [b, nq]
    int *idx;             // [b, nq]
    int nq, nt;
    const int *need;      // null = every cloud; else int[8 * b] (16-byte aligned): cloud c is scanned only if one of its 8
                          // flags is set (a workgroup of the paired grid search, chamfer_grid.hip, gave up on it)
};
struct ChamferArgs {
    ChamferScan s[4];
    int tiles, clouds, scans;   // 1-D grid decomposition (XCD aware, see the kernel)
};

constexpr int CH_CHUNK = 8;        // targets per argmin chunk
constexpr int CH_STAGE = 2048;     // targets per LDS stage: 3 planes x 8 KB (24 KB per workgroup, so that two
                                   // workgroups fit next to an encoder workgroup on one CU)

__device__ __forceinline__ float sqdist(float tx, float ty, float tz, float qx, float qy, float qz) {
    const float dx = tx - qx, dy = ty - qy, dz = tz - qz;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    return (xx + yy) + zz;
}

// __launch_bounds__(256, 4): at most 128 VGPRs, so 4 workgroups (16 waves) fit per CU and the 1024
// workgroups of the B = 32, N = 2048 attack step are all resident at once (no partial second round).
// WAVES = 4 (256 threads) is the throughput shape; WAVES = 8 keeps R = 4 when there are only 384..767 query tiles (B = 32,
// N = 2048 through the public op: 52.6 -> 49.6 us against R = 2 on 4 waves).  WAVES = 16 (R = 1 only): the same 64 queries against the same staged
// targets with each wave walking a sixteenth of them -- for launches too small to fill the chip (the attack loop at
// B <= 8), where a workgroup's run time is one wave's serial walk (17 us for 512 targets) and nothing else is waiting
// for the CU.
template <int R, int WAVES>
__global__ __launch_bounds__(kWave * WAVES, 16 / WAVES) void chamfer_scan_kernel(ChamferArgs args) {
    constexpr int CH_THREADS = kWave * WAVES, CH_WAVES = WAVES;
    GA_STAMP(0, 0);
    // XCD-aware block mapping: workgroups are dealt round-robin over the 8 XCDs (each with its own
    // L2), so all query tiles of one (scan, cloud) group -- which stream the same target cloud -- are
    // given the same `blockIdx % 8`: the cloud is then fetched into ONE L2 instead of eight.
    const int lin = blockIdx.x;
    const int xcd = lin & 7, slot = lin >> 3;
    const int group = (slot / args.tiles) * 8 + xcd, tile = slot % args.tiles;
    if (group >= args.clouds * args.scans) return;
    const ChamferScan sc = args.s[group / args.clouds];
    const int nq = sc.nq, nt = sc.nt;
    const int q0 = tile * (kWave * R);
    if (q0 >= nq) return;
    const int c = group % args.clouds;
    if (sc.need) {
        const int4 lo = reinterpret_cast<const int4 *>(sc.need)[2 * c], hi = reinterpret_cast<const int4 *>(sc.need)[2 * c + 1];
        if ((lo.x | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) == 0) return;
    }
    const float *Q = sc.query + (size_t)c * nq * 3;
    const float *T = sc.target + (size_t)c * nt * 3;

    __shared__ __attribute__((aligned(16))) float stage[3 * CH_STAGE];
    float *sx = stage, *sy = stage + CH_STAGE, *sz = stage + 2 * CH_STAGE;
    // the wave-merge arrays alias the stage planes (dead after the last scan)
    float (*mdist)[kWave * R] = reinterpret_cast<float (*)[kWave * R]>(stage);
    int (*midx)[kWave * R] = reinterpret_cast<int (*)[kWave * R]>(stage + CH_WAVES * kWave * R);
    static_assert(2 * CH_WAVES * kWave * R <= 3 * CH_STAGE, "merge arrays must fit in the stage buffer");

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;

    float qx[R], qy[R], qz[R], best[R];
    int bestk[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int qi = q0 + r * kWave + lane;
        qi = qi < nq ? qi : nq - 1;
        qx[r] = Q[3 * qi];
        qy[r] = Q[3 * qi + 1];
        qz[r] = Q[3 * qi + 2];
        best[r] = INFINITY;
        bestk[r] = -1;
    }

    for (int t0 = 0; t0 < nt; t0 += CH_STAGE) {
        const int cnt = min(CH_STAGE, nt - t0);
        const int cntp = (cnt + CH_CHUNK - 1) / CH_CHUNK * CH_CHUNK;
        __syncthreads();   // the previous stage has been consumed
        for (int e = threadIdx.x; e < cntp; e += CH_THREADS) {
            float x = INFINITY, y = INFINITY, z = INFINITY;   // padding: distance +inf, never wins
            if (e < cnt) {
                x = T[3 * (size_t)(t0 + e)];
                y = T[3 * (size_t)(t0 + e) + 1];
                z = T[3 * (size_t)(t0 + e) + 2];
            }
            sx[e] = x; sy[e] = y; sz[e] = z;
        }
        __syncthreads();
        GA_STAMP(0, 1);
        const int nchunks = cntp / CH_CHUNK;
        const int cbeg = nchunks * wave / CH_WAVES, cend = nchunks * (wave + 1) / CH_WAVES;
        if (cbeg < cend) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (bestk[r] < 0) bestk[r] = t0 + cbeg * CH_CHUNK;
        }
        for (int ch = cbeg; ch < cend; ++ch) {
            const int k0 = ch * CH_CHUNK;
            float tx[CH_CHUNK], ty[CH_CHUNK], tz[CH_CHUNK];
#pragma unroll
            for (int v = 0; v < CH_CHUNK / 4; ++v) {
                const float4 a = *reinterpret_cast<const float4 *>(&sx[k0 + 4 * v]);
                const float4 bb = *reinterpret_cast<const float4 *>(&sy[k0 + 4 * v]);
                const float4 cc = *reinterpret_cast<const float4 *>(&sz[k0 + 4 * v]);
                tx[4 * v] = a.x; tx[4 * v + 1] = a.y; tx[4 * v + 2] = a.z; tx[4 * v + 3] = a.w;
                ty[4 * v] = bb.x; ty[4 * v + 1] = bb.y; ty[4 * v + 2] = bb.z; ty[4 * v + 3] = bb.w;
                tz[4 * v] = cc.x; tz[4 * v + 1] = cc.y; tz[4 * v + 2] = cc.z; tz[4 * v + 3] = cc.w;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float cm = sqdist(tx[0], ty[0], tz[0], qx[r], qy[r], qz[r]);
#pragma unroll
                for (int u = 1; u < CH_CHUNK; ++u)
                    cm = fminf(cm, sqdist(tx[u], ty[u], tz[u], qx[r], qy[r], qz[r]));
                if (cm < best[r]) {       // strict: an equal later chunk never replaces an earlier one
                    best[r] = cm;
                    bestk[r] = t0 + k0;
                }
            }
        }
    }

    GA_STAMP(0, 2);
    // Re-scan the winning chunk of every query for the first index attaining the minimum.  With a single LDS stage
    // (nt <= 2048: the attack's shape) the chunk is still in the stage planes: two ds_read_b128 per plane, hits taken in
    // DESCENDING order so that the last one kept is the lowest index (padding is +inf and never equals a finite minimum).
    int found[R];
    const bool staged = nt <= CH_STAGE;                    // uniform
    if (staged) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            found[r] = INT_MAX;
            if (bestk[r] < 0) continue;                    // (uniform: a wave either scanned targets or did not)
            const int kb = bestk[r];
            const float4 xa = *reinterpret_cast<const float4 *>(&sx[kb]), xb = *reinterpret_cast<const float4 *>(&sx[kb + 4]);
            const float4 ya = *reinterpret_cast<const float4 *>(&sy[kb]), yb = *reinterpret_cast<const float4 *>(&sy[kb + 4]);
            const float4 za = *reinterpret_cast<const float4 *>(&sz[kb]), zb = *reinterpret_cast<const float4 *>(&sz[kb + 4]);
            const float tx[CH_CHUNK] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
            const float ty[CH_CHUNK] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
            const float tz[CH_CHUNK] = {za.x, za.y, za.z, za.w, zb.x, zb.y, zb.z, zb.w};
            int f = kb;
#pragma unroll
            for (int u = CH_CHUNK - 1; u >= 0; --u)
                f = sqdist(tx[u], ty[u], tz[u], qx[r], qy[r], qz[r]) == best[r] ? kb + u : f;
            found[r] = f;
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            found[r] = INT_MAX;
            if (bestk[r] >= 0) {
                found[r] = bestk[r];
                bool hit = false;
                for (int u = 0; u < CH_CHUNK; ++u) {
                    const int k = bestk[r] + u;
                    if (k < nt) {
                        const float d = sqdist(T[3 * (size_t)k], T[3 * (size_t)k + 1], T[3 * (size_t)k + 2], qx[r], qy[r], qz[r]);
                        if (!hit && d == best[r]) { hit = true; found[r] = k; }
                    }
                }
            }
        }
    }
    __syncthreads();   // every wave is done with the stage planes: they become the merge arrays
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mdist[wave][r * kWave + lane] = best[r];
        midx[wave][r * kWave + lane] = found[r];
    }
    __syncthreads();
    for (int qq = threadIdx.x; qq < kWave * R; qq += CH_THREADS) {
        float d = mdist[0][qq];
        int k = midx[0][qq];
#pragma unroll
        for (int w = 1; w < CH_WAVES; ++w) {
            const float dw = mdist[w][qq];
            const int kw = midx[w][qq];
            if (dw < d || (dw == d && kw < k)) { d = dw; k = kw; }
        }
        if (q0 + qq < nq) {
            sc.dist[(size_t)c * nq + q0 + qq] = d;
            sc.idx[(size_t)c * nq + q0 + qq] = k;
        }
    }
    GA_STAMP(0, 7);
}

// m == 0: the reference CPU loop leaves best = 0, besti = 0 (tf_nndistance.cpp:27-28,39-40).
__global__ void chamfer_fill_empty_kernel(float *dist, int *idx, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) { dist[i] = 0.f; idx[i] = 0; }
}

// Non-finite inputs, the reference's rule (tf_nndistance.cpp:31-40): candidate 0 is ALWAYS taken (`k==0 || d<best`) and nothing
// compares below a NaN, so a query whose distance to candidate 0 is NaN keeps (NaN, 0), a NaN distance to a later candidate never
// wins, and an infinite minimum keeps its lowest index.  The fast kernels are written for finite clouds (fminf / integer minima
// skip NaNs, the grid search sorts points into cells): the OPERATOR entry points therefore finish with this launch -- one
// workgroup per cloud looks for a non-finite coordinate in either cloud (one pass over 12 (n + m) bytes) and, only if there is
// one, recomputes both directions of that cloud with the reference's own loop, a thread per query.  (The attack loop's internal
// launches skip it: a non-finite cloud there is a diverged attack whose losses are NaN either way.)
__global__ __launch_bounds__(1024) void nn_nonfinite_redo_kernel(int n, int m, const float *xyz1, const float *xyz2, float *dist1,
                                                                 int *idx1, float *dist2, int *idx2) {
    const int c = blockIdx.x, t = threadIdx.x;
    const float *P = xyz1 + (size_t)c * n * 3, *Q = xyz2 + (size_t)c * m * 3;
    int bad = 0;
    for (int i = t; i < 3 * n; i += 1024) bad |= !(fabsf(P[i]) <= 3.402823466e38f);
    for (int i = t; i < 3 * m; i += 1024) bad |= !(fabsf(Q[i]) <= 3.402823466e38f);
    if (!__syncthreads_or(bad)) return;
    for (int dir = 0; dir < 2; ++dir) {
        const float *A = dir ? Q : P, *T = dir ? P : Q;
        const int na = dir ? m : n, nt = dir ? n : m;
        float *dist = (dir ? dist2 : dist1) + (size_t)c * na;
        int *idx = (dir ? idx2 : idx1) + (size_t)c * na;
        for (int j = t; j < na; j += 1024) {
            const float x = A[3 * (size_t)j], y = A[3 * (size_t)j + 1], z = A[3 * (size_t)j + 2];
            float best = sqdist(T[0], T[1], T[2], x, y, z);
            int bi = 0;
            for (int k = 1; k < nt; ++k) {
                const float d = sqdist(T[3 * (size_t)k], T[3 * (size_t)k + 1], T[3 * (size_t)k + 2], x, y, z);
                if (d < best) { best = d; bi = k; }
            }
            dist[j] = best;
            idx[j] = bi;
        }
    }
}
int launch_nn_nonfinite_fix(int b, int n, const float *xyz1, int m, const float *xyz2, float *dist1, int *idx1, float *dist2,
                            int *idx2, hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0) return GEOADV_OK;
    nn_nonfinite_redo_kernel<<<b, 1024, 0, stream>>>(n, m, xyz1, xyz2, dist1, idx1, dist2, idx2);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

int launch_chamfer_scans(const ChamferScan *scans, int nscan, int b, hipStream_t stream) {
    if (b <= 0) return GEOADV_OK;
    ChamferArgs args;
    int live = 0, maxq = 0;
    for (int i = 0; i < nscan; ++i) {
        const ChamferScan &s = scans[i];
        if (s.nq <= 0) continue;
        if (s.nt <= 0) {
            const size_t count = (size_t)b * s.nq;
            chamfer_fill_empty_kernel<<<(unsigned)((count + 255) / 256), 256, 0, stream>>>(s.dist, s.idx, count);
            GA_LAUNCH_CHECK();
            continue;
        }
        args.s[live++] = s;
        maxq = s.nq > maxq ? s.nq : maxq;
    }
    if (!live) return GEOADV_OK;
    // Pick the register blocking so that the grid still fills 256 CUs x 4 workgroups.
    auto groups = [&](int R) { return (long)cdiv(maxq, kWave * R) * b * live; };
    args.clouds = b; args.scans = live;
    auto grid = [&](int R) {
        args.tiles = cdiv(maxq, kWave * R);
        return dim3((unsigned)(args.tiles * 8 * cdiv(b * live, 8)));
    };
    if (groups(4) >= 768) {
        const dim3 g = grid(4);
        chamfer_scan_kernel<4, 4><<<g, 256, 0, stream>>>(args);
    } else if (groups(4) >= 384) {                         // half as many workgroups of twice the waves: still 4 queries per lane
        const dim3 g = grid(4);
        chamfer_scan_kernel<4, 8><<<g, 512, 0, stream>>>(args);
    } else if (groups(2) >= 768) {
        const dim3 g = grid(2);
        chamfer_scan_kernel<2, 4><<<g, 256, 0, stream>>>(args);
    } else if (groups(1) > 1024) {
        const dim3 g = grid(1);
        chamfer_scan_kernel<1, 4><<<g, 256, 0, stream>>>(args);
    } else {                                               // fewer workgroups than the chip holds: shorten each one
        const dim3 g = grid(1);
        chamfer_scan_kernel<1, 16><<<g, 1024, 0, stream>>>(args);
    }
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

}  // namespace geoadv

namespace geoadv {

// ------------------------------------------------------------------------------------------
// Gradient (NnDistanceGrad).  One workgroup per (cloud, output side); see chamfer_grad.h.
// ------------------------------------------------------------------------------------------
constexpr int CG_THREADS = 512;

__global__ __launch_bounds__(CG_THREADS) void chamfer_grad_kernel(
    int n, const float *xyz1, int m, const float *xyz2, const float *gd1, const int *idx1,
    const float *gd2, const int *idx2, float *g1, float *g2, int P1, int P2) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int c = blockIdx.x;
    const float *p = xyz1 + (size_t)c * n * 3, *q = xyz2 + (size_t)c * m * 3;
    GradSide s;
    s.jstar = -1; s.extra = 0.f; s.gd_own_s = 0.f; s.gd_oth_s = 0.f;
    if (blockIdx.y == 0) {
        if (!g1) return;
        s.n_own = n; s.n_oth = m; s.own = p; s.oth = q;
        s.match_own = idx1 + (size_t)c * n; s.match_oth = idx2 + (size_t)c * m;
        s.gd_own = gd1 + (size_t)c * n; s.gd_oth = gd2 + (size_t)c * m;
        s.gout = g1 + (size_t)c * n * 3;
        chamfer_grad_side<true, CG_THREADS>(s, lds, P2);
    } else {
        if (!g2) return;
        s.n_own = m; s.n_oth = n; s.own = q; s.oth = p;
        s.match_own = idx2 + (size_t)c * m; s.match_oth = idx1 + (size_t)c * n;
        s.gd_own = gd2 + (size_t)c * m; s.gd_oth = gd1 + (size_t)c * n;
        s.gout = g2 + (size_t)c * m * 3;
        chamfer_grad_side<false, CG_THREADS>(s, lds, P1);
    }
}

static int pow2_at_least(int v) {
    int p = 2;
    while (p < v) p <<= 1;
    return p;
}
static size_t grad_lds_bytes(int P) { return sizeof(unsigned) * (size_t)P * (P <= CG_TERMS_MAX_P ? 4 : 1); }

int launch_chamfer_grad(int b, int n, const float *xyz1, int m, const float *xyz2, const float *gd1,
                        const int *idx1, const float *gd2, const int *idx2, float *g1, float *g2,
                        hipStream_t stream) {
    if (b <= 0 || (n <= 0 && m <= 0)) return GEOADV_OK;
    GA_REQUIRE(n <= 32768 && m <= 32768, "nn_distance_grad supports at most 32768 points per cloud (got n=%d m=%d)", n, m);
    if (n == 0 || m == 0) {   // nothing matches anything: both gradients are zero
        if (g1 && n) GA_HIP(hipMemsetAsync(g1, 0, sizeof(float) * (size_t)b * n * 3, stream));
        if (g2 && m) GA_HIP(hipMemsetAsync(g2, 0, sizeof(float) * (size_t)b * m * 3, stream));
        return GEOADV_OK;
    }
    const int P1 = pow2_at_least(n), P2 = pow2_at_least(m);
    const size_t l1 = grad_lds_bytes(P1), l2 = grad_lds_bytes(P2);
    const size_t lds = l1 > l2 ? l1 : l2;
    static DeviceOnce attr;
    if (int rc = attr.run([]() -> int {
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chamfer_grad_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
            return GEOADV_OK;
        })) return rc;
    chamfer_grad_kernel<<<dim3(b, g2 ? 2 : 1), CG_THREADS, lds, stream>>>(n, xyz1, m, xyz2, gd1, idx1, gd2, idx2,
                                                                         g1, g2, P1, P2);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

}  // namespace geoadv

using namespace geoadv;

extern "C" int geoadv_nn_distance(int b, int n, const float *xyz1, int m, const float *xyz2, float *dist1,
                                  int *idx1, float *dist2, int *idx2, void *stream) {
    GA_REQUIRE(b >= 0 && n >= 0 && m >= 0, "nn_distance: negative dimension (b=%d n=%d m=%d)", b, n, m);
    GA_REQUIRE(b <= 65535, "nn_distance: batch %d exceeds 65535", b);
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE((n == 0 || (xyz1 && dist1 && idx1)) && (m == 0 || (xyz2 && dist2 && idx2)), "nn_distance: null pointer");
    ChamferScan scans[2] = {{xyz1, xyz2, dist1, idx1, n, m}, {xyz2, xyz1, dist2, idx2, m, n}};
    if (int rc = launch_chamfer_scans(scans, 2, b, as_stream(stream))) return rc;
    return launch_nn_nonfinite_fix(b, n, xyz1, m, xyz2, dist1, idx1, dist2, idx2, as_stream(stream));
}

extern "C" int geoadv_nn_distance_grad(int b, int n, const float *xyz1, int m, const float *xyz2,
                                       const float *grad_dist1, const int *idx1, const float *grad_dist2,
                                       const int *idx2, float *grad_xyz1, float *grad_xyz2, void *stream) {
    GA_REQUIRE(b >= 0 && n >= 0 && m >= 0, "nn_distance_grad: negative dimension");
    GA_REQUIRE(b <= 65535, "nn_distance_grad: batch %d exceeds 65535", b);
    return launch_chamfer_grad(b, n, xyz1, m, xyz2, grad_dist1, idx1, grad_dist2, idx2, grad_xyz1, grad_xyz2,
                               as_stream(stream));
}
GA_STAMPS_GETTER(geoadv_debug_stamps_chamfer)
