// The loop's per-cloud losses / metrics / keep-best and its two Chamfer gradients as DEVICE BODIES (moved out of attack.hip in round
// 6): one workgroup of the fused pass = loss_cgrad_block.  attack.hip launches them as loss_cgrad_kernel; chamfer_sym.hip hosts the
// same workgroups as the LAST riders of the symmetric scan's launch (LossRider): they wait for their cloud's scan and search
// workgroups through a per-cloud arrival counter instead of a kernel boundary.
#pragma once
#include "common.h"
#include "chamfer_sym.h"
#include <limits.h>
#include <math.h>

#ifndef LC_STAMP
#define LC_STAMP(K, I)
#endif

namespace geoadv {

// ------------------------------------------------------------------------------------------
// Per-cloud losses + metrics history + keep-best.  grid = clouds, 256 threads.
// ------------------------------------------------------------------------------------------
struct LossArgs {
    int n;                         // points per cloud (n_input == n_output)
    int loss_adv_type, loss_dist_type;
    float mp_pert_w, mp_dist_w;
    const float *r1, *r2, *a1, *a2;   // [B][n] squared NN distances (recon->gt, gt->recon, adv->x, x->adv)
    const float *pert;                // [B][n][3]
    const float *z, *tz;              // [B][128]
    const float *w;                   // [B] dist_weight
    const float *emd_cost;            // [B] match_cost(recon, gt) or null
    float emd_weight;                 // loss_adv += emd_weight * emd_cost / n  (build-defined)
    float *losses;                    // [8][B]: loss_adv, loss_dist, loss_pert, loss_max|max_dist, input_dist, loss_ae, loss_max(pert), max_dist
    int *jstar;                       // [B] argmax_j a1 (first), [B] argmax_n |pert_n|^2 (first)
    float *dz_latent;                 // [B][128] d loss_adv / d z in latent mode (else untouched)
    float *hist;                      // [6][B] slot of this iteration or null
    int keep;                         // 1: take part in the keep-best update
    float *best_err;                  // [B]
    float *best_metrics;              // [B][4]
    const float *adv, *recon;         // [B][n][3]
    float *best_adv, *best_recon;     // [B][n][3]
    // the symmetric scan's row minima still in one partial per column slice (chamfer_sym.h; slices <= 1: r1 / a1 are final):
    // r1 of every cloud, a1 of the clouds the all-pairs kernel computed (a1_all, or their `a1_need` flags).  This block forms
    // the minima on its way in and leaves them in r1 / a1 (mutable here for that reason).
    SymPartials part;
    const int *a1_need; int a1_all;
    float *r1_out, *a1_out;
};

// One pass for everything: 5 sums and 2 (max, lowest index) pairs per thread, reduced across the
// wave with shuffles and across the 4 waves through LDS -- two barriers instead of ~60.
struct CloudRed { float s1, s2, s3, s4, sp, ma, mp; int ja, jp; };

__device__ __forceinline__ void argmax_merge(float &v, int &i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

__device__ __forceinline__ CloudRed block_reduce(CloudRed r, float (*shf)[8], int (*shi)[2]) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        r.s1 += __shfl_xor(r.s1, off); r.s2 += __shfl_xor(r.s2, off); r.s3 += __shfl_xor(r.s3, off);
        r.s4 += __shfl_xor(r.s4, off); r.sp += __shfl_xor(r.sp, off);
        argmax_merge(r.ma, r.ja, __shfl_xor(r.ma, off), __shfl_xor(r.ja, off));
        argmax_merge(r.mp, r.jp, __shfl_xor(r.mp, off), __shfl_xor(r.jp, off));
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        shf[wave][0] = r.s1; shf[wave][1] = r.s2; shf[wave][2] = r.s3; shf[wave][3] = r.s4; shf[wave][4] = r.sp;
        shf[wave][5] = r.ma; shf[wave][6] = r.mp; shi[wave][0] = r.ja; shi[wave][1] = r.jp;
    }
    __syncthreads();
    CloudRed t;
    t.s1 = ((shf[0][0] + shf[1][0]) + shf[2][0]) + shf[3][0];
    t.s2 = ((shf[0][1] + shf[1][1]) + shf[2][1]) + shf[3][1];
    t.s3 = ((shf[0][2] + shf[1][2]) + shf[2][2]) + shf[3][2];
    t.s4 = ((shf[0][3] + shf[1][3]) + shf[2][3]) + shf[3][3];
    t.sp = ((shf[0][4] + shf[1][4]) + shf[2][4]) + shf[3][4];
    t.ma = shf[0][5]; t.ja = shi[0][0]; t.mp = shf[0][6]; t.jp = shi[0][1];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        argmax_merge(t.ma, t.ja, shf[w][5], shi[w][0]);
        argmax_merge(t.mp, t.jp, shf[w][6], shi[w][1]);
    }
    return t;
}

__device__ __forceinline__ float wave4_sum128(float v, float *sh2) {   // sum over threads 0..127 (others pass 0)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) sh2[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh2[0] + sh2[1];
}

constexpr int LOSS_PRE_MAX_N = 2048;           // deferred row partials exist for clouds of one row super-tile only (chamfer_sym.hip)

// The symmetric scan's row partials of cloud b folded by ALL threads of the workgroup (nthreads; one pass at n = 2048 on 512)
// into LDS -- m1 = r1, m3 = a1 (only where the all-pairs kernel computed it) -- and left in r1 / a1 for later readers; the
// caller puts a workgroup barrier between this and loss_metrics_body, which then sums them in its own fixed order.
__device__ __forceinline__ void loss_premerge(const LossArgs &a, const int b, const int B, const int nthreads, float *m1, float *m3) {
    const int n = a.n;
    const bool part3 = a.a1_all || (a.a1_need && sym_needed(a.a1_need, b));
    const size_t sl = (size_t)a.part.slices * n, o = (size_t)b * n;
    const float *p1 = a.part.rowpart_d + (size_t)b * sl, *p3 = a.part.rowpart_d + ((size_t)B + b) * sl;
    const unsigned long long *w1 = a.part.row64 + o, *w3 = a.part.row64 + (size_t)B * n + o;      // packed form (small batches)
    for (int j = threadIdx.x; j < n; j += nthreads) {
        float v1, v3 = 0.f;
        if (a.part.row64) {
            v1 = __uint_as_float((unsigned)(w1[j] >> 32));
            if (part3) v3 = __uint_as_float((unsigned)(w3[j] >> 32));
        } else {
            v1 = sym_merge_min(p1 + j, a.part.slices, n);
            if (part3) v3 = sym_merge_min(p3 + j, a.part.slices, n);
        }
        m1[j] = v1; a.r1_out[o + j] = v1;
        if (part3) { m3[j] = v3; a.a1_out[o + j] = v3; }
    }
}

// cloud b of B; executed by threads 0..255 of the workgroup (whole waves beyond that may have exited).  m1 / m3: loss_premerge's
// LDS arrays when a.part.deferred (else unused)
__device__ __forceinline__ void loss_metrics_body(const LossArgs &a, const int b, const int B, const float *m1 = nullptr, const float *m3 = nullptr) {
    __shared__ float shf[4][8];
    __shared__ int shi[4][2];
    __shared__ float sh2[4];
    __shared__ int take;
    const int t = threadIdx.x, n = a.n;
    const size_t o = (size_t)b * n;
    CloudRed r;
    r.s1 = r.s2 = r.s3 = r.s4 = r.sp = 0.f;
    r.ma = r.mp = -1.f;
    r.ja = r.jp = INT_MAX;
    const bool part1 = a.part.deferred;                                                  // (uniform)
    const bool part3 = part1 && (a.a1_all || (a.a1_need && sym_needed(a.a1_need, b)));
    constexpr int U = 4;                                  // points per thread and pass: all 7 * U loads requested first
    for (int j0 = t; j0 < n; j0 += U * 256) {             // (same order of accumulation as one point per pass)
        float v1[U], v2[U], v3[U], v4[U], vx[U], vy[U], vz[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * 256 < n ? j0 + u * 256 : t;
            v1[u] = part1 ? m1[j] : a.r1[o + j];
            v3[u] = part3 ? m3[j] : a.a1[o + j];
            v2[u] = a.r2[o + j]; v4[u] = a.a2[o + j];
            vx[u] = a.pert[(o + j) * 3]; vy[u] = a.pert[(o + j) * 3 + 1]; vz[u] = a.pert[(o + j) * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * 256;
            if (j >= n) continue;
            r.s1 += v1[u]; r.s2 += v2[u];
            const float d = v3[u];
            r.s3 += d; r.s4 += v4[u];
            if (d > r.ma) { r.ma = d; r.ja = j; }
            const float p2 = (vx[u] * vx[u] + vy[u] * vy[u]) + vz[u] * vz[u];
            r.sp += p2;
            if (p2 > r.mp) { r.mp = p2; r.jp = j; }
        }
    }
    r = block_reduce(r, shf, shi);
    float ma = r.ma, mp = r.mp;
    const int ja = r.ja, jp = r.jp;
    const float inv_n = 1.0f / (float)n;
    const float loss_ae = r.s1 * inv_n + r.s2 * inv_n;                                 // adv_ae.py:121
    const float input_dist = r.s3 * inv_n + r.s4 * inv_n;                              // adv_ae.py:132, max :133
    const float pert_sq = r.sp;                                                        // adversary.py:41-44
    const float loss_pert = sqrtf(pert_sq), loss_max = sqrtf(mp);                      // adversary.py:47,50
    float loss_adv = loss_ae;
    if (a.emd_cost) loss_adv = loss_ae + a.emd_weight * (a.emd_cost[b] * inv_n);
    if (a.loss_adv_type == GEOADV_LOSS_ADV_LATENT) {                                   // adv_ae.py:107-116
        float d = 0.f;
        if (t < 128) { d = a.z[(size_t)b * 128 + t] - a.tz[(size_t)b * 128 + t]; }
        const float nsq = wave4_sum128(d * d, sh2);
        loss_adv = sqrtf(nsq);
        if (t < 128) a.dz_latent[(size_t)b * 128 + t] = d / loss_adv;
    }
    float loss_dist;
    if (a.loss_dist_type == GEOADV_LOSS_DIST_PERT)
        loss_dist = a.mp_pert_w > 0.f ? loss_pert + a.mp_pert_w * loss_max : loss_pert;   // adv_ae.py:93-97
    else
        loss_dist = a.mp_dist_w > 0.f ? input_dist + a.mp_dist_w * ma : input_dist;       // adv_ae.py:98-102
    const float fourth = a.loss_dist_type == GEOADV_LOSS_DIST_PERT ? loss_max : ma;       // adv_ae.py:204-207
    if (t == 0) {
        a.losses[0 * B + b] = loss_adv; a.losses[1 * B + b] = loss_dist; a.losses[2 * B + b] = loss_pert;
        a.losses[3 * B + b] = fourth;   a.losses[4 * B + b] = input_dist; a.losses[5 * B + b] = loss_ae;
        a.losses[6 * B + b] = loss_max; a.losses[7 * B + b] = ma;
        a.jstar[b] = ja; a.jstar[B + b] = jp;
        if (a.hist) {
            a.hist[0 * B + b] = loss_adv; a.hist[1 * B + b] = loss_dist; a.hist[2 * B + b] = loss_pert;
            a.hist[3 * B + b] = fourth;   a.hist[4 * B + b] = input_dist; a.hist[5 * B + b] = loss_ae;
        }
        int tk = 0;
        if (a.keep && loss_ae < a.best_err[b]) {                                          // adv_ae.py:239 (strict)
            tk = 1;
            a.best_err[b] = loss_ae;
            a.best_metrics[b * 4 + 0] = loss_adv; a.best_metrics[b * 4 + 1] = loss_dist;
            a.best_metrics[b * 4 + 2] = input_dist; a.best_metrics[b * 4 + 3] = loss_ae;   // nre = this / ref, at read-out
        }
        take = tk;
    }
    __syncthreads();
    if (take) {
        const size_t base = o * 3;
        for (int e = t; e < 3 * n; e += 256) {
            a.best_adv[base + e] = a.adv[base + e];
            a.best_recon[base + e] = a.recon[base + e];
        }
    }
}

struct CGradArgs { CGradProblem pr[2]; int n, P; };
constexpr int CGA_THREADS = 512;

// Fast variant for the loop (n*24 B of LDS must fit): instead of sorting, every scatter term is
// added to its target point as a 64-bit FIXED-POINT number (2^-44 resolution) with LDS integer
// atomics.  Integer addition commutes, so the result is independent of the arrival order --
// deterministic run to run like the sorted form -- and the sum is exact to 6e-14 absolute (tighter
// than the CPU op's sequential fp32 sum, from which it differs by normal fp32 rounding only).
// The bit-exact-vs-CPU sorted form stays behind the public NnDistanceGrad op (chamfer.hip).
constexpr double CG_FX = 17592186044416.0;          // 2^44
constexpr int CG_FX_MAX_N = 5000;                   // 3 * 8 B * n <= 120 KB: one workgroup holds the accumulators of a whole cloud
constexpr int CG_FX_MAX_N_PLANE = 15000;            // larger clouds: H = 2 or 3 workgroups per (cloud, problem), each owning a
                                                    // contiguous range of the receiving points (config 4: n = 8192, H = 2)
inline int cgrad_fx_parts(int n) { return (n + CG_FX_MAX_N - 1) / CG_FX_MAX_N; }
inline int cgrad_fx_range(int n) { return (n + cgrad_fx_parts(n) - 1) / cgrad_fx_parts(n); }
inline size_t cgrad_fx_lds_bytes(int n) { return sizeof(unsigned long long) * 3 * (size_t)cgrad_fx_range(n); }

// part h of H: this workgroup owns the receiving points [j_lo, j_hi); it scans ALL scatter sources and keeps those that land there
__device__ __forceinline__ void cgrad_fx_body(const CGradArgs &a, const int pi, const int b, const int h, const int H, unsigned *lds) {
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(lds);     // [n][3]
    const CGradProblem pr = a.pr[pi];
    const int n = a.n;
    const float wb = pr.w ? pr.w[b] : 1.0f;
    const float gd = wb * (1.0f / (float)n);
    const float g2 = gd * 2;
    const float *p = pr.p + (size_t)b * n * 3, *q = pr.q + (size_t)b * n * 3;
    const int *i1 = pr.idx1 + (size_t)b * n, *i2 = pr.idx2 + (size_t)b * n;
    const bool part = (pr.part_d || pr.part_w) && (!pr.part_need || sym_needed(pr.part_need, b));          // (uniform)
    const float *pd = pr.part_d + (size_t)b * pr.part_slices * n;
    const int *pi_ = pr.part_i + (size_t)b * pr.part_slices * n;
    const unsigned long long *pw = pr.part_w + (size_t)b * n;
    const int js = (pr.jstar && pr.extra_w > 0.f) ? pr.jstar[b] : -1;
    const int range = (n + H - 1) / H;
    const int j_lo = h * range, j_hi = min(n, j_lo + range);
    constexpr int U = 4;
    // row partials of the symmetric scan: this workgroup's own points' matches are folded FIRST (their loads run beside the
    // scatter phase below) when one pass covers them (n <= 2048 on 512 threads: always, where partials exist)
    const bool pre = part && (j_hi - j_lo) <= U * CGA_THREADS;
    int mpre[U];
    if (pre && pr.part_w) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j_lo + threadIdx.x + u * CGA_THREADS;
            mpre[u] = j < j_hi ? (int)(unsigned)pw[j] : 0;
        }
    } else if (pre) {
        int sl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j_lo + threadIdx.x + u * CGA_THREADS;
            sl[u] = 0;
            if (j < j_hi) (void)sym_merge_pick(pd + j, pr.part_slices, n, sl[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j_lo + threadIdx.x + u * CGA_THREADS;
            mpre[u] = j < j_hi ? pi_[(size_t)sl[u] * n + j] : 0;
        }
    }
    if (pre) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j_lo + threadIdx.x + u * CGA_THREADS;
            if (j < j_hi) pr.idx1_out[(size_t)b * n + j] = mpre[u];
        }
    }
    for (int e = threadIdx.x; e < 3 * (j_hi - j_lo); e += CGA_THREADS) acc[e] = 0ull;
    __syncthreads();
    LC_STAMP(1, 1);
    // Four points per thread and pass, index loads first, then all the dependent gathers: the launch is latency-bound
    // (one workgroup per cloud, problem and part), and a loop of "load index, gather, add" pays two global round trips per point.
    for (int k0 = threadIdx.x; k0 < n; k0 += U * CGA_THREADS) {
        int jj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) jj[u] = k0 + u * CGA_THREADS < n ? i2[k0 + u * CGA_THREADS] : -1;   // other point k matched our point j
        float qv[U][3], pv[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool mine = jj[u] >= j_lo && jj[u] < j_hi;
            const int k = mine ? k0 + u * CGA_THREADS : 0, j = mine ? jj[u] : 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) { qv[u][c] = q[3 * k + c]; pv[u][c] = p[3 * j + c]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (jj[u] < j_lo || jj[u] >= j_hi) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float t = g2 * (qv[u][c] - pv[u][c]);
                const long long f = __double2ll_rn((double)t * CG_FX);
                atomicAdd(&acc[3 * (jj[u] - j_lo) + c], (unsigned long long)f);
            }
        }
    }
    __syncthreads();
    LC_STAMP(1, 2);
    for (int j0 = j_lo + threadIdx.x; j0 < j_hi; j0 += U * CGA_THREADS) {
        int mj[U];
        if (pre) {
#pragma unroll
            for (int u = 0; u < U; ++u) mj[u] = mpre[u];
        } else if (part) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                mj[u] = 0;
                if (j0 + u * CGA_THREADS < j_hi) {
                    float d_;
                    if (pr.part_w) mj[u] = (int)(unsigned)pw[j0 + u * CGA_THREADS];
                    else sym_merge_slices(pd + j0 + u * CGA_THREADS, pi_ + j0 + u * CGA_THREADS, pr.part_slices, n, d_, mj[u]);
                    pr.idx1_out[(size_t)b * n + j0 + u * CGA_THREADS] = mj[u];
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) mj[u] = j0 + u * CGA_THREADS < j_hi ? i1[j0 + u * CGA_THREADS] : 0;
        }
        float qv[U][3], pv[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * CGA_THREADS < j_hi ? j0 + u * CGA_THREADS : j_lo;
#pragma unroll
            for (int c = 0; c < 3; ++c) { pv[u][c] = p[3 * j + c]; qv[u][c] = q[3 * mj[u] + c]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * CGA_THREADS;
            if (j >= j_hi) continue;
            const float gown = (j == js ? gd + wb * pr.extra_w : gd) * 2;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float own = gown * (pv[u][c] - qv[u][c]);
                const float sc = (float)((double)(long long)acc[3 * (j - j_lo) + c] * (1.0 / CG_FX));
                pr.g[((size_t)b * n + j) * 3 + c] = own - sc;
            }
        }
    }
}

// One workgroup (CGA_THREADS) of the fused pass for cloud b of B: row 0 = losses / metrics / keep-best (its four upper waves leave
// after folding the row partials), rows 1.. = gradients (problem (row - 1) / H, part (row - 1) % H).  dyn: the workgroup's dynamic
// LDS, loss_cgrad_lds_bytes(n) at least.
inline size_t loss_cgrad_lds_bytes(int n) { return std::max(cgrad_fx_lds_bytes(n), sizeof(float) * 2 * (size_t)LOSS_PRE_MAX_N); }
__device__ __forceinline__ void loss_cgrad_block(const LossArgs &la, const CGradArgs &ca, const int H, const int b, const int B, const int row, unsigned *dyn) {
    if (row == 0) {
        float *m1 = reinterpret_cast<float *>(dyn), *m3 = m1 + LOSS_PRE_MAX_N;
        if (la.part.deferred) {                            // (uniform) all eight waves fold the row partials, four sum them
            loss_premerge(la, b, B, CGA_THREADS, m1, m3);
            __syncthreads();
        }
        if (threadIdx.x < 256) loss_metrics_body(la, b, B, m1, m3);
    } else {
        LC_STAMP(1, 0);
        cgrad_fx_body(ca, (row - 1) / H, b, (row - 1) % H, H, dyn);
        LC_STAMP(1, 7);
    }
}

// The same workgroups as riders of another launch (chamfer_sym.hip): `rows` workgroups per cloud behind everything else in the grid,
// dealt so that cloud b's sit on XCD b % 8 -- where that cloud's scan and search workgroups ran, whose plain stores are then visible
// in the shared L2 (tools/handoff_probe.py: same-XCD hand-offs need no agent-scope data accesses; geoadv_attack verifies the
// block -> XCD dealing once per device before it asks for this).  A rider waits until done[b] has reached `target` (every producer
// of cloud b adds one on its way out, after draining its stores), bounded; then one L1 invalidate and the usual body.
struct LossRider {
    LossArgs la; CGradArgs ca;
    int H, rows, first_block, blocks, clouds;
    unsigned *done;            // [clouds] arrival counters (never reset: `target` is cumulative over the calls)
    unsigned target;
    int *spin_timeout;
    // host side only: the hosting launcher knows how the row minima will leave the scan (final / packed words / partials per slice)
    // only once it has chosen its launch shape; it hands that to the caller's `patch`, which completes la / ca before the launch
    void (*patch)(LossRider &, const SymPartials &, void *);
    void *ctx;
    bool force;                // host them wherever the launch CAN (the parity tests; the default hosts them where it pays)
};
inline int loss_rider_blocks(int clouds, int rows) { return 8 * ((clouds + 7) / 8) * rows; }
__device__ __forceinline__ bool loss_rider_block(const LossRider &r, unsigned *dyn) {
    if (r.blocks == 0 || (int)blockIdx.x < r.first_block || (int)blockIdx.x >= r.first_block + r.blocks) return false;
    const int l = blockIdx.x - r.first_block;
    const int idx = l >> 3, row = idx % r.rows, b = (idx / r.rows) * 8 + (l & 7);
    if (b >= r.clouds) return true;
#ifdef GA_STAMPS
#define LR_STAMP(I) do { if (threadIdx.x == 0 && l < GA_STAMP_BLOCKS) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); ga_stamps[((4 & 7) * GA_STAMP_BLOCKS + l) * 8 + (I)] = t_; } } while (0)
#else
#define LR_STAMP(I)
#endif
    LR_STAMP(0);
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(r.done + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - r.target) < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 24)) { *r.spin_timeout = 1; break; }
        }
    }
    __syncthreads();
    // this CU's vector L1 only (every wave; sc0): the producers ran on this XCD, their stores are in its L2.  (An agent-scope
    // acquire -- buffer_inv sc1 -- also drops L2 lines the scan workgroups still running beside us are using: measured + 7 us
    // per iteration at B = 32.)
    asm volatile("buffer_inv sc0\n\ts_waitcnt vmcnt(0)" ::: "memory");
    LR_STAMP(1);
    loss_cgrad_block(r.la, r.ca, r.H, b, r.clouds, row, dyn);
    LR_STAMP(7);
#undef LR_STAMP
    return true;
}
// a producer workgroup of cloud b on its way out: every wave drains its stores, then ONE lane counts the workgroup in
__device__ __forceinline__ void loss_rider_arrive(unsigned *done, int b) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(done + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace geoadv
