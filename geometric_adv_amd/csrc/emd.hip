// approx-EMD (ApproxMatch / MatchCost / MatchCostGrad) for gfx950.
//
// Replaces approxmatchLauncher / matchcostLauncher / matchcostgradLauncher
// (external/structural_losses/tf_approxmatch.cpp:141-143; kernels tf_approxmatch_g.cu).
// Parity target: the reference CPU op (tf_approxmatch.cpp:23-140): 11 levels j = 8..-2 with
// level = -4^j (0 at j = -2), double bookkeeping of the remaining capacities.
//
// The reference GPU kernel read-modify-writes the (b,m,n) match matrix once per level -- 10 x 2 x 4
// bytes per pair, the dominant HBM traffic.  Here the per-level weights are kept in
// FACTORISED form: with w_j(k,l) = exp(level_j * |p_k - q_l|^2),
//     weight_j(k,l) = w_j(k,l) * fL_j[k] * fR_j[l],
//     fL_j[k] = remL[k] / (1e-9 + sum_l w_j remR[l]),                       (pass A, one sum per k)
//     T_l = sum_k w_j fL_j[k];  r = min(remR[l] / (1e-9 + remR[l] T_l), 1);
//     fR_j[l] = remR[l] r;  remR[l] <- max(remR[l] - fR_j[l] T_l, 0)        (pass B, one sum per l)
//     remL[k] <- max(remL[k] - fL_j[k] sum_l w_j fR_j[l], 0)                (pass C, one sum per k; fused with pass A of
//                                                                            level j+1: one distance, two weights)
// which is the CPU loop (:36-78) with the row/column normalisations pulled out of the pair sums.
// Only 12 (n+m) doubles per cloud live in HBM during the levels; match is written ONCE at the end
// (sum over the 11 levels, accumulated level by level in float like the CPU's `match[k] += weight[k]`) -- and not at
// all inside the attack loop, which only needs the plan's cost and gradient (geoadv_emd_cost_grad1).
//
// Arithmetic (round 2; round 1 walked every pair in fp64 with the accurate expf at one wave per SIMD: 4.6 ms per
// approx_match at B = 32, N = 2048): the pair WEIGHT is fp32 -- d2 with FMAs, one v_exp_f32 of d2 * (level * log2 e) --
// and everything it is multiplied into stays fp64: factors, running sums (v_fma_f64), capacities, the 1e-9 guards.
// Why exactly this split.  The algorithm is not forgiving about its sums: a source next to an exhausted target has
// s = 1e-9 + w * (residual capacity ~1e-9..1e-7), so residual capacities compete with the 1e-9 guard, and they are
// themselves differences like rem - f * T with f * T = rem * (1 - 3e-7): a 6e-8 rounding of a factor, or an fp32 partial
// sum, turns into a 10 % error of such a residual and then into a 1e-3 error of a plan entry two levels later (measured on
// golden cloud a[1]; reproduced in numpy).  What it IS forgiving about is the weight itself, as long as passes A, B, C and
// the plan use the SAME value: w appears in numerator and denominator of every normalisation, so a relative error of w
// (here <= ~1e-6: fp32 distance, v_exp argument) perturbs the plan by the same order only.  Hence: one PairWeight::d2() for every
// kernel (bit-identical w everywhere), fp64 for the rest.  Tests hold match to rtol 2e-5 / atol 2e-6 of the reference CPU
// goldens (tests/test_gpu_emd.py); measured worst case 7e-7 absolute.
// On big clouds the fp32 weight shows: a handful of plan entries per million sit up to ~1e-4 relative from the CPU op (the
// statistical bound in tests/test_gpu_emd.py).  GEOADV_EMD_REFERENCE (the *_mode entry points) removes that: the weight's
// ARGUMENT is then the CPU op's own bits -- double coordinates, double distance without FMA, level * d2 in double, rounded
// to float where the CPU calls expf (:41-46) -- and the exponential is glibc's expf algorithm evaluated in fp64 (a 1.5-ulp
// exponential was measured NOT to be enough: one entry of 8.4 M at n = 4096 still moved by 1.1e-4 relative), so the weights
// are the CPU op's bit for bit, and the plan's level terms are formed and added in double like the CPU's `match += weight`.
// What remains is the order of the fp64 sums: plans within ~2 float ulps of the CPU op on EVERY entry at every size tested
// (2.0e-7 relative worst over 13 M entries; 40-80 % bit-equal).  3.7x the time of the fast mode (4.6 ms per approx_match
// at B = 32 x 2048^2, round 1's speed).  The attack loop defaults to the fast mode.
// Level j = -2 has level 0, i.e. w = 1 for every pair: its three sweeps are O(n + m) reductions, not O(n m) walks.
#include <atomic>
#include "common.h"
#include <math.h>

#pragma clang fp contract(off)

namespace geoadv {

constexpr int EMD_LEVELS = 11;                  // j = 8 .. -2 (tf_approxmatch.cpp:31)
constexpr int EMD_TILE = 1024;                  // "other" points staged per LDS tile (cost / gradient kernels)

static inline double emd_level(int li) {        // li = 0..10  <->  j = 8..-2
    const int j = 8 - li;
    return j == -2 ? 0.0 : -(double)powf(4.0f, (float)j);     // level = -powf(4.0, j) (:33-35)
}

// The pair weight exp(level * |p - o|^2) in its two modes.  C: coordinate / distance type; L: what a level is passed as.
template <bool REF> struct PairWeight;
template <> struct PairWeight<false> {                     // fast: fp32 distance with FMAs, one v_exp_f32 of d2 * (level * log2 e)
    using C = float;
    using L = float;
    using F = float;                                       // the plan's factors: fp32 products, `match += weight` by fmaf
    static L level(int li) { return (float)(emd_level(li) * 1.4426950408889634); }
    // the ONE form every kernel uses (sign-symmetric: the same bits whichever cloud is "own")
    static __device__ __forceinline__ float d2(float px, float py, float pz, float ox, float oy, float oz) {
        const float dx = px - ox, dy = py - oy, dz = pz - oz;
        return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    }
    static __device__ __forceinline__ float w(float d2, float c, const unsigned long long *) { return __builtin_amdgcn_exp2f(d2 * c); }
};
// expf as glibc >= 2.27 evaluates it (sysdeps/ieee754/flt-32/e_expf.c: the ARM optimized-routines algorithm): in double,
// x * 32 / ln 2 = k + r, 2^(k/32) from a 32-entry table, a cubic in r, one rounding to float.  Restated from the published
// algorithm with its constants; tests/test_expf_table.py checks the table against 2^(i/32) and, without a GPU, this very
// sequence of double operations bit for bit against the host libm (220 000 arguments incl. the denormal range).
// EXPF_TAB[i] = bits(2^(i/32)) - (i << 47), so that adding k << 47 yields bits(2^(k/32)).
__device__ const unsigned long long EXPF_TAB[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

template <> struct PairWeight<true> {                      // reference: the CPU op's argument bits and glibc's expf
    using C = double;
    using L = double;
    using F = double;                                      // the plan's level terms in double, added into the float like the CPU's (:75-76)
    static L level(int li) { return emd_level(li); }
    static __device__ __forceinline__ double d2(double px, double py, double pz, double ox, double oy, double oz) {
        const double dx = px - ox, dy = py - oy, dz = pz - oz;
        return (dx * dx + dy * dy) + dz * dz;              // (:46), no contraction in this file
    }
    static __device__ __forceinline__ float w(double d2, double level, const unsigned long long *tab) {
        const float a = (float)(level * d2);               // expf's argument, rounded like the CPU's call; a <= 0
        constexpr double INV_LN2_N = 0x1.71547652b82fep+0 * 32, SHIFT = 0x1.8p52;
        constexpr double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
        const double z = INV_LN2_N * (double)a;
        double kd = z + SHIFT;
        const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
        kd -= SHIFT;
        const double r = z - kd;
        const double s = __longlong_as_double((long long)(tab[ki & 31] + (ki << 47)));
        const double p = C0 * r + C1;
        const double r2 = r * r;
        double y = C2 * r + 1.0;
        y = p * r2 + y;
        y = y * s;
        return a < -0x1.9fe368p6f ? 0.f : (float)y;        // glibc's underflow cut; below it the table arithmetic is meaningless
    }
};

// temp layout per cloud (doubles): remL[n] remR[m] then per level: fL[n] fR[m]
__host__ __device__ inline size_t emd_temp_doubles_per_cloud(int n, int m) { return (size_t)(n + m) * (1 + EMD_LEVELS); }

__global__ void emd_init_kernel(int n, int m, double *temp) {
    const int c = blockIdx.y;
    double *t = temp + (size_t)c * emd_temp_doubles_per_cloud(n, m);
    const int big = n > m ? n : m;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) t[i] = (double)(big / n);                     // factorl = max(n,m)/n, integer division (:25)
    else if (i < n + m) t[i] = (double)(big / m);            // factorr (:26)
}

// ------------------------------------------------------------------------------------------
// Sparse levels (round 4).  exp(level * d2) is EXACTLY zero in float beyond a reach that is small at the first levels
// (level = -4^8, -4^7, -4^6: reach 0.040, 0.080, 0.160 -- glibc's expf returns 0 below -103.97, v_exp_f32 of an argument below
// -150 likewise), and a zero weight adds nothing to any of the fp64 sums.  The first sweeps (A and B of level 8, C8 + A7, B7;
// with SP_LEVELS = 3 also C7 + A6, B6) therefore only have to meet the pairs within reach: both clouds are binned per call on a
// common uniform grid per sparse level (cells no smaller than 1.01 x the reach, at most 16 per axis), in a stable order (cell,
// then point index: the sums stay reproducible run to run), and an own point walks the 3 x 3 x 3 cells around its own -- a dozen
// candidates at level 8, thirty at level 7 on 2048-point clouds in the unit cube, instead of all 2048.  Same pair arithmetic
// (PairWeight), same epilogues; the sums differ from the dense sweep's only in the order of their fp64 additions.  Where the
// cells would not thin the pairs out enough -- a box of a few cells, points piled into a few cells, clouds of very different
// extent -- a level keeps its dense sweeps: the choice is made per cloud pair and level on the device (SparseGrid::use,
// emd_sparse_bin_kernel), both kinds of workgroups are in every such launch.
// Measured (B = 32 / 128, N = 2048, uniform clouds; per launch under rocprofv3, profiles/r04_emd_trace_b32.txt / _b128.txt):
// A8 8.2 / 23 us, B8 9.2 / 25, C8 + A7 11.2 / 35, B7 10.4 / 29, C7 + A6 32.9 / 107, B6 24.5 / 79 against 46 / 160 (passes A, B) and
// 65 / 235 (C + A) dense, binning 32 / 57 us: approx_match 1.23 -> 1.08 ms and 4.40 -> 3.77 ms (reference weights: 4.21 -> 3.94 ms at
// B = 32).  A sparse pair costs 4 (long candidate lists) to 10 (short ones) dense pairs -- gathers, an fp64 fma chain per lane, the dense sweep's broadcast LDS reads
// gone -- hence the device-side test SP_COST_RATIO * (pairs met) <= n * m.  W is an average, though, and a sparse launch lasts as
// long as its longest candidate list: a cloud with a dense core (the attack's reconstruction of a random-init decoder: extent
// 0.15 inside a unit target cloud) puts thousands of candidates in front of the few points of the other cloud inside it.  So an
// own point whose 27 cells hold more than SP_HEAVY candidates is HEAVY: the binning kernel lists such points per level and cloud,
// the sparse form skips them, and the dense form's workgroups -- present in the launch anyway -- take the list instead of
// leaving: the same dense arithmetic against the whole other cloud for just those points, 128 per workgroup.
// On the attack's own pairs (a blob of extent 0.15 in a unit cube, r04_emd_trace_blob_b32.txt / _b128.txt) the sweeps whose own
// cloud is the wide one stay as above (A8 7.7 / 21 us, C8 + A7 11 / 33, C7 + A6 33 / 111) and the ones whose own cloud is the blob
// run at their heavy points' pace (B8 33 / 66 us, B7 33 / 66, B6 34 / 99 -- still below the dense 46 / 160), binning 53 / 91 us:
// configs[3]'s attack iteration 5.41 -> 4.89 ms at B = 128.
// Dead ends on the way, all measured: one lane per own point walking one candidate at a time (a chain of dependent L2 round
// trips: level 6 in 127 / 66 us at B = 32); four lanes and batches of loads but factors gathered by original index (91 / 59: the
// scattered 8-byte loads saturate the CU's address path); a loop per cell run instead of one flat candidate list (A8 78 us at
// B = 128: a run holds 1.5 points, the sweep was bound by its number of memory instructions); and the two forms' workgroups
// interleaved per cloud pair in dispatch order (a launch that stayed dense: 265 instead of 140 us -- see emd_sweep_kernel).
// ------------------------------------------------------------------------------------------
#ifndef SP_LEVELS_V
#define SP_LEVELS_V 3
#endif
constexpr int SP_LEVELS = SP_LEVELS_V;          // li = 0, 1 (, 2)  <->  j = 8, 7 (, 6)
constexpr int SP_MAX_G = 16, SP_MAX_CELLS = SP_MAX_G * SP_MAX_G * SP_MAX_G;
constexpr int SP_MAX_N = 8192;                  // points per cloud the binning kernel sorts in LDS (keys: 16 bits of index)
constexpr int SP_BIN_THREADS = 1024;

struct SparseGrid { float lo[3], ih[3]; int g[3], use; };
// per cloud pair, 16-byte aligned pieces: SparseGrid[SP_LEVELS]; then per level and cloud X in {1, 2}: float4 sorted[nX]
// (x, y, z, original index), int cell_start[SP_MAX_CELLS + 4], int inv[nX] (original index -> place in `sorted`), and two
// arrays of nX doubles: the factors the sparse sweeps of this level read for THIS cloud as the "other" one, in `sorted`'s
// order (cloud 1: fL; cloud 2: fR and remR) -- written by the epilogue of the sweep that produces them (through inv), so a
// candidate costs two sequential loads instead of a point and a dependent gather by original index; and int heavy[nX + 4]: the
// own points of this cloud whose candidate list would exceed SP_HEAVY (they take the dense form, see emd_sweep_kernel)
__host__ __device__ inline size_t sp_up(size_t v) { return (v + 15) & ~(size_t)15; }
__host__ __device__ inline size_t sp_grid_bytes() { return sp_up(sizeof(SparseGrid) * SP_LEVELS); }
__host__ __device__ inline size_t sp_cloud_bytes(int nx) {
    return sp_up(sizeof(float4) * (size_t)nx) + sp_up(sizeof(int) * (SP_MAX_CELLS + 4)) + sp_up(sizeof(int) * (size_t)nx) + 2 * sp_up(sizeof(double) * (size_t)nx) +
           sp_up(sizeof(int) * ((size_t)nx + 4));
}
__host__ __device__ inline size_t sp_bytes_per_pair(int n, int m) { return sp_grid_bytes() + SP_LEVELS * (sp_cloud_bytes(n) + sp_cloud_bytes(m)); }
struct SparseView { const SparseGrid *grid; const float4 *sorted[2]; const int *cell_start[2]; int *inv[2]; double *fac[2][2]; int *heavy[2]; };
__host__ __device__ inline SparseView sp_view(char *base, int n, int m, int level) {
    SparseView v;
    v.grid = reinterpret_cast<const SparseGrid *>(base) + level;
    char *p = base + sp_grid_bytes() + (size_t)level * (sp_cloud_bytes(n) + sp_cloud_bytes(m));
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        const int nx = w ? m : n;
        char *q = p;
        v.sorted[w] = reinterpret_cast<const float4 *>(q); q += sp_up(sizeof(float4) * (size_t)nx);
        v.cell_start[w] = reinterpret_cast<const int *>(q); q += sp_up(sizeof(int) * (SP_MAX_CELLS + 4));
        v.inv[w] = reinterpret_cast<int *>(q); q += sp_up(sizeof(int) * (size_t)nx);
        v.fac[w][0] = reinterpret_cast<double *>(q); q += sp_up(sizeof(double) * (size_t)nx);
        v.fac[w][1] = reinterpret_cast<double *>(q); q += sp_up(sizeof(double) * (size_t)nx);
        v.heavy[w] = reinterpret_cast<int *>(q);            // [0] = count, then the original indices of this cloud's HEAVY own points, ascending
        p += sp_cloud_bytes(nx);
    }
    return v;
}
__device__ __forceinline__ int sp_cell1(float v, float lo, float ih, int g) {
    const float t = fminf(fmaxf((v - lo) * ih, 0.f), (float)(g - 1));       // NaN -> 0; monotone in v
    return (int)t;
}

// grid = (SP_LEVELS, b): one workgroup bins BOTH clouds of a pair on the level's grid and decides whether the level's sparse
// sweeps pay: they meet W = sum over cells of (own points in the cell) x (other points in the 27 cells around it) pairs -- the
// same number whichever cloud is "own" -- at several times the dense sweep's cost per pair (gathers, four lanes per own point),
// so `use` needs SP_COST_RATIO * W' <= n * m, W' = W without the HEAVY own points' share plus n (m) dense pairs for each of
// those (they go to the dense form's workgroups); a level whose cells do not thin the pairs out keeps the dense sweeps.  LDS (dynamic): cntA, cntB, cursor [SP_MAX_CELLS + 4] ints, perm [SP_MAX_N].
#ifndef SP_COST_RATIO_V
#define SP_COST_RATIO_V 6
#endif
constexpr int SP_COST_RATIO = SP_COST_RATIO_V;
#ifndef SP_HEAVY_V
#define SP_HEAVY_V 384
#endif
constexpr int SP_HEAVY = SP_HEAVY_V;            // an own point whose 27 cells hold more candidates than this takes the DENSE form: W is an average, a cloud
                                                // with a dense core (the reconstruction of a random-init decoder is one) puts thousands of candidates in front of
                                                // the few points of the other cloud inside it, and a sparse launch lasts as long as its longest list
constexpr size_t SP_BIN_LDS = sizeof(int) * (5 * (SP_MAX_CELLS + 4) + SP_MAX_N);     // cntA, cntB, cursor, nbA, nbB, perm

// PER: points of a cloud per thread (n, m <= PER * SP_BIN_THREADS): every point is loaded ONCE, all requests of a thread in
// flight together, and stays in registers with its cell through the five passes (box, count, scatter, rank, write) -- the first
// form re-read the clouds in every pass, a dependent round trip each: 36 us per call at B = 32, all of it latency.
template <int PER>
__global__ __launch_bounds__(SP_BIN_THREADS) void emd_sparse_bin_kernel(int n, int m, const float *xyz1, const float *xyz2, char *sparse,
                                                                        float reach0) {
    extern __shared__ __attribute__((aligned(16))) int sp_lds[];
    int *cntA = sp_lds, *cntB = cntA + SP_MAX_CELLS + 4, *cur = cntB + SP_MAX_CELLS + 4, *nbA = cur + SP_MAX_CELLS + 4, *nbB = nbA + SP_MAX_CELLS + 4;
    int *perm = nbB + SP_MAX_CELLS + 4;
    __shared__ float red[6][SP_BIN_THREADS / 64];
    __shared__ int wsum[SP_BIN_THREADS / 64];
    __shared__ unsigned long long work, work2;
    __shared__ int hcount;
    __shared__ SparseGrid g;
    const int level = blockIdx.x, c = blockIdx.y, t = threadIdx.x;
    char *base = sparse + (size_t)c * sp_bytes_per_pair(n, m);
    const float *pc[2] = {xyz1 + (size_t)c * n * 3, xyz2 + (size_t)c * m * 3};
    const int nn[2] = {n, m};
    float px[2][PER], py[2][PER], pz[2][PER];
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = t + u * SP_BIN_THREADS;
            const bool in = i < nn[w];
            const float *q = pc[w] + 3 * (size_t)(in ? i : 0);
            px[w][u] = q[0]; py[w][u] = q[1]; pz[w][u] = q[2];
        }
    // the box of BOTH clouds
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (t + u * SP_BIN_THREADS < nn[w]) {
                lo[0] = fminf(lo[0], px[w][u]); hi[0] = fmaxf(hi[0], px[w][u]);
                lo[1] = fminf(lo[1], py[w][u]); hi[1] = fmaxf(hi[1], py[w][u]);
                lo[2] = fminf(lo[2], pz[w][u]); hi[2] = fmaxf(hi[2], pz[w][u]);
            }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], off)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off)); }
    if ((t & 63) == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[a][t >> 6] = lo[a]; red[3 + a][t >> 6] = hi[a]; }
    for (int i = t; i < 2 * (SP_MAX_CELLS + 4); i += SP_BIN_THREADS) cntA[i] = 0;       // (cntA and cntB are adjacent)
    if (t == 0) { work = 0ull; work2 = 0ull; }
    __syncthreads();
    if (t == 0) {
        const float reach = reach0 * (float)(1 << level) * 1.01f;       // the reach doubles from level to level (level = -4^j)
        for (int a = 0; a < 3; ++a) {
            float l = red[a][0], h = red[3 + a][0];
            for (int w = 1; w < SP_BIN_THREADS / 64; ++w) { l = fminf(l, red[a][w]); h = fmaxf(h, red[3 + a][w]); }
            const float ext = h - l;
            int ga = 1;
            if (ext > 0.f && ext < INFINITY) ga = (int)fminf(fmaxf(floorf(ext / reach), 1.f), (float)SP_MAX_G);
            g.g[a] = ga; g.lo[a] = (l > -INFINITY && l < INFINITY) ? l : 0.f;
            g.ih[a] = ga > 1 ? (float)ga / ext : 0.f;
        }
        g.use = 0;
    }
    __syncthreads();
    const int gx = g.g[0], gy = g.g[1], gz = g.g[2], cells = gx * gy * gz;
    int cid[2][PER];
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            cid[w][u] = (sp_cell1(pz[w][u], g.lo[2], g.ih[2], gz) * gy + sp_cell1(py[w][u], g.lo[1], g.ih[1], gy)) * gx +
                        sp_cell1(px[w][u], g.lo[0], g.ih[0], gx);
            if (t + u * SP_BIN_THREADS < nn[w]) atomicAdd(&(w ? cntB : cntA)[cid[w][u]], 1);
        }
    __syncthreads();
    {   // per cell: the points of either cloud in the 27 cells around it (= the candidate list of an own point of the OTHER cloud in
        // this cell); W = the pairs the sparse sweeps would meet if no point were heavy
        unsigned long long w = 0ull, w2 = 0ull;
        for (int cell = t; cell < cells; cell += SP_BIN_THREADS) {
            const int x = cell % gx, y = (cell / gx) % gy, z = cell / (gx * gy);
            int sa = 0, sb = 0;
            for (int zz = max(z - 1, 0); zz <= min(z + 1, gz - 1); ++zz)
                for (int yy = max(y - 1, 0); yy <= min(y + 1, gy - 1); ++yy)
                    for (int xx = max(x - 1, 0); xx <= min(x + 1, gx - 1); ++xx) {
                        sa += cntA[(zz * gy + yy) * gx + xx];
                        sb += cntB[(zz * gy + yy) * gx + xx];
                    }
            nbA[cell] = sa; nbB[cell] = sb;
            // the cost of this cell's own points in dense-pair units, for either cloud as "own": a sparse pair costs SP_COST_RATIO
            // dense ones, a heavy point a dense row (m or n pairs)
            w += (unsigned long long)cntA[cell] * (unsigned long long)(sb <= SP_HEAVY ? SP_COST_RATIO * sb : m);
            w2 += (unsigned long long)cntB[cell] * (unsigned long long)(sa <= SP_HEAVY ? SP_COST_RATIO * sa : n);
        }
        atomicAdd(&work, w);                                  // (integers: the sums do not depend on the order)
        atomicAdd(&work2, w2);
    }
    __syncthreads();
    if (t == 0) {
        g.use = (max(work, work2) <= (unsigned long long)n * (unsigned long long)m && max(n, m) <= SP_MAX_N) ? 1 : 0;
        reinterpret_cast<SparseGrid *>(base)[level] = g;
    }
    __syncthreads();
    if (!g.use) return;                                     // (uniform; nothing else of this level's data is read then)
    const SparseView v = sp_view(base, n, m, level);
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const int nx = nn[which];
        int *cnt = which ? cntB : cntA;
        float4 *sorted = const_cast<float4 *>(v.sorted[which]);
        int *cs = const_cast<int *>(v.cell_start[which]);
        {   // exclusive scan of cnt[0 .. SP_MAX_CELLS): 4 entries per thread
            const int lane = t & 63, wave = t >> 6;
            int vv[4], sm = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) { vv[u] = cnt[4 * t + u]; sm += vv[u]; }
            int inc = sm;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(inc, off);
                if (lane >= off) inc += o;
            }
            if (lane == 63) wsum[wave] = inc;
            __syncthreads();
            int before = inc - sm;
            for (int w = 0; w < wave; ++w) before += wsum[w];
#pragma unroll
            for (int u = 0; u < 4; ++u) { cnt[4 * t + u] = before; cur[4 * t + u] = before; before += vv[u]; }
        }
        __syncthreads();
        for (int i = t; i < cells; i += SP_BIN_THREADS) cs[i] = cnt[i];
        if (t == 0) cs[cells] = nx;
        int spos[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = t + u * SP_BIN_THREADS;
            spos[u] = 0;
            if (i < nx) { spos[u] = atomicAdd(&cur[cid[which][u]], 1); perm[spos[u]] = i; }      // (scheduling order inside a cell)
        }
        __syncthreads();
        // stable order: a point's place in its cell = the number of the cell's points with a smaller index (cur[cell] is now the
        // cell's end), so the order -- and with it the order of every fp64 sum of the sparse sweeps -- does not depend on
        // scheduling.  Cells fuller than SP_HEAVY keep the scatter's order: every list such a cell would be part of is longer than
        // SP_HEAVY, i.e. belongs to a heavy point, which does not read the sorted arrays at all.
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = t + u * SP_BIN_THREADS;
            if (i < nx) {
                const int cell = cid[which][u];
                const int b0 = cnt[cell], b1 = cur[cell];
                int place = spos[u];                         // (a full cell: the scatter's place)
                if (b1 - b0 <= SP_HEAVY) {
                    int r = 0;
                    for (int j = b0; j < b1; ++j) r += perm[j] < i ? 1 : 0;
                    place = b0 + r;
                }
                sorted[place] = make_float4(px[which][u], py[which][u], pz[which][u], __int_as_float(i));
                v.inv[which][i] = place;
                // pass A of the first level reads remR of cloud 2 before any sweep has written it: its initial value (:26), everywhere
                if (level == 0 && which == 1) v.fac[1][1][i] = (double)((n > m ? n : m) / m);
            }
        }
        // this cloud's heavy own points (candidate list = the OTHER cloud's points around the point's cell), ascending index
        {
            const int *nbo = which ? nbA : nbB;
            int *hl = v.heavy[which];
            if (t == 0) hcount = 0;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int i = t + u * SP_BIN_THREADS;
                const bool hv = i < nx && nbo[cid[which][u]] > SP_HEAVY;
                const unsigned long long bal = __ballot(hv);
                if ((t & 63) == 0) wsum[t >> 6] = __popcll(bal);
                __syncthreads();
                int before = hcount, total = 0;
                for (int w = 0; w < SP_BIN_THREADS / 64; ++w) { before += w < (t >> 6) ? wsum[w] : 0; total += wsum[w]; }
                if (hv) hl[1 + before + __popcll(bal & ((1ull << (t & 63)) - 1ull))] = i;
                __syncthreads();
                if (t == 0) hcount += total;
                __syncthreads();
            }
            if (t == 0) hl[0] = hcount;
        }
        __syncthreads();
    }
}

// `cons_level`: the level whose sparse sweep reads what this sweep's epilogue produces (-1: a dense one does)
struct SparseArgs { char *base; int level, dense_blocks, cons_level, sparse_blocks; };   // blocks: per cloud pair

// ------------------------------------------------------------------------------------------
// One level sweep.  A workgroup = 8 waves owns 128 "own" points (two per lane, the same in every wave); the "other" cloud
// is staged through LDS 1024 points at a time (coordinates fp32, factors fp64) and each wave walks one eighth of every tile, so a
// B = 32 x N = 2048 sweep is 512 workgroups = 4 waves per SIMD (round 1: thread per own point over the WHOLE other cloud,
// 1 wave per SIMD).  The eight partial sums of a point are folded in wave order -- a fixed order, so results are
// reproducible run to run.  PASS 0 = A, 1 = B, 2 = C, 3 = C of level li and A of level li + 1 in one walk (one distance,
// two weights: pass A of the next level reads nothing pass C writes for other points).
// ------------------------------------------------------------------------------------------
constexpr int SW_WAVES = 8, SW_THREADS = 64 * SW_WAVES, SW_TILE = 1024;
constexpr int SW_R = 2, SW_OWN = 64 * SW_R;   // own points per lane: 1 is 8 % slower (LDS reads per pair double), 4 the same (measured)

// what a sweep does with the pair sums of one own point (the CPU loop's row / column normalisations, :49-72)
// copies of an epilogue's results for the sparse sweep that reads them next, in that sweep's order of this cloud (inv == null: none)
struct EpiCopy { const int *inv; double *d0, *d1; };
template <int PASS>
__device__ __forceinline__ EpiCopy epi_copy(const SparseArgs &sp, char *base, int n, int m) {
    EpiCopy e{nullptr, nullptr, nullptr};
    if (!sp.base || sp.cons_level < 0 || PASS == 2) return e;
    if (!reinterpret_cast<const SparseGrid *>(base)[sp.cons_level].use) return e;
    const SparseView cv = sp_view(base, n, m, sp.cons_level);
    const int w = PASS == 1 ? 1 : 0;                         // the cloud this sweep owns: 2 in pass B, else 1
    e.inv = cv.inv[w]; e.d0 = cv.fac[w][0]; e.d1 = cv.fac[w][1];
    return e;
}
template <int PASS>
__device__ __forceinline__ void sweep_epilogue(int i, double tot0, double tot1, double *remL, double *remR, double *fL, double *fR, double *fL_next,
                                               const EpiCopy &ec) {
    if (PASS == 0) {
        const double f = remL[i] / (1e-9 + tot0);           // the CPU starts its row sum at 1e-9 (:49)
        fL[i] = f;
        if (ec.inv) ec.d0[ec.inv[i]] = f;
    } else if (PASS == 1) {
        const double rr = remR[i];
        const double ss = 1e-9 + rr * tot0;
        double r = rr / ss;
        r = r < 1.0 ? r : 1.0;
        const double f = rr * r;
        fR[i] = f;
        const double left = rr - f * tot0;
        const double rn = left > 0.0 ? left : 0.0;
        remR[i] = rn;
        if (ec.inv) { const int q = ec.inv[i]; ec.d0[q] = f; ec.d1[q] = rn; }
    } else {
        const double left = remL[i] - fL[i] * tot0;
        const double rl = left > 0.0 ? left : 0.0;
        remL[i] = rl;
        if (PASS == 3) {
            const double f = rl / (1e-9 + tot1);
            fL_next[i] = f;
            if (ec.inv) ec.d0[ec.inv[i]] = f;
        }
    }
}

// The sparse form of a sweep for one workgroup: SP_LANES adjacent lanes share one own point (taken in its cloud's cell order:
// neighbouring points sit in the same or adjacent cells and meet the same candidates) and walk the three-cell runs along x of
// the 3 x 3 rows around its cell, each lane every SP_LANES-th candidate, SP_BATCH of them per round: a candidate costs two
// DEPENDENT loads (the sorted point, then its factor by original index), so all of a round's points are requested before any of
// its factors, and those before any arithmetic (the first form -- one lane per point, one candidate at a time -- ran the level-6
// sweeps in 127 / 66 us against 65 / 45 dense: a chain of L2 round trips, one wave per SIMD at B = 32; with four lanes and
// batches but the factors still gathered by original index: 91 / 59 us, the scattered 8-byte loads saturate the CU's address
// path -- hence the factors in sorted order, SparseView::fac).  The lanes' sums are folded in a fixed order.
constexpr int SP_LANES = 4, SP_BATCH = 4;
template <int PASS, bool REF>
__device__ __forceinline__ void sweep_sparse_body(int n, int m, typename PairWeight<REF>::L c0, typename PairWeight<REF>::L c1,
                                                  const SparseView &sv, double *remL, double *remR, double *fL, double *fR, double *fL_next,
                                                  const EpiCopy &ec, const int block, const unsigned long long *etab) {
    using PW = PairWeight<REF>;
    using C = typename PW::C;
    constexpr int NF = PASS == 3 ? 2 : 1;
    const bool own_is_1 = PASS != 1;
    const int n_own = own_is_1 ? n : m;
    const int sub = threadIdx.x & (SP_LANES - 1);
    int s = block * (SW_THREADS / SP_LANES) + (threadIdx.x / SP_LANES);
    const bool live = s < n_own;
    s = live ? s : n_own - 1;                               // (a dead group computes a valid point and stores nothing: the fold below shuffles)
    const SparseGrid g = *sv.grid;
    const float4 P = sv.sorted[own_is_1 ? 0 : 1][s];
    const float4 *oth = sv.sorted[own_is_1 ? 1 : 0];
    const int *cs = sv.cell_start[own_is_1 ? 1 : 0];
    // the other cloud's factors in ITS sorted order: pass A reads remR of cloud 2, pass B fL of cloud 1, pass C + A fR and remR of cloud 2
    const double *sf0 = PASS == 0 ? sv.fac[1][1] : (PASS == 1 ? sv.fac[0][0] : sv.fac[1][0]);
    const double *sf1 = sv.fac[1][1];
    const int gx = g.g[0], gy = g.g[1], gz = g.g[2];
    const int cx = sp_cell1(P.x, g.lo[0], g.ih[0], gx), cy = sp_cell1(P.y, g.lo[1], g.ih[1], gy), cz = sp_cell1(P.z, g.lo[2], g.ih[2], gz);
    const C px = (C)P.x, py = (C)P.y, pz = (C)P.z;
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, gx - 1);
    // the bounds of the (up to) nine rows first: one round trip for all of them
    int rlo[9], rhi[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int z = cz + r / 3 - 1, y = cy + r % 3 - 1;
        const bool in = z >= 0 && z < gz && y >= 0 && y < gy;
        const int row = in ? (z * gy + y) * gx : 0;
        rlo[r] = in ? cs[row + x0] : 0;
        rhi[r] = in ? cs[row + x1 + 1] : 0;
    }
    // the nine runs as ONE list of T candidates (a loop per run would issue its loads for one or two candidates at a time: at
    // the first levels a run holds 1.5 points on average, and the sweep was bound by the number of memory instructions):
    // candidate k sits at sorted position k + off[r] for the run r with pre[r] <= k < pre[r + 1]
    int pre[10], off[9];
    pre[0] = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) { off[r] = rlo[r] - pre[r]; pre[r + 1] = pre[r] + (rhi[r] - rlo[r]); }
    const int T = pre[9];
    if (T > SP_HEAVY) return;                               // a heavy point: the dense form's workgroups take it (the same count the binning kernel saw;
                                                            // uniform in the four lanes of the point, and nothing below crosses points)
    double acc0 = 0.0, acc1 = 0.0;
    for (int k0 = sub; k0 < T; k0 += SP_LANES * SP_BATCH) {
        float4 O[SP_BATCH];
        double f0[SP_BATCH], f1[SP_BATCH];
        bool ok[SP_BATCH];
#pragma unroll
        for (int u = 0; u < SP_BATCH; ++u) {
            const int k = k0 + SP_LANES * u;
            ok[u] = k < T;
            const int kk = ok[u] ? k : 0;
            int o = off[0];
#pragma unroll
            for (int r = 1; r < 9; ++r) o = kk >= pre[r] ? off[r] : o;
            const int ee = T > 0 ? kk + o : 0;
            O[u] = oth[ee];
            f0[u] = sf0[ee];
            f1[u] = NF == 2 ? sf1[ee] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < SP_BATCH; ++u) {
            const C d2 = PW::d2(px, py, pz, (C)O[u].x, (C)O[u].y, (C)O[u].z);
            acc0 = fma(ok[u] ? (double)PW::w(d2, c0, etab) : 0.0, f0[u], acc0);
            if (NF == 2) acc1 = fma(ok[u] ? (double)PW::w(d2, c1, etab) : 0.0, f1[u], acc1);
        }
    }
#pragma unroll
    for (int off = 1; off < SP_LANES; off <<= 1) {          // (a + b is the same double on both sides: every lane ends with the same total)
        acc0 += __shfl_xor(acc0, off);
        if (NF == 2) acc1 += __shfl_xor(acc1, off);
    }
    if (live && sub == 0) sweep_epilogue<PASS>(__float_as_int(P.w), acc0, acc1, remL, remR, fL, fR, fL_next, ec);
}

template <int PASS, bool REF>
__global__ __launch_bounds__(SW_THREADS, 2) void emd_sweep_kernel(int n, int m, int li, typename PairWeight<REF>::L c0,
                                                                 typename PairWeight<REF>::L c1, const float *xyz1,
                                                                 const float *xyz2, double *temp, SparseArgs sp) {
    using PW = PairWeight<REF>;
    using C = typename PW::C;
    struct alignas(16) Pt { C x, y, z, pad; };
    constexpr int NF = PASS == 3 ? 2 : 1;
    __shared__ Pt st[SW_TILE];                              // x, y, z of the other cloud's points
    __shared__ double sf[NF][SW_TILE];                      // their factors, fp64 (see the note on consistency above)
    __shared__ double part[NF][SW_WAVES][SW_OWN];
    __shared__ unsigned long long etab[REF ? 32 : 1];
    if (REF && threadIdx.x < 32) etab[threadIdx.x] = EXPF_TAB[threadIdx.x];       // (the tile loop's barrier orders it)
    // 1-D grid: the sparse form's workgroups of ALL cloud pairs first, then the dense form's -- whichever kind a pair does not use
    // leaves at once, and such workgroups must not sit BETWEEN working ones in dispatch order: with (kind, pair) interleaved per
    // pair a launch that stayed dense took 265 instead of 140 us at B = 128 (2048 leaving workgroups among 2048 working ones)
    const int n_sparse_all = sp.sparse_blocks * (int)gridDim.y;          // (gridDim.y = batch; gridDim.x = blocks per pair of both kinds)
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const bool sparse_wg = lin < n_sparse_all;
    // (the dense form's workgroups block-major -- block 0 of every pair, then block 1 of every pair ... --: where a level is sparse
    // they only take the pair's heavy points, a few blocks per pair, and the ones that leave must again not sit between the ones
    // that work: pair-major, pass B of a blob / uniform pair took 150 us for ONE working block per pair)
    const int c = sparse_wg ? lin / max(sp.sparse_blocks, 1) : (lin - n_sparse_all) % (int)gridDim.y;
    const int blk = sparse_wg ? lin % max(sp.sparse_blocks, 1) : (lin - n_sparse_all) / (int)gridDim.y;
    double *t = temp + (size_t)c * emd_temp_doubles_per_cloud(n, m);
    double *remL = t, *remR = t + n, *fL = t + (size_t)(n + m) * (1 + li), *fR = fL + n, *fL_next = fL + (n + m);
    const bool own_is_1 = PASS != 1;
    const int n_own = own_is_1 ? n : m, n_oth = own_is_1 ? m : n;
    const float *own = (own_is_1 ? xyz1 : xyz2) + (size_t)c * n_own * 3;
    const float *oth = (own_is_1 ? xyz2 : xyz1) + (size_t)c * n_oth * 3;
    const double *fac0 = PASS == 0 ? remR : (PASS == 1 ? fL : fR);      // PASS 3: C's factor fR_li ...
    const double *fac1 = remR;                                           // ... and A's factor remR
    EpiCopy ec{nullptr, nullptr, nullptr};
    const int *hlist = nullptr;                             // non-null: own points = hlist[1 .. hcnt] instead of blk * SW_OWN + ...
    int hcnt = 0;
    if (sp.base) {   // a launch with both forms: this cloud pair's grid says which workgroups work (uniform per workgroup)
        char *base = sp.base + (size_t)c * sp_bytes_per_pair(n, m);
        ec = epi_copy<PASS>(sp, base, n, m);
        const int use = sp.level >= 0 ? reinterpret_cast<const SparseGrid *>(base)[sp.level].use : 0;
        if (sparse_wg) {
            if (!use) return;
            if (REF) __syncthreads();                       // etab
            sweep_sparse_body<PASS, REF>(n, m, c0, c1, sp_view(base, n, m, sp.level), remL, remR, fL, fR, fL_next, ec, blk, etab);
            return;
        }
        if (use) {   // this level is sparse for the pair: the dense form only takes the HEAVY own points (the binning kernel's list)
            hlist = sp_view(base, n, m, sp.level).heavy[own_is_1 ? 0 : 1];
            hcnt = hlist[0];
            if (blk * SW_OWN >= hcnt) return;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    C px[SW_R], py[SW_R], pz[SW_R];
#pragma unroll
    for (int r = 0; r < SW_R; ++r) {
        int i = blk * SW_OWN + r * 64 + lane;
        if (hlist) i = hlist[1 + (i < hcnt ? i : hcnt - 1)];
        i = i < n_own ? i : n_own - 1;                       // (clamped lanes compute a valid point and are not stored)
        px[r] = own[3 * i]; py[r] = own[3 * i + 1]; pz[r] = own[3 * i + 2];
    }
    double acc[NF][SW_R] = {};
    // A tile of the other cloud travels global -> registers -> LDS, and the NEXT tile's loads are issued before the walk over the
    // current one (the walk reads LDS only, so they land while it runs): at m = 2048 the second tile's round trip (~2 us of a
    // ~40 us workgroup) is no longer exposed.
    constexpr int SW_PER = SW_TILE / SW_THREADS;            // points per thread and tile
    float qx[SW_PER], qy[SW_PER], qz[SW_PER];
    double qf0[SW_PER], qf1[SW_PER];
    auto request = [&](int t0) {
        const int cnt = min(SW_TILE, n_oth - t0);
#pragma unroll
        for (int k = 0; k < SW_PER; ++k) {
            int e = threadIdx.x + k * SW_THREADS;
            e = e < cnt ? e : cnt - 1;                       // (clamped: loaded, never stored)
            const float *q = oth + 3 * (size_t)(t0 + e);
            qx[k] = q[0]; qy[k] = q[1]; qz[k] = q[2];
            qf0[k] = fac0[t0 + e];
            qf1[k] = NF == 2 ? fac1[t0 + e] : 0.0;
        }
    };
    request(0);
    for (int t0 = 0; t0 < n_oth; t0 += SW_TILE) {
        const int cnt = min(SW_TILE, n_oth - t0);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SW_PER; ++k) {
            const int e = threadIdx.x + k * SW_THREADS;
            if (e < cnt) {
                st[e] = Pt{(C)qx[k], (C)qy[k], (C)qz[k], (C)0};
                sf[0][e] = qf0[k];
                if (NF == 2) sf[1][e] = qf1[k];
            }
        }
        __syncthreads();
        if (t0 + SW_TILE < n_oth) request(t0 + SW_TILE);
        const int per = (cnt + SW_WAVES - 1) / SW_WAVES;
        const int lo = wave * per, hi = min(cnt, lo + per);
#pragma unroll 4
        for (int e = lo; e < hi; ++e) {
            const Pt o = st[e];
            const double f0 = sf[0][e];
            const double f1 = NF == 2 ? sf[1][e] : 0.0;
#pragma unroll
            for (int r = 0; r < SW_R; ++r) {
                const C d2 = PW::d2(px[r], py[r], pz[r], o.x, o.y, o.z);
                acc[0][r] = fma((double)PW::w(d2, c0, etab), f0, acc[0][r]);
                if (NF == 2) acc[1][r] = fma((double)PW::w(d2, c1, etab), f1, acc[1][r]);
            }
        }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < SW_R; ++r) part[f][wave][r * 64 + lane] = acc[f][r];
    __syncthreads();
    if (threadIdx.x >= SW_OWN) return;
    int i = blk * SW_OWN + threadIdx.x;
    if (hlist) {
        if (i >= hcnt) return;
        i = hlist[1 + i];
    }
    if (i >= n_own) return;
    double tot[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        double v = part[f][0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < SW_WAVES; ++w) v += part[f][w][threadIdx.x];
        tot[f] = v;
    }
    sweep_epilogue<PASS>(i, tot[0], tot[NF - 1], remL, remR, fL, fR, fL_next, ec);
}

// Level j = -2: level = 0, every pair weight is expf(0) = 1, so the row sums are the same for every point:
//   A: fL[k] = remL[k] / (1e-9 + sum_l remR[l]);  B: T = sum_k fL[k], fR[l] = remR[l] * min(remR[l] / (1e-9 + remR[l] T), 1).
// (The capacities left after this last level are never read.)  One workgroup per cloud.
__device__ __forceinline__ double block_sum_1024(double v, double *red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
}

__global__ __launch_bounds__(1024) void emd_level0_kernel(int n, int m, int li, double *temp) {
    __shared__ double red[16];
    const int c = blockIdx.x;
    double *t = temp + (size_t)c * emd_temp_doubles_per_cloud(n, m);
    double *remL = t, *remR = t + n, *fL = t + (size_t)(n + m) * (1 + li), *fR = fL + n;
    double a = 0.0;
    for (int l = threadIdx.x; l < m; l += blockDim.x) a += remR[l];
    const double s = 1e-9 + block_sum_1024(a, red);
    double b = 0.0;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const double f = remL[k] / s;
        fL[k] = f;
        b += f;
    }
    const double T = block_sum_1024(b, red);
    for (int l = threadIdx.x; l < m; l += blockDim.x) {
        const double rr = remR[l];
        double r = rr / (1e-9 + rr * T);
        r = r < 1.0 ? r : 1.0;
        fR[l] = rr * r;
    }
}

template <bool REF> struct EmdLevels { typename PairWeight<REF>::L c[EMD_LEVELS]; };      // fast: level * log2(e); c[10] = 0

// match value of one pair from its squared distance: sum over the levels of exp2(c_j d2) * fR_j[l] * fL_j[k], accumulated
// level by level in float like the CPU's `match[k] += weight[k]` (:75-76).  fr: LDS row of the 11 column factors.
template <bool REF>
__device__ __forceinline__ float plan_value(const EmdLevels<REF> &lv, typename PairWeight<REF>::C d2,
                                            const typename PairWeight<REF>::F (&fl)[EMD_LEVELS], const typename PairWeight<REF>::F *fr,
                                            int stride, const unsigned long long *etab) {
    float mf = 0.f;
    if (REF) {
#pragma unroll
        for (int j = 0; j < EMD_LEVELS - 1; ++j)
            mf = (float)((double)mf + ((double)PairWeight<REF>::w(d2, lv.c[j], etab) * fl[j]) * fr[j * stride]);
        return (float)((double)mf + (double)fl[EMD_LEVELS - 1] * fr[(EMD_LEVELS - 1) * stride]);
    }
#pragma unroll
    for (int j = 0; j < EMD_LEVELS - 1; ++j) mf = fmaf(PairWeight<REF>::w(d2, lv.c[j], etab) * fr[j * stride], fl[j], mf);
    return fmaf(fr[(EMD_LEVELS - 1) * stride], fl[EMD_LEVELS - 1], mf);      // level 0: w = 1
}

// match[c][l][k] for the public op.  grid = (n/256, m/32, b): thread = one k, 32 l's.
constexpr int EMD_LT = 32;
template <bool REF>
__global__ __launch_bounds__(256) void emd_match_kernel(int n, int m, EmdLevels<REF> lv, const float *xyz1, const float *xyz2,
                                                        const double *temp, float *match) {
    using PW = PairWeight<REF>;
    using C = typename PW::C;
    using F = typename PW::F;
    __shared__ C qx[EMD_LT], qy[EMD_LT], qz[EMD_LT];
    __shared__ F fr[EMD_LEVELS][EMD_LT];
    __shared__ unsigned long long etab[REF ? 32 : 1];
    if (REF && threadIdx.x < 32) etab[threadIdx.x] = EXPF_TAB[threadIdx.x];
    const int c = blockIdx.z;
    const double *t = temp + (size_t)c * emd_temp_doubles_per_cloud(n, m);
    const int l0 = blockIdx.y * EMD_LT;
    const int lcnt = min(EMD_LT, m - l0);
    for (int e = threadIdx.x; e < lcnt * (3 + EMD_LEVELS); e += 256) {
        const int l = e % lcnt, what = e / lcnt;
        if (what < 3) (what == 0 ? qx : what == 1 ? qy : qz)[l] = xyz2[((size_t)c * m + l0 + l) * 3 + what];
        else fr[what - 3][l] = (F)t[(size_t)(n + m) * (1 + (what - 3)) + n + l0 + l];
    }
    __syncthreads();
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const float *p = xyz1 + ((size_t)c * n + k) * 3;
    const C px = p[0], py = p[1], pz = p[2];
    F fl[EMD_LEVELS];
#pragma unroll
    for (int j = 0; j < EMD_LEVELS; ++j) fl[j] = (F)t[(size_t)(n + m) * (1 + j) + k];
    for (int l = 0; l < lcnt; ++l) {
        match[((size_t)c * m + l0 + l) * n + k] = plan_value<REF>(lv, PW::d2(px, py, pz, qx[l], qy[l], qz[l]), fl, &fr[0][l], EMD_LT, etab);
    }
}

// The attack loop's use of the plan (adv_ae.py:120-124 with the build-defined EMD term): match_cost and
// match_cost_grad w.r.t. xyz1 only, with the plan treated as a constant (ApproxMatch is NoGradient, tf_approxmatch.py:19).
// Both are sums over pairs of match[l][k] times a function of the pair, so the plan is formed pair by pair in registers
// and never written: 4 n m bytes per cloud (537 MB at B = 32) neither stored nor re-read twice.  Same pair arithmetic as
// matchcost_cpu / matchcostgrad_cpu (:85-133): float distance, sqrtf, max(d, 1e-20).  Workgroup = 8 waves x 64 points k;
// each wave walks one eighth of the other cloud; partials folded in wave order.
template <bool REF> constexpr int PL_TILE = REF ? 512 : 1024;      // (double factors: the same LDS either way)
template <bool REF>
__global__ __launch_bounds__(SW_THREADS, 2) void emd_plan_cost_grad1_kernel(int n, int m, EmdLevels<REF> lv, const float *xyz1, const float *xyz2,
                                                                           const double *temp, double *cost_partial, float *grad1) {
    using PW = PairWeight<REF>;
    using F = typename PW::F;
    constexpr int TILE = PL_TILE<REF>;
    __shared__ float qx[TILE], qy[TILE], qz[TILE];
    __shared__ F fr[EMD_LEVELS][TILE];
    __shared__ float gpart[SW_WAVES][3][64];
    __shared__ double cpart[SW_WAVES];
    __shared__ unsigned long long etab[REF ? 32 : 1];
    if (REF && threadIdx.x < 32) etab[threadIdx.x] = EXPF_TAB[threadIdx.x];
    const int c = blockIdx.y;
    const double *t = temp + (size_t)c * emd_temp_doubles_per_cloud(n, m);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int k = blockIdx.x * 64 + lane;
    const bool live = k < n;
    k = live ? k : n - 1;
    const float *p = xyz1 + ((size_t)c * n + k) * 3;
    const float px = p[0], py = p[1], pz = p[2];
    F fl[EMD_LEVELS];
#pragma unroll
    for (int j = 0; j < EMD_LEVELS; ++j) fl[j] = (F)t[(size_t)(n + m) * (1 + j) + k];
    float gx = 0.f, gy = 0.f, gz = 0.f;
    double cost = 0.0;
    // A tile of the other cloud (coordinates + its eleven factors per point) travels global -> registers -> LDS, and the NEXT
    // tile's loads are issued before the walk over the current one, which reads LDS only (as in the level sweeps above).
    constexpr int PER = TILE / SW_THREADS;                  // points per thread and tile
    float rqx[PER], rqy[PER], rqz[PER];
    F rfr[PER][EMD_LEVELS];
    auto request = [&](int t0) {
        const int cnt = min(TILE, m - t0);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            int e = threadIdx.x + u * SW_THREADS;
            e = e < cnt ? e : cnt - 1;                       // (clamped: loaded, never stored)
            const float *q = xyz2 + ((size_t)c * m + t0 + e) * 3;
            rqx[u] = q[0]; rqy[u] = q[1]; rqz[u] = q[2];
#pragma unroll
            for (int j = 0; j < EMD_LEVELS; ++j) rfr[u][j] = (F)t[(size_t)(n + m) * (1 + j) + n + t0 + e];
        }
    };
    request(0);
    for (int t0 = 0; t0 < m; t0 += TILE) {
        const int cnt = min(TILE, m - t0);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = threadIdx.x + u * SW_THREADS;
            if (e < cnt) {
                qx[e] = rqx[u]; qy[e] = rqy[u]; qz[e] = rqz[u];
#pragma unroll
                for (int j = 0; j < EMD_LEVELS; ++j) fr[j][e] = rfr[u][j];
            }
        }
        __syncthreads();
        if (t0 + TILE < m) request(t0 + TILE);
        const int per = (cnt + SW_WAVES - 1) / SW_WAVES;
        const int lo = wave * per, hi = min(cnt, lo + per);  // (wave: an SGPR, so the walk is a scalar loop)
        float cs = 0.f;
        if constexpr (REF) {
            for (int l = lo; l < hi; ++l) {
                const float ox = qx[l] - px, oy = qy[l] - py, oz = qz[l] - pz;          // q - p, like the CPU loops
                const float w = plan_value<REF>(lv, PW::d2(px, py, pz, qx[l], qy[l], qz[l]), fl, &fr[0][l], TILE, etab);
                const float d = sqrtf(ox * ox + oy * oy + oz * oz);                     // matchcost_cpu's own float distance (:93-96)
                cs += d * w;                                                            // (:97-99) float product, summed below in double
                const float inv = 1.0f / fmaxf(d, 1e-20f);
                gx = fmaf(-w, ox * inv, gx); gy = fmaf(-w, oy * inv, gy); gz = fmaf(-w, oz * inv, gz);
            }
        } else {
            // The fast mode's walk is bound by VALU issue, and half of what it issued was not the plan: correctly rounded sqrtf and
            // 1 / d (8 + 10 instructions), a second distance beside the weight's, a vector loop counter.  Here one distance serves
            // both (q - p and p - q have the same squares: d2 is PairWeight::d2's bits), d = d2 * rsq(d2) and 1 / d = rsq(d2) (1 ulp
            // each), products fused.  Same sums up to rounding: the fused op is held to the three ops by tolerance, not bit for bit.
            // v_rsq_f32 is no good below ~1e-38 (coincident points: 0 * inf; denormal d2; the clamp of d at 1e-20): a wave that
            // met such a pair in a tile walks the tile again with the exact forms -- one v_cmp per pair buys that.
            const float gx0 = gx, gy0 = gy, gz0 = gz;
            unsigned long long tiny = 0ull;
#pragma unroll 2
            for (int l = lo; l < hi; ++l) {
                const float ox = qx[l] - px, oy = qy[l] - py, oz = qz[l] - pz;
                const float d2 = fmaf(oz, oz, fmaf(oy, oy, ox * ox));
                const float w = plan_value<REF>(lv, d2, fl, &fr[0][l], TILE, etab);
                tiny |= __builtin_amdgcn_ballot_w64(d2 < 1e-30f);
                const float inv = __builtin_amdgcn_rsqf(d2);
                cs = fmaf(d2 * inv, w, cs);
                const float wi = w * inv;
                gx = fmaf(-wi, ox, gx); gy = fmaf(-wi, oy, gy); gz = fmaf(-wi, oz, gz);
            }
            if (__builtin_expect(tiny != 0ull, 0)) {
                cs = 0.f; gx = gx0; gy = gy0; gz = gz0;
                for (int l = lo; l < hi; ++l) {
                    const float ox = qx[l] - px, oy = qy[l] - py, oz = qz[l] - pz;
                    const float d2 = fmaf(oz, oz, fmaf(oy, oy, ox * ox));
                    const float w = plan_value<REF>(lv, d2, fl, &fr[0][l], TILE, etab);
                    const float d = sqrtf(d2);
                    const float wi = w * (1.0f / fmaxf(d, 1e-20f));
                    cs = fmaf(d, w, cs);
                    gx = fmaf(-wi, ox, gx); gy = fmaf(-wi, oy, gy); gz = fmaf(-wi, oz, gz);
                }
            }
        }
        cost += (double)cs;
    }
    if (!live) cost = 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cost += __shfl_xor(cost, off);
    gpart[wave][0][lane] = gx; gpart[wave][1][lane] = gy; gpart[wave][2][lane] = gz;
    if (lane == 0) cpart[wave] = cost;
    __syncthreads();
    if (threadIdx.x < 192) {
        const int a = threadIdx.x / 64, ln = threadIdx.x % 64;
        float g = gpart[0][a][ln];
#pragma unroll
        for (int w = 1; w < SW_WAVES; ++w) g += gpart[w][a][ln];
        const int kk = blockIdx.x * 64 + ln;
        if (kk < n) grad1[((size_t)c * n + kk) * 3 + a] = g;
    }
    if (threadIdx.x == 0) {
        double s = cpart[0];
#pragma unroll
        for (int w = 1; w < SW_WAVES; ++w) s += cpart[w];
        cost_partial[(size_t)c * gridDim.x + blockIdx.x] = s;
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

// scratch of the sparse levels behind the doubles (0 where the clouds are too large for the binning kernel's LDS sort)
static size_t emd_sparse_floats(int b, int n, int m) { return std::max(n, m) <= SP_MAX_N ? (size_t)b * sp_bytes_per_pair(n, m) / 4 + 8 : 0; }

extern "C" size_t geoadv_approx_match_temp_floats(int b, int n, int m) {
    if (b <= 0 || n + m <= 0) return 16;
    return 2 * (size_t)b * emd_temp_doubles_per_cloud(n, m) + emd_sparse_floats(b, n, m) + 16;
}

static int emd_check(const char *op, int b, int n, int m) {
    GA_REQUIRE(b >= 0 && n >= 1 && m >= 1, "%s: needs b >= 0 and at least one point per cloud (b=%d n=%d m=%d)", op, b, n, m);
    GA_REQUIRE(b <= 65535, "%s: batch %d exceeds 65535", op, b);
    GA_REQUIRE((size_t)n * m <= ((size_t)1 << 31), "%s: n*m too large", op);
    return GEOADV_OK;
}

// the eleven levels: capacities and factors into temp (fp64), no plan yet
static std::atomic<int> g_emd_sparse{1};   // geoadv_emd_sparse_levels: the process default of calls whose mode carries no GEOADV_EMD_DENSE_LEVELS
                                           // flag (0 = every sweep dense); tests and measurements only -- per call, use the flag

// `sparse`: scratch of emd_sparse_floats(b, n, m) floats, 16-byte aligned, or null (clouds too large / switched off)
template <bool REF>
static int emd_run_levels(int b, int n, int m, const float *xyz1, const float *xyz2, double *t, char *sparse, EmdLevels<REF> &lv,
                          bool dense_levels, hipStream_t st) {
    emd_init_kernel<<<dim3(cdiv(n + m, 256), b), 256, 0, st>>>(n, m, t);
    GA_LAUNCH_CHECK();
    for (int li = 0; li < EMD_LEVELS; ++li) lv.c[li] = PairWeight<REF>::level(li);
    if (dense_levels || !g_emd_sparse.load()) sparse = nullptr;
    if (sparse) {
        // reach of level 8: beyond it both weight forms are exactly 0 -- glibc's expf below -103.97 (REF: float(level * d2)), v_exp_f32
        // of an argument below -150 (fast: d2 * level * log2 e) -- i.e. level * d2 <= -104.7 covers both; the reach doubles per level
        const float reach0 = sqrtf(104.7f / 65536.0f);
        static DeviceOnce attr;
        if (int rc = attr.run([]() -> int {
                GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(emd_sparse_bin_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SP_BIN_LDS));
                GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(emd_sparse_bin_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SP_BIN_LDS));
                GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(emd_sparse_bin_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SP_BIN_LDS));
                return GEOADV_OK;
            })) return rc;
        const dim3 bg(SP_LEVELS, b);
        const int big = std::max(n, m);
        if (big <= 2 * SP_BIN_THREADS) emd_sparse_bin_kernel<2><<<bg, SP_BIN_THREADS, SP_BIN_LDS, st>>>(n, m, xyz1, xyz2, sparse, reach0);
        else if (big <= 4 * SP_BIN_THREADS) emd_sparse_bin_kernel<4><<<bg, SP_BIN_THREADS, SP_BIN_LDS, st>>>(n, m, xyz1, xyz2, sparse, reach0);
        else emd_sparse_bin_kernel<8><<<bg, SP_BIN_THREADS, SP_BIN_LDS, st>>>(n, m, xyz1, xyz2, sparse, reach0);
        GA_LAUNCH_CHECK();
    }
    const int d1 = cdiv(n, SW_OWN), d2 = cdiv(m, SW_OWN);
    // a sweep whose larger level is a sparse one gets the sparse workgroups behind the dense ones (one of the two kinds leaves at once)
    // (blocks per pair of both kinds in x, pairs in y: the kernel linearises them itself, sparse form first)
    auto nsp = [&](int n_own, int level) { return sparse && level < SP_LEVELS ? cdiv(n_own, SW_THREADS / SP_LANES) : 0; };
    auto grid = [&](int dense, int n_own, int level) { return dim3(dense + nsp(n_own, level), b); };
    // (own level: the grid this sweep walks in its sparse form; consumer level: the sparse sweep that reads what its epilogue writes)
    auto spa = [&](int dense, int n_own, int level, int cons) {
        const bool any = sparse && (level < SP_LEVELS || cons < SP_LEVELS);
        return SparseArgs{any ? sparse : nullptr, level < SP_LEVELS ? level : -1, dense, cons < SP_LEVELS ? cons : -1, nsp(n_own, level)};
    };
    emd_sweep_kernel<0, REF><<<grid(d1, n, 0), SW_THREADS, 0, st>>>(n, m, 0, lv.c[0], 0, xyz1, xyz2, t, spa(d1, n, 0, 0));
    for (int li = 0; li < EMD_LEVELS - 1; ++li) {
        emd_sweep_kernel<1, REF><<<grid(d2, m, li), SW_THREADS, 0, st>>>(n, m, li, lv.c[li], 0, xyz1, xyz2, t, spa(d2, m, li, li + 1));
        if (li + 2 < EMD_LEVELS)                       // pass C of this level with pass A of the next (whose reach is the larger one)
            emd_sweep_kernel<3, REF><<<grid(d1, n, li + 1), SW_THREADS, 0, st>>>(n, m, li, lv.c[li], lv.c[li + 1], xyz1, xyz2, t, spa(d1, n, li + 1, li + 1));
        else                                           // the next level is the weightless one: it forms its own fL
            emd_sweep_kernel<2, REF><<<dim3(d1, b), SW_THREADS, 0, st>>>(n, m, li, lv.c[li], 0, xyz1, xyz2, t, SparseArgs{nullptr, -1, d1, -1, 0});
        GA_LAUNCH_CHECK();
    }
    emd_level0_kernel<<<b, 1024, 0, st>>>(n, m, EMD_LEVELS - 1, t);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

extern "C" int geoadv_emd_sparse_levels(int on) {
    g_emd_sparse.store(on ? 1 : 0);
    return GEOADV_OK;
}

// mode = weight mode (GEOADV_EMD_FAST / _REFERENCE), optionally | GEOADV_EMD_DENSE_LEVELS
static int emd_mode_check(const char *op, int mode) {
    const int w = mode & ~GEOADV_EMD_DENSE_LEVELS;
    GA_REQUIRE(w == GEOADV_EMD_FAST || w == GEOADV_EMD_REFERENCE, "%s: unknown weight mode %d", op, mode);
    return GEOADV_OK;
}

template <bool REF>
static int approx_match_impl(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, double *t, bool dense_levels, hipStream_t st) {
    EmdLevels<REF> lv;
    char *sparse = emd_sparse_floats(b, n, m) ? reinterpret_cast<char *>((reinterpret_cast<size_t>(t + (size_t)b * emd_temp_doubles_per_cloud(n, m)) + 15) & ~(size_t)15) : nullptr;
    if (int rc = emd_run_levels<REF>(b, n, m, xyz1, xyz2, t, sparse, lv, dense_levels, st)) return rc;
    emd_match_kernel<REF><<<dim3(cdiv(n, 256), cdiv(m, EMD_LT), b), 256, 0, st>>>(n, m, lv, xyz1, xyz2, t, match);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

extern "C" int geoadv_approx_match_mode(int mode, int b, int n, int m, const float *xyz1, const float *xyz2, float *match,
                                        float *temp, void *stream) {
    if (int rc = emd_mode_check("approx_match", mode)) return rc;
    if (int rc = emd_check("approx_match", b, n, m)) return rc;
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(xyz1 && xyz2 && match && temp, "approx_match: null pointer");
    hipStream_t st = as_stream(stream);
    double *t = reinterpret_cast<double *>((reinterpret_cast<size_t>(temp) + 7) & ~(size_t)7);
    const bool dense = (mode & GEOADV_EMD_DENSE_LEVELS) != 0;
    return (mode & ~GEOADV_EMD_DENSE_LEVELS) == GEOADV_EMD_REFERENCE ? approx_match_impl<true>(b, n, m, xyz1, xyz2, match, t, dense, st)
                                                                    : approx_match_impl<false>(b, n, m, xyz1, xyz2, match, t, dense, st);
}

extern "C" int geoadv_approx_match(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp,
                                   void *stream) {
    return geoadv_approx_match_mode(GEOADV_EMD_FAST, b, n, m, xyz1, xyz2, match, temp, stream);
}

extern "C" size_t geoadv_emd_cost_grad1_temp_floats(int b, int n, int m) {
    if (b <= 0 || n + m <= 0) return 16;
    return geoadv_approx_match_temp_floats(b, n, m) + 2 * (size_t)b * cdiv(n, 64) + 16;
}

template <bool REF>
static int emd_cost_grad1_impl(int b, int n, int m, const float *xyz1, const float *xyz2, float *cost, float *grad1, double *t,
                               bool dense_levels, hipStream_t st) {
    double *partial = t + (size_t)b * emd_temp_doubles_per_cloud(n, m);
    EmdLevels<REF> lv;
    char *sparse = emd_sparse_floats(b, n, m) ? reinterpret_cast<char *>((reinterpret_cast<size_t>(partial + (size_t)b * cdiv(n, 64)) + 15) & ~(size_t)15) : nullptr;
    if (int rc = emd_run_levels<REF>(b, n, m, xyz1, xyz2, t, sparse, lv, dense_levels, st)) return rc;
    const int parts = cdiv(n, 64);
    emd_plan_cost_grad1_kernel<REF><<<dim3(parts, b), SW_THREADS, 0, st>>>(n, m, lv, xyz1, xyz2, t, partial, grad1);
    GA_LAUNCH_CHECK();
    emd_cost_fold_kernel<<<b, 256, 0, st>>>(parts, partial, cost);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

extern "C" int geoadv_emd_cost_grad1_mode(int mode, int b, int n, int m, const float *xyz1, const float *xyz2, float *cost,
                                          float *grad1, float *temp, void *stream) {
    if (int rc = emd_mode_check("emd_cost_grad1", mode)) return rc;
    if (int rc = emd_check("emd_cost_grad1", b, n, m)) return rc;
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(xyz1 && xyz2 && cost && grad1 && temp, "emd_cost_grad1: null pointer");
    hipStream_t st = as_stream(stream);
    double *t = reinterpret_cast<double *>((reinterpret_cast<size_t>(temp) + 7) & ~(size_t)7);
    const bool dense = (mode & GEOADV_EMD_DENSE_LEVELS) != 0;
    return (mode & ~GEOADV_EMD_DENSE_LEVELS) == GEOADV_EMD_REFERENCE ? emd_cost_grad1_impl<true>(b, n, m, xyz1, xyz2, cost, grad1, t, dense, st)
                                                                    : emd_cost_grad1_impl<false>(b, n, m, xyz1, xyz2, cost, grad1, t, dense, st);
}

extern "C" int geoadv_emd_cost_grad1(int b, int n, int m, const float *xyz1, const float *xyz2, float *cost, float *grad1,
                                     float *temp, void *stream) {
    return geoadv_emd_cost_grad1_mode(GEOADV_EMD_FAST, b, n, m, xyz1, xyz2, cost, grad1, temp, stream);
}

extern "C" size_t geoadv_match_cost_workspace_floats(int b, int n, int m) {
    (void)n;
    if (b <= 0 || m <= 0) return 4;
    return 2 * (size_t)b * cdiv(m, EMD_COST_ROWS) + 4;     // one double per (cloud, part) + alignment slack
}

// workspace: geoadv_match_cost_workspace_floats(b, n, m) floats, caller-owned (the reference's pattern: temp tensors are the
// caller's, tf_approxmatch.cpp:164-170)
extern "C" int geoadv_match_cost_ws(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match, float *out,
                                    float *workspace, size_t workspace_floats, void *stream) {
    if (int rc = emd_check("match_cost", b, n, m)) return rc;
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(xyz1 && xyz2 && match && out, "match_cost: null pointer");
    GA_REQUIRE(workspace && workspace_floats >= geoadv_match_cost_workspace_floats(b, n, m), "match_cost: workspace too small (%zu floats, need %zu)",
               workspace_floats, geoadv_match_cost_workspace_floats(b, n, m));
    hipStream_t st = as_stream(stream);
    const int parts = cdiv(m, EMD_COST_ROWS);
    double *partial = reinterpret_cast<double *>((reinterpret_cast<size_t>(workspace) + 7) & ~(size_t)7);
    emd_cost_partial_kernel<<<dim3(parts, b), 256, 0, st>>>(n, m, xyz1, xyz2, match, partial);
    emd_cost_fold_kernel<<<b, 256, 0, st>>>(parts, partial, out);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

// reference-shaped (matchcostLauncher has no scratch argument): stream-ordered scratch of its own
extern "C" int geoadv_match_cost(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match, float *out,
                                 void *stream) {
    if (int rc = emd_check("match_cost", b, n, m)) return rc;
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(xyz1 && xyz2 && match && out, "match_cost: null pointer");
    hipStream_t st = as_stream(stream);
    const size_t wf = geoadv_match_cost_workspace_floats(b, n, m);
    float *ws = nullptr;                              // stream-ordered scratch: concurrent callers never share it
    GA_HIP(hipMallocAsync(reinterpret_cast<void **>(&ws), wf * sizeof(float), st));
    const int rc = geoadv_match_cost_ws(b, n, m, xyz1, xyz2, match, out, ws, wf, stream);
    GA_HIP(hipFreeAsync(ws, st));
    return rc;
}

extern "C" int geoadv_match_cost_grad(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                                      float *grad1, float *grad2, void *stream) {
    if (int rc = emd_check("match_cost_grad", b, n, m)) return rc;
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(xyz1 && xyz2 && match && grad1, "match_cost_grad: null pointer");
    hipStream_t st = as_stream(stream);
    emd_grad1_kernel<<<dim3(cdiv(n, 256), b), 256, 0, st>>>(n, m, xyz1, xyz2, match, grad1);
    GA_LAUNCH_CHECK();
    if (grad2) {                                       // (a caller that differentiates w.r.t. xyz1 only passes NULL)
        emd_grad2_kernel<<<dim3(m, b), 64, 0, st>>>(n, m, xyz1, xyz2, match, grad2);
        GA_LAUNCH_CHECK();
    }
    return GEOADV_OK;
}
