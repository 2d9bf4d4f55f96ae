// The decoder backward's last kernel body (dd2 -> dd1 -> dz, then the encoder's backward through the pool Jacobian), shared by
// decoder.hip (a launch of its own) and encoder.hip (the same workgroups + the dense recomputing backward for clouds with a
// tied pool maximum in ONE launch).
#pragma once
#include "ae.h"

namespace geoadv {

constexpr int LD_THREADS = 1024;

// out[o] (o < 256) = sum_k in[k] * Wt[k][256 + ...]: K split over 4 thread groups, partials in LDS,
// summed in a fixed order by the first 256 threads.
// THREADS = 1024: one K quarter per thread; 512: two (quarters ks and ks + 2, one after the other) -- the four partial sums
// and their order are the same either way, so both give the same bits.
template <int K, int THREADS = LD_THREADS>
__device__ __forceinline__ float fc256_split4(const float *in_lds, const float *W /*[K][256]*/, float (*part)[256]) {
    const int t = threadIdx.x, o = t & 255;
    constexpr int PER = K / 4;
#pragma unroll
    for (int ks = t >> 8; ks < 4; ks += THREADS / 256) {
        float w[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) w[k] = W[(size_t)(ks * PER + k) * 256 + o];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) s = fmaf(in_lds[ks * PER + k], w[k], s);
        part[ks][o] = s;
    }
    __syncthreads();
    return ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];     // meaningful for t < 256
}

// The same product for the output columns [col0, col0 + NCOLS) only (a workgroup that shares a cloud's layer with others: every
// column's four partial sums and their order are those of fc256_split4, so the bits do not depend on the split).  4 * NCOLS threads
// work, one K quarter each; returns the column's sum to threads t < NCOLS (column col0 + t).
template <int K, int NCOLS>
__device__ __forceinline__ float fc256_split4_cols(const float *in_lds, const float *W /*[K][256]*/, float (*part)[256], const int col0) {
    const int t = threadIdx.x, o = col0 + t % NCOLS, ks = t / NCOLS;
    constexpr int PER = K / 4;
    if (ks < 4) {
        float w[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) w[k] = W[(size_t)(ks * PER + k) * 256 + o];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) s = fmaf(in_lds[ks * PER + k], w[k], s);
        part[ks][o] = s;
    }
    __syncthreads();
    return ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];     // meaningful for t < NCOLS
}

// dd2 = sum of partials, masked by d2 > 0; dd1 = dd2 @ V1^T masked by d1 > 0; dz = dd1 @ V0^T.
// grid = clouds, 1024 threads.  TF ReluGrad masks by the layer OUTPUT being > 0.
// (Requesting V1^T / V0^T ahead of the partial sums was measured and lost, 7.8 vs 7.2 us: a workgroup streams ~480 KB through
// ONE CU, so the early weights only delay the partials the chain starts with.)
// ja.jac != null: the encoder's part of the backward happens here too -- g_enc[b][p] = sum over the channels c with
// crit[b][c] == p of dz[b][c] * J[b][c] (encoder_jac.h), channels in ascending order; every channel of a point writes the
// same total.  Clouds with a tied pool maximum are left to the dense recomputing kernel.
struct JacApply { const int *crit; const float *jac; const int *dense; float *g_enc; int n; };
__device__ __forceinline__ bool is_dense_all(const JacApply &ja, int b) { return ja.dense && ja.dense[b] != 0; }   // (uniform)

// ready (or null): where a tied cloud's workgroup announces that dz[b] is in memory (the dense recomputing backward of the same
// launch waits for it, encoder.hip: decoder_tail_dense_kernel); epoch: the value to store
__device__ __forceinline__ void decoder_bwd_tail_body(const DeviceAE &A, int batch, int chunks, const float *partial,
                                                      const float *d1, const float *d2, float *dz, const JacApply &ja,
                                                      const int b, unsigned *ready, unsigned epoch) {
    __shared__ float g2[256];
    __shared__ float g1[256];
    __shared__ float part[8][256];
    __shared__ __attribute__((aligned(16))) float dzs[128], jxs[128], jys[128], jzs[128];
    __shared__ __attribute__((aligned(16))) int crs[128];
    const int t = threadIdx.x, o = t & 255, ks = t >> 8;
    GA_STAMP(4, 0);
    // the Jacobian rows and critical points of this cloud do not depend on anything computed here: requested first
    float jx = 0.f, jy = 0.f, jz = 0.f;
    int my_crit = -1, is_dense = 0;
    if (ja.jac && t < 128) {
        const float *jp = ja.jac + ((size_t)b * 128 + t) * 3;
        jx = jp[0]; jy = jp[1]; jz = jp[2];
        my_crit = ja.crit[(size_t)b * 128 + t];
        is_dense = ja.dense[b];
    }
    {   // split-K partials: 4 contiguous chunk groups, ascending inside, merged in order
        const int cb = chunks * ks / 4, ce = chunks * (ks + 1) / 4;
        float s = 0.f;
#pragma unroll 8
        for (int ch = cb; ch < ce; ++ch) s += partial[((size_t)ch * batch + b) * 256 + o];
        part[ks][o] = s;
    }
    __syncthreads();
    if (t < 256) {
        const float s = ((part[0][t] + part[1][t]) + part[2][t]) + part[3][t];
        g2[t] = d2[(size_t)b * 256 + t] > 0.f ? s : 0.f;
    }
    __syncthreads();
    GA_STAMP(4, 1);
    {
        const float s = fc256_split4<256>(g2, A.v1t, part);
        __syncthreads();
        if (t < 256) g1[t] = d1[(size_t)b * 256 + t] > 0.f ? s : 0.f;
    }
    __syncthreads();
    GA_STAMP(4, 2);
    {   // dz: 128 outputs x 8 K-slices of 32
        const int c = t & 127, k8 = t >> 7;
        float w[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) w[k] = A.v0t[(size_t)(k8 * 32 + k) * 128 + c];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) s = fmaf(g1[k8 * 32 + k], w[k], s);
        part[k8][c] = s;
        __syncthreads();
        float r = 0.f;
        if (t < 128) {
            r = part[0][t];
#pragma unroll
            for (int q = 1; q < 8; ++q) r += part[q][t];
            dz[(size_t)b * 128 + t] = r;
        }
        // (The fences are not decoration: the dense blocks may run on another XCD, whose L2 is not coherent with this one inside a
        // launch -- tools/handoff_probe.py: plain loads across XCDs returned stale data in 32 736 of 32 768 reads.)
        if (ready && is_dense_all(ja, b)) {   // rare: a tied cloud -- publish dz for the dense blocks of this launch (agent scope)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its stores ...
            __syncthreads();                                       // ... before ONE lane releases and raises the flag
            if (t == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(ready + b, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        GA_STAMP(4, 3);
        if (ja.jac) {
            if (t < 128) { dzs[t] = r; crs[t] = my_crit; jxs[t] = jx; jys[t] = jy; jzs[t] = jz; }
            __syncthreads();
            // g(point of channel ch) = sum over the channels c of that point of dz[c] * J[c]: thread (ch, part) adds the 16 channels
            // [16 part, 16 part + 16) in ascending order (all LDS reads of a thread requested at once: vector reads, broadcast),
            // the 8 parts are then added in ascending order -- a fixed order, so every channel of a point writes the same bits
            {
                const int ch = t & 127, pq = t >> 7, mine = crs[ch];
                float gx = 0.f, gy = 0.f, gz = 0.f;
                const int4 *c4 = reinterpret_cast<const int4 *>(crs) + 4 * pq;
                const float4 *d4 = reinterpret_cast<const float4 *>(dzs) + 4 * pq, *x4 = reinterpret_cast<const float4 *>(jxs) + 4 * pq;
                const float4 *y4 = reinterpret_cast<const float4 *>(jys) + 4 * pq, *z4 = reinterpret_cast<const float4 *>(jzs) + 4 * pq;
                int4 cc[4]; float4 dd[4], xx[4], yy[4], zz[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { cc[q] = c4[q]; dd[q] = d4[q]; xx[q] = x4[q]; yy[q] = y4[q]; zz[q] = z4[q]; }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ci[4] = {cc[q].x, cc[q].y, cc[q].z, cc[q].w};
                    const float di[4] = {dd[q].x, dd[q].y, dd[q].z, dd[q].w}, xi[4] = {xx[q].x, xx[q].y, xx[q].z, xx[q].w};
                    const float yi[4] = {yy[q].x, yy[q].y, yy[q].z, yy[q].w}, zi[4] = {zz[q].x, zz[q].y, zz[q].z, zz[q].w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float d = ci[u] == mine ? di[u] : 0.f;
                        gx = fmaf(d, xi[u], gx); gy = fmaf(d, yi[u], gy); gz = fmaf(d, zi[u], gz);
                    }
                }
                __syncthreads();                                   // (part[][] is free: the dz partials were consumed above)
                part[pq][ch] = gx; part[pq][128 + ch] = gy;
                __syncthreads();
                float sx = 0.f, sy = 0.f;
                if (t < 128) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { sx += part[q][t]; sy += part[q][128 + t]; }
                }
                __syncthreads();
                part[pq][ch] = gz;
                __syncthreads();
                if (t < 128 && !is_dense) {
                    float sz = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) sz += part[q][t];
                    float *g = ja.g_enc + ((size_t)b * ja.n + my_crit) * 3;
                    g[0] = sx; g[1] = sy; g[2] = sz;
                }
            }
        }
    }
    GA_STAMP(4, 7);
}


// arguments of the merged tail + dense launch (encoder.hip: decoder_tail_dense_kernel)
struct TailDenseArgs {
    int batch, chunks, n;
    const float *partial, *d1, *d2;
    float *dz;
    JacApply ja;
    const float *adv, *z;
    const int *zcnt, *dense_flag;
    float *g_enc;
    unsigned *ready;              // [batch] epoch of the last step whose dz a tied cloud's tail block published
    unsigned epoch;
    int *spin_timeout;            // set to 1 if a dense block ever gave up waiting (never observed; bounds the spin)
};

}  // namespace geoadv
