// Can the Chamfer scan's coordinate differences run on the MATRIX pipe beside the VALU?  dx[i][j] = tx[j] - px[i] is a K = 2
// product [1, -px[i]] . [tx[j], 1]^T: v_mfma_f32_32x32x2_f32 forms it with ONE rounding (fma(-px, 1, fma(1, tx, 0))), i.e. the bits of
// v_sub_f32.  The squares and sums must stay separately rounded VALU ops (the reference's arithmetic).  This probe prices the two
// forms of the same work -- per step and lane 16 pair distances, folded into 16 running row minima and one column minimum:
//   which 0: 48 v_sub + 48 v_mul + 32 v_add + minima, all VALU (the shipped scan's arithmetic per pair);
//   which 1: 3 MFMAs (dx, dy, dz of a 32 x 32 tile) + 48 v_mul + 32 v_add + minima.
// Operands come from LDS every step (3 ds_read_b32), results are folded so that nothing can be hoisted.  Also checks that the
// MFMA's differences equal the VALU's bit for bit on random operands (mismatch count in out[1]).  Measurement tooling only.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WHICH>
__global__ __launch_bounds__(512, 2) void sub_probe_kernel(const float *cols, const float *rows, float *out, int steps, int ncols) {
    __shared__ float sx[2048], sy[2048], sz[2048];
    for (int e = threadIdx.x; e < 2048; e += 512) {
        sx[e] = cols[3 * (e % ncols)]; sy[e] = cols[3 * (e % ncols) + 1]; sz[e] = cols[3 * (e % ncols) + 2];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x * 8 + wave;
    float rmin[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rmin[r] = 3.0e38f;
    float cmin = 3.0e38f;
    if (WHICH == 0) {
        // rows of this lane: 16 rows of a 32-row block, as the MFMA's output layout hands them out (row = 8 * (r / 4) + 4 * (lane / 32) + r % 4)
        float px[16], py[16], pz[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (g * 32 + 8 * (r / 4) + 4 * (lane >> 5) + (r & 3)) % 1024;
            px[r] = rows[3 * row]; py[r] = rows[3 * row + 1]; pz[r] = rows[3 * row + 2];
        }
        for (int s = 0; s < steps; ++s) {
            const int c = ((s * 32) & 2047) + (lane & 31);
            const float tx = sx[c], ty = sy[c], tz = sz[c];
            float cm = 3.0e38f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float dx = tx - px[r], dy = ty - py[r], dz = tz - pz[r];
                const float d = (dx * dx + dy * dy) + dz * dz;
                rmin[r] = fminf(rmin[r], d);
                cm = fminf(cm, d);
            }
            cmin = fminf(cmin, cm);
        }
    } else {
        // A operand of the 32 x 32 x 2 product: lanes 0..31 hold A[i][0] = 1, lanes 32..63 A[i][1] = -p[i]; B: lanes 0..31 B[0][j] = t[j],
        // lanes 32..63 B[1][j] = 1
        const int row = (g * 32 + (lane & 31)) % 1024;
        const bool hi = lane >= 32;
        const float ax = hi ? -rows[3 * row] : 1.0f, ay = hi ? -rows[3 * row + 1] : 1.0f, az = hi ? -rows[3 * row + 2] : 1.0f;
        // software-pipelined: the three MFMAs of step s + 1 are issued before the VALU work on step s's differences, so that
        // nothing waits for the matrix pipe (a single wave would otherwise sit in s_nops behind every MFMA)
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        auto tile = [&](int s, f32x16 &dx, f32x16 &dy, f32x16 &dz) {
            const int c = ((s * 32) & 2047) + (lane & 31);
            const float lx = sx[c], ly = sy[c], lz = sz[c];                  // (unconditional reads + selects: no exec-masked branches)
            const float bx = hi ? 1.0f : lx, by = hi ? 1.0f : ly, bz = hi ? 1.0f : lz;
            dx = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bx, zero, 0, 0, 0);
            dy = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, by, zero, 0, 0, 0);
            dz = __builtin_amdgcn_mfma_f32_32x32x2f32(az, bz, zero, 0, 0, 0);
        };
        f32x16 dx, dy, dz, nx, ny, nz;
        tile(0, dx, dy, dz);
        for (int s = 0; s < steps; ++s) {
            tile(s + 1, nx, ny, nz);
            float cm = 3.0e38f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = (dx[r] * dx[r] + dy[r] * dy[r]) + dz[r] * dz[r];
                rmin[r] = fminf(rmin[r], d);
                cm = fminf(cm, d);
            }
            cmin = fminf(cmin, cm);
            dx = nx; dy = ny; dz = nz;
        }
    }
    float acc = cmin;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = fminf(acc, rmin[r] + (float)r);
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = acc;
}

// bit-for-bit check of the MFMA differences against v_sub on random operands (one wave)
__global__ __launch_bounds__(64) void sub_check_kernel(const float *cols, const float *rows, int trials, unsigned *mism) {
    const int lane = threadIdx.x;
    const bool hi = lane >= 32;
    unsigned bad = 0;
    for (int t = 0; t < trials; ++t) {
        const float p = rows[(t * 32 + (lane & 31)) % 3072], q = cols[(t * 32 + (lane & 31)) % 6144];
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x2f32(hi ? -p : 1.0f, hi ? 1.0f : q, zero, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 8 * (r / 4) + 4 * (lane >> 5) + (r & 3), j = lane & 31;           // D[i][j] in register r of lane (j, i-half)
            const float pi = __shfl(p, i), qj = __shfl(q, j);
            const float want = qj - pi;
            bad += __float_as_uint(want) != __float_as_uint(d[r]);
        }
    }
    atomicAdd(mism, bad);
}

}  // namespace

// out_ms[0] = ms of `which` at `blocks` workgroups x 512 threads x `steps` steps; out_ms[1] = mismatches of the bit check
extern "C" int geoadv_probe_mfma_sub(int which, int blocks, int steps, int special, double *out_ms) {
    float *cols = nullptr, *rows = nullptr, *out = nullptr;
    unsigned *mism = nullptr;
    if (hipMalloc(&cols, 6144 * 4) != hipSuccess || hipMalloc(&rows, 3072 * 4) != hipSuccess || hipMalloc(&out, (size_t)blocks * 512 * 4) != hipSuccess ||
        hipMalloc(&mism, 4) != hipSuccess) return 1;
    float h[6144];
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f - 0.5f; };
    for (int i = 0; i < 6144; ++i) h[i] = rnd();
    if (special) {                   // awkward values: zeros, equal operands, tiny, huge, denormal, infinity
        const float sp[8] = {0.0f, -0.0f, 1e-38f, 1e-42f, 3e38f, -3e38f, __builtin_inff(), 0.25f};
        for (int i = 0; i < 6144; ++i) if ((i % 5) == 0) h[i] = sp[(i / 5) % 8];
    }
    hipMemcpy(cols, h, 6144 * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < 3072; ++i) h[i] = (special && (i % 3) == 0) ? h[(i * 7) % 6144] : rnd();
    hipMemcpy(rows, h, 3072 * 4, hipMemcpyHostToDevice);
    hipMemset(mism, 0, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        if (which == 0) sub_probe_kernel<0><<<blocks, 512, 0, 0>>>(cols, rows, out, steps, 2048);
        else sub_probe_kernel<1><<<blocks, 512, 0, 0>>>(cols, rows, out, steps, 2048);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    out_ms[0] = ms;
    sub_check_kernel<<<1, 64, 0, 0>>>(cols, rows, 4096, mism);
    unsigned hm = 0;
    hipMemcpy(&hm, mism, 4, hipMemcpyDeviceToHost);
    out_ms[1] = hm;
    const hipError_t err = hipGetLastError();
    hipFree(cols); hipFree(rows); hipFree(out); hipFree(mism);
    return err == hipSuccess ? 0 : 2;
}
