// Hand-off latency between two workgroups of ONE launch: a producer writes 256 bytes and raises a flag, a consumer polls the flag
// and reads the data back -- on the same XCD (one L2) or on different XCDs, with plain or agent-scope (sc1) data accesses.
// What the attack loop's in-launch hand-offs (common.h: ga_publish / ga_wait_flags) and any future per-cloud launch merging
// are priced against.  Workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8; the kernel records XCC_ID so that
// the assumption is checked, not trusted).  Measurement tooling: built into tools/probe/libgeoadv_probe_handoff.so, never shipped.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;                                               // 100 MHz
}
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

struct Slot {                      // one producer / consumer pair
    unsigned flag, ack, pad0[30];
    float data[64];
    unsigned long long t_write[64], t_seen[64], t_read[64];     // per round: producer before its stores, consumer after the poll, after the data
    unsigned mismatches, xcc_prod, xcc_cons, pad1;
};

// blocks [0, 8): producers (pair p); blocks [8, 16): consumers; consumer 8 + x serves pair (x + shift) % 8, so shift = 0 pairs
// workgroups of the same XCD and shift = 1 .. 7 workgroups of different ones.  sc1: data through agent-scope accesses.
template <bool SC1>
__global__ __launch_bounds__(64) void handoff_kernel(Slot *slots, int rounds, int shift) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const bool producer = b < 8;
    Slot *s = slots + (producer ? b : (b - 8 + shift) % 8);
    if (lane == 0) (producer ? s->xcc_prod : s->xcc_cons) = xcc_id();
    unsigned bad = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (producer) {
            if (lane == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(&s->ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)(r - 1) && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
                s->t_write[r - 1] = now();
            }
            __syncthreads();
            const float v = (float)r;
            if (SC1) __hip_atomic_store(&s->data[lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else s->data[lane] = v;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (lane == 0) __hip_atomic_store(&s->flag, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(&s->flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)r && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
                s->t_seen[r - 1] = now();
            }
            __syncthreads();
            float v;
            if (SC1) v = __hip_atomic_load(&s->data[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v = *reinterpret_cast<volatile float *>(&s->data[lane]);
            bad += v != (float)r;
            __syncthreads();
            if (lane == 0) {
                s->t_read[r - 1] = now();
                __hip_atomic_store(&s->ack, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (!producer) atomicAdd(&s->mismatches, bad);
}

}  // namespace

// out: per pair [xcc_prod, xcc_cons, mismatches, median flag latency (10 ns ticks), median data-read time (ticks)] as doubles
extern "C" int geoadv_probe_handoff(int sc1, int shift, int rounds, double *out) {
    if (rounds < 1 || rounds > 64) return 1;
    Slot *d = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&d), sizeof(Slot) * 8) != hipSuccess) return 2;
    (void)hipMemset(d, 0, sizeof(Slot) * 8);
    if (sc1) handoff_kernel<true><<<16, 64>>>(d, rounds, shift);
    else handoff_kernel<false><<<16, 64>>>(d, rounds, shift);
    if (hipDeviceSynchronize() != hipSuccess) { (void)hipFree(d); return 3; }
    static Slot h[8];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    for (int p = 0; p < 8; ++p) {
        double lat[64], rd[64];
        for (int r = 0; r < rounds; ++r) { lat[r] = (double)(h[p].t_seen[r] - h[p].t_write[r]); rd[r] = (double)(h[p].t_read[r] - h[p].t_seen[r]); }
        for (int i = 0; i < rounds; ++i)
            for (int j = i + 1; j < rounds; ++j) {
                if (lat[j] < lat[i]) { double t = lat[i]; lat[i] = lat[j]; lat[j] = t; }
                if (rd[j] < rd[i]) { double t = rd[i]; rd[i] = rd[j]; rd[j] = t; }
            }
        out[5 * p] = h[p].xcc_prod; out[5 * p + 1] = h[p].xcc_cons; out[5 * p + 2] = h[p].mismatches;
        out[5 * p + 3] = lat[rounds / 2]; out[5 * p + 4] = rd[rounds / 2];
    }
    return 0;
}
