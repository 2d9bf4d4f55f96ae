// Exact nearest-neighbour search with a uniform grid, for problems whose two clouds are PAIRED: nn_distance(P, Q)
// with equally many points where Q_j is a good guess for the neighbour of P_j -- the attack's source-distance term
// nn_distance(adv, x) with adv = x + pert (src/adv_ae.py:131): most points barely move.
//
// Results are the reference's, bit for bit (tf_nndistance.cpp:21-43): the same fp32 expression for every distance
// that is evaluated, minimum taken over (distance, index) lexicographically = strict '<' in ascending index order.
// Only the set of evaluated candidates shrinks: d(P_j, Q_j) bounds the answer from above, every target at least that
// close lies inside the ball of radius r = sqrt(bound) (1 + 1e-4) + 1e-5 around the query, and the cells the ball
// touches are enumerated conservatively (cell indices are monotone in the coordinates, the margins dwarf every fp32
// rounding involved).  Queries whose ball touches more than GR_MAX_SPAN cells along an axis or holds more than
// GR_LANE_BUDGET candidates -- the few points an attack moves far -- are queued and scanned against ALL targets, one wave
// per query.
//
// One workgroup = (cloud, direction, quarter of the queries); n <= 8192 (the sorted targets of a cloud live in LDS).  The targets are bucketed into a 16^3 grid fitted to their
// bounding box (so the cell size follows the scale of the shape; queries outside clamp to the boundary cells, which keeps
// containment) by a counting sort in LDS every call -- the adversarial cloud moves every iteration -- and stay there,
// sorted by cell, for the queries.
//
// The cost depends on the data, so it is bounded: after the sort every query knows exactly how many candidates its cells
// hold; if the slice averages more than GR_MEAN_BUDGET or has more than 1/GR_FAR_DIV far queries, the workgroup raises its
// `need` flag and leaves -- the caller's all-pairs kernel then computes that cloud -- and looks again only every
// GR_RETRY-th call.  Worst case: the all-pairs price plus one sort every 16 calls.
#include "chamfer_grid.h"
#include "chamfer_sym.h"

namespace geoadv {

template <int MAXN>
__global__ __launch_bounds__(GR_THREADS) void chamfer_grid_kernel(GridArgs a) { grid_nn_block<MAXN>(a, blockIdx.x, blockIdx.y, blockIdx.z); }

bool chamfer_grid_supports(int n, int m) { return n == m && n >= 1 && n <= GR_MAX_N_BIG; }
bool chamfer_grid_rides(int n) { return n >= 1 && n <= GR_MAX_N; }      // small enough to share the latent_decode launch (decoder.hip)

template <int MAXN>
static int launch_grid(const GridArgs &a, int b, hipStream_t stream) {
    static DeviceOnce attr;
    if (int rc = attr.run([]() -> int {
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chamfer_grid_kernel<MAXN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)chamfer_grid_lds_bytes(MAXN)));
            return GEOADV_OK;
        })) return rc;
    chamfer_grid_kernel<MAXN><<<dim3(b, 2, GR_QSPLIT), GR_THREADS, chamfer_grid_lds_bytes(a.n), stream>>>(a);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

// Both directions of nn_distance(P, Q), n == m <= 8192, exact.  Fast when P_j is near Q_j for most j.
int launch_chamfer_grid(const float *P, const float *Q, float *d1, int *i1, float *d2, int *i2, int b, int n, int *need, int call,
                        const float *box, hipStream_t stream) {
    if (b <= 0) return GEOADV_OK;
    GA_REQUIRE(chamfer_grid_supports(n, n), "chamfer_grid: needs 1 <= n <= %d", GR_MAX_N_BIG);
    const GridArgs a{P, Q, d1, i1, d2, i2, n, need, nullptr, call, box};
    return n <= GR_MAX_N ? launch_grid<GR_MAX_N>(a, b, stream) : launch_grid<GR_MAX_N_BIG>(a, b, stream);
}

// min / max of every cloud of Q: grid = b, 256 threads
__global__ __launch_bounds__(256) void chamfer_grid_box_kernel(const float *Q, int n, float *box) {
    __shared__ float red[4][6];
    const float *q = Q + (size_t)blockIdx.x * n * 3;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = threadIdx.x; i < n; i += 256)
#pragma unroll
        for (int c = 0; c < 3; ++c) { mn[c] = fminf(mn[c], q[3 * i + c]); mx[c] = fmaxf(mx[c], q[3 * i + c]); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int c = 0; c < 3; ++c) { mn[c] = fminf(mn[c], __shfl_xor(mn[c], off)); mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], off)); }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int c = 0; c < 3; ++c) { red[threadIdx.x >> 6][c] = mn[c]; red[threadIdx.x >> 6][3 + c] = mx[c]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        box[blockIdx.x * 6 + c] = fminf(fminf(red[0][c], red[1][c]), fminf(red[2][c], red[3][c]));
        box[blockIdx.x * 6 + 3 + c] = fmaxf(fmaxf(red[0][3 + c], red[1][3 + c]), fmaxf(red[2][3 + c], red[3][3 + c]));
    }
}

// bounding boxes of the b clouds of Q into box[b][6] (GridArgs::box)
int launch_chamfer_grid_box(const float *Q, int b, int n, float *box, hipStream_t stream) {
    if (b <= 0) return GEOADV_OK;
    chamfer_grid_box_kernel<<<b, 256, 0, stream>>>(Q, n, box);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

}  // namespace geoadv

extern "C" int geoadv_nn_distance_paired(int b, int n, const float *xyz1, const float *xyz2, float *dist1, int *idx1,
                                         float *dist2, int *idx2, void *stream) {
    GA_REQUIRE(b >= 0 && geoadv::chamfer_grid_supports(n, n), "nn_distance_paired: bad dimensions (b=%d n=%d, n <= %d)", b, n,
               geoadv::GR_MAX_N_BIG);
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(xyz1 && xyz2 && dist1 && idx1 && dist2 && idx2, "nn_distance_paired: null pointer");
    if (int rc = geoadv::launch_chamfer_grid(xyz1, xyz2, dist1, idx1, dist2, idx2, b, n, nullptr, 0, nullptr, geoadv::as_stream(stream))) return rc;
    return geoadv::launch_nn_nonfinite_fix(b, n, xyz1, n, xyz2, dist1, idx1, dist2, idx2, geoadv::as_stream(stream));
}
