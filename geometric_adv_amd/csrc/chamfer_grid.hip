// Exact nearest-neighbour search with a uniform grid, for problems whose two clouds are PAIRED: nn_distance(P, Q)
// with equally many points where Q_j is a good guess for the neighbour of P_j -- the attack's source-distance term
// nn_distance(adv, x) with adv = x + pert (src/adv_ae.py:131): most points barely move.
//
// Results are the reference's, bit for bit (tf_nndistance.cpp:21-43): the same fp32 expression for every distance
// that is evaluated, minimum taken over (distance, index) lexicographically = strict '<' in ascending index order.
// Only the set of evaluated candidates shrinks: d(P_j, Q_j) bounds the answer from above, every target at least that
// close lies inside the ball of radius r = sqrt(bound) (1 + 1e-4) + 1e-5 around the query, and the cells the ball
// touches are enumerated conservatively (cell indices are monotone in the coordinates, the margins dwarf every fp32
// rounding involved).  Queries whose ball touches more than GRID_MAX_SPAN^3 cells -- the few points an attack moves
// far -- are queued and scanned against ALL targets, one wave per query.
//
// One workgroup = (cloud, direction, quarter of the queries).  The targets are bucketed into a 16^3 grid over [-0.5, 0.5]^3 (coordinates
// outside clamp to the boundary cells, which keeps containment) by a counting sort in LDS every call -- the adversarial
// cloud moves every iteration -- and stay there, sorted by cell, for the queries.
#include "chamfer_grid.h"

namespace geoadv {

__global__ __launch_bounds__(GR_THREADS) void chamfer_grid_kernel(GridArgs a) { grid_nn_block(a, blockIdx.x, blockIdx.y, blockIdx.z); }

bool chamfer_grid_supports(int n, int m) { return n == m && n >= 1 && n <= GR_MAX_N; }

// Both directions of nn_distance(P, Q), n == m <= 4096, exact.  Fast when P_j is near Q_j for most j.
int launch_chamfer_grid(const float *P, const float *Q, float *d1, int *i1, float *d2, int *i2, int b, int n, int *need,
                        hipStream_t stream) {
    if (b <= 0) return GEOADV_OK;
    GA_REQUIRE(chamfer_grid_supports(n, n), "chamfer_grid: needs 1 <= n <= %d", GR_MAX_N);
    static bool attr = false;
    if (!attr) {
        GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chamfer_grid_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)chamfer_grid_lds_bytes(GR_MAX_N)));
        attr = true;
    }
    const GridArgs a{P, Q, d1, i1, d2, i2, n, need};
    chamfer_grid_kernel<<<dim3(b, 2, GR_QSPLIT), GR_THREADS, chamfer_grid_lds_bytes(n), stream>>>(a);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

}  // namespace geoadv

extern "C" int geoadv_nn_distance_paired(int b, int n, const float *xyz1, const float *xyz2, float *dist1, int *idx1,
                                         float *dist2, int *idx2, void *stream) {
    GA_REQUIRE(b >= 0 && geoadv::chamfer_grid_supports(n, n), "nn_distance_paired: bad dimensions (b=%d n=%d, n <= %d)", b, n,
               geoadv::GR_MAX_N);
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(xyz1 && xyz2 && dist1 && idx1 && dist2 && idx2, "nn_distance_paired: null pointer");
    return geoadv::launch_chamfer_grid(xyz1, xyz2, dist1, idx1, dist2, idx2, b, n, nullptr, geoadv::as_stream(stream));
}
