// Interface of the symmetric Chamfer scan (chamfer_sym.hip) shared with its callers (attack.hip, train.hip).
#pragma once
#include "common.h"

namespace geoadv {

struct ChamferPair {
    const float *p, *q;        // [b][n][3] rows, [b][m][3] columns
    float *dist1; int *idx1;   // [b][n]  row minima  (nn_distance outputs 0,1)
    float *dist2; int *idx2;   // [b][m]  column minima (outputs 2,3)
};

// What launch_chamfer_sym_loop leaves to the caller's next launch when asked to (`defer`): the row minima as one
// (distance, index) partial per column slice -- [pair][cloud][slice][n] -- whose lexicographic minimum is dist1 / idx1.
// deferred == false: dist1 / idx1 are final in the pairs' own arrays (one slice, or the merge launch ran).
// row64 (narrow column slices, i.e. small batches: more than 8 partials per row would cost a merge launch or a heavy consumer):
// the scan instead folds every (row, slice) into ONE packed word per row -- (distance bits << 32) | index, 64-bit unsigned atomic
// minimum: squared distances are >= +0, so the order of the words is the lexicographic order of (distance, index) -- in
// row64[pair][cloud][n], which must hold all ones when the scan starts (the loop's FC2 launch fills it on its way).
struct SymPartials {
    const float *rowpart_d;
    const int *rowpart_i;
    int slices, clouds;
    bool deferred;
    unsigned long long *row64;       // in: the caller's packed buffer (or null: never use the atomic form); out: null unless used
};

#ifdef __HIPCC__
__device__ __forceinline__ bool sym_needed(const int *need, int c) {
    if (!need) return true;
    const int4 lo = reinterpret_cast<const int4 *>(need)[2 * c], hi = reinterpret_cast<const int4 *>(need)[2 * c + 1];
    return (lo.x | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) != 0;
}

// Lexicographic (distance, index) minimum over `slices` partials `stride` elements apart.  The slices are column ranges in
// ascending order, so the lowest slice attaining the minimal distance holds the lowest index: only the DISTANCES are compared
// (eight loads in flight per step) and the winner's index is one further load.
__device__ __forceinline__ float sym_merge_pick(const float *d, int slices, size_t stride, int &slice) {
    float bd = d[0];
    int bs = 0;
    for (int s0 = 1; s0 < slices; s0 += 8) {
        float vd[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) vd[u] = d[(size_t)(s0 + u < slices ? s0 + u : 0) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s0 + u < slices && vd[u] < bd) { bd = vd[u]; bs = s0 + u; }
    }
    slice = bs;
    return bd;
}
__device__ __forceinline__ void sym_merge_slices(const float *d, const int *i, int slices, size_t stride, float &od, int &oi) {
    int s;
    od = sym_merge_pick(d, slices, stride, s);
    oi = i[(size_t)s * stride];
}
__device__ __forceinline__ unsigned long long sym_pack(float d, int i) {
    return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)i;
}
// the distance alone
__device__ __forceinline__ float sym_merge_min(const float *d, int slices, size_t stride) {
    float bd = d[0];
    for (int s0 = 1; s0 < slices; s0 += 8) {
        float vd[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) vd[u] = d[(size_t)(s0 + u < slices ? s0 + u : 0) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) bd = fminf(bd, vd[u]);
    }
    return bd;
}
#endif

// Chamfer gradient w.r.t. the FIRST cloud of a problem (attack.hip: chamfer_grad_attack_*_kernel; shared with train.hip)
struct CGradProblem {
    const float *p, *q;          // [B][n][3] own / other cloud
    const int *idx1, *idx2;      // [B][n] own->other matches, other->own matches
    float *g;                    // [B][n][3]
    const float *w;              // [B] or null: upstream factor (dist_weight)
    const int *jstar;            // [B] or null: point receiving the extra max-term
    float extra_w;               // max_point_dist_weight (0 = none)
    // idx1 still as the symmetric scan's (distance, index) partials per column slice (null: idx1 is final); this kernel then
    // takes their lexicographic minimum on its way in and leaves it in idx1_out.  part_need: only the clouds flagged there.
    const float *part_d; const int *part_i; int part_slices; const int *part_need; int *idx1_out;
    const unsigned long long *part_w;      // ... or as packed (distance, index) words [B][n] (SymPartials::row64): the index is the low half
};
int launch_chamfer_grad(const CGradProblem *pr, int np, int B, int n, hipStream_t st);

// the reference's rule for a NaN distance to candidate 0, applied to finished nn_distance outputs (chamfer.hip)
int launch_nn_nonfinite_fix(int b, int n, const float *xyz1, int m, const float *xyz2, float *dist1, int *idx1, float *dist2,
                            int *idx2, hipStream_t stream);

size_t chamfer_sym_workspace_floats(int pairs, int b, int n, int m);
bool chamfer_sym_packs_rows(long live_groups, int n, int m);
int launch_chamfer_sym(const ChamferPair *pairs, int np, int b, int n, int m, float *workspace, hipStream_t stream);
struct GridArgs;
struct JacRider;
struct LossRider;
// loss (or null): the loop's loss + gradient workgroups as the LAST riders of the launch (loss_cgrad.h; first_block / blocks /
// target are set here).  Returns with loss->blocks == 0 when the launch could not host them (the caller then launches them itself).
int launch_chamfer_sym_loop(const ChamferPair *pairs, int np, int b, int n, int m, float *workspace, const int *need1,
                            const GridArgs *rider, const JacRider *jac, SymPartials *defer, hipStream_t stream, LossRider *loss = nullptr);
// can the symmetric scan's launch of this shape host the loss riders?  (the unscreened kernel, one row super-tile, whole groups of 8 clouds)
bool chamfer_sym_hosts_loss(long live_groups, int b, int n, int m);

}  // namespace geoadv
