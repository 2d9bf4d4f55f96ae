"""The reference's operator API for the attack path, on MI355X.

Function names, argument order and output order are those of the reference's Python wrappers
(external/structural_losses/tf_nndistance.py:15-41, tf_approxmatch.py:10-50,
external/grouping/tf_grouping.py:8-75).  Inputs and outputs are torch tensors on an AMD GPU
(containers only: the arithmetic runs in libgeoadv.so's hand-written gfx950 kernels).
Shape errors raise ValueError (the reference raises InvalidArgumentError from OP_REQUIRES).
"""
import ctypes as C

import torch

from . import _lib


def _f32(t, name, rank):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU (got %s); there is no CPU path" % (name, t.device))
    if t.dtype != torch.float32:
        raise ValueError("%s must be float32 (got %s)" % (name, t.dtype))
    if t.dim() != rank:
        raise ValueError("%s must have rank %d (got shape %s)" % (name, rank, tuple(t.shape)))
    return t.contiguous()


def _i32(t, name, shape):
    if not t.is_cuda or t.dtype != torch.int32:
        raise ValueError("%s must be an int32 GPU tensor" % name)
    if tuple(t.shape) != tuple(shape):
        raise ValueError("%s must be of shape %s (got %s)" % (name, tuple(shape), tuple(t.shape)))
    return t.contiguous()


def _xyz_pair(xyz1, xyz2, op):
    xyz1 = _f32(xyz1, "xyz1", 3)
    xyz2 = _f32(xyz2, "xyz2", 3)
    if xyz1.shape[2] != 3:
        raise ValueError("%s only accepts 3d point set xyz1" % op)          # tf_nndistance.cpp:52
    if xyz2.shape[2] != 3:
        raise ValueError("%s only accepts 3d point set xyz2" % op)          # tf_nndistance.cpp:56
    if xyz1.shape[0] != xyz2.shape[0]:
        raise ValueError("%s expects xyz1 and xyz2 have same batch size" % op)  # tf_nndistance.cpp:58
    return xyz1, xyz2


NN_SYM_MIN_PAIRS = 1 << 26      # b * n * m from which nn_distance(kernel="auto") takes the symmetric scan (measured, tools/debug/nn_op_time.py:
                                # 32 x 2048 x 2048: 0.050 -> 0.039 ms, 256 x 2048 x 2048: 0.35 -> 0.22, 32 x 8192 x 8192: 0.62 -> 0.37; smaller
                                # problems are host-bound either way and keep the one-launch kernel)


def nn_distance(xyz1, xyz2, kernel="auto"):
    """tf_nndistance.py:15-26.  xyz1 (b,n,3), xyz2 (b,m,3) ->
    dist1 (b,n) squared distance from each xyz1 point to its nearest xyz2 point, idx1 (b,n) int32,
    dist2 (b,m), idx2 (b,m).  Bit-identical to the reference CPU op; lowest index wins ties.
    kernel: "scan" = the reference-shaped entry point geoadv_nn_distance (two scans, no scratch), "symmetric" = every pair distance
    once for both directions (nn_distance_sym: the attack loop's kernel, scratch from torch), "auto" = by size.  Same bits."""
    if kernel not in ("auto", "scan", "symmetric"):
        raise ValueError("kernel must be 'auto', 'scan' or 'symmetric'")
    xyz1, xyz2 = _xyz_pair(xyz1, xyz2, "NnDistance")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    if n >= 1 and m >= 1 and b >= 1 and (kernel == "symmetric" or (kernel == "auto" and b * n * m >= NN_SYM_MIN_PAIRS)):
        return nn_distance_sym(xyz1, xyz2)
    dist1 = torch.empty((b, n), dtype=torch.float32, device=xyz1.device)
    idx1 = torch.empty((b, n), dtype=torch.int32, device=xyz1.device)
    dist2 = torch.empty((b, m), dtype=torch.float32, device=xyz1.device)
    idx2 = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        st = _lib.lib().geoadv_nn_distance(b, n, _lib.ptr(xyz1), m, _lib.ptr(xyz2), _lib.ptr(dist1), _lib.ptr(idx1),
                                           _lib.ptr(dist2), _lib.ptr(idx2), _lib.stream_handle())
    _lib.check(st, "nn_distance")
    return dist1, idx1, dist2, idx2


def chamfer_screen(on):
    """Process-wide switch of the matrix-pipe-screened symmetric scan (csrc/chamfer_mx.h; default on): False = the unscreened scan
    everywhere.  Same results either way, bit for bit; returns the previous setting."""
    return bool(_lib.lib().geoadv_set_chamfer_screen(1 if on else 0))


def nn_distance_sym(xyz1, xyz2):
    """nn_distance with every pair distance evaluated ONCE for both directions (the attack loop's kernel): identical outputs,
    bit for bit; both clouds need at least one point."""
    xyz1, xyz2 = _xyz_pair(xyz1, xyz2, "NnDistance")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    if n < 1 or m < 1:
        raise ValueError("nn_distance_sym needs non-empty clouds")
    dist1 = torch.empty((b, n), dtype=torch.float32, device=xyz1.device)
    idx1 = torch.empty((b, n), dtype=torch.int32, device=xyz1.device)
    dist2 = torch.empty((b, m), dtype=torch.float32, device=xyz1.device)
    idx2 = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    wf = int(_lib.lib().geoadv_nn_distance_sym_workspace_floats(b, n, m))
    ws = torch.empty((wf,), dtype=torch.float32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        st = _lib.lib().geoadv_nn_distance_sym(b, n, _lib.ptr(xyz1), m, _lib.ptr(xyz2), _lib.ptr(dist1), _lib.ptr(idx1),
                                               _lib.ptr(dist2), _lib.ptr(idx2), _lib.ptr(ws), C.c_size_t(wf), _lib.stream_handle())
    _lib.check(st, "nn_distance_sym")
    return dist1, idx1, dist2, idx2


def chamfer_per_pc(dist1, dist2):
    """tf.reduce_mean(dist1, axis=1) + tf.reduce_mean(dist2, axis=1) of nn_distance's outputs (adv_ae.py:121,132;
    get_dists_per_point.py:75): (b,n), (b,m) -> (b,).  Summation order = the attack loop's own metrics, so a Chamfer
    distance recomputed from saved clouds is bit-identical to adversarial_metrics[:, :, 2]."""
    dist1, dist2 = _f32(dist1, "dist1", 2), _f32(dist2, "dist2", 2)
    if dist1.shape[0] != dist2.shape[0]:
        raise ValueError("chamfer_per_pc expects dist1 and dist2 have same batch size")
    b, n = dist1.shape
    m = dist2.shape[1]
    out = torch.empty((b,), dtype=torch.float32, device=dist1.device)
    with torch.cuda.device(dist1.device):
        st = _lib.lib().geoadv_chamfer_per_pc(b, n, m, _lib.ptr(dist1), _lib.ptr(dist2), _lib.ptr(out), _lib.stream_handle())
    _lib.check(st, "chamfer_per_pc")
    return out


def nn_distance_paired(xyz1, xyz2):
    """nn_distance for paired clouds of equal size (xyz1[j] close to xyz2[j] for most j, like adv = x + pert): exact
    grid search seeded with the pairing; identical results, fast when the pairing is good."""
    xyz1, xyz2 = _xyz_pair(xyz1, xyz2, "NnDistance")
    b, n, _ = xyz1.shape
    if xyz2.shape[1] != n:
        raise ValueError("nn_distance_paired needs clouds of equal size")
    dist1 = torch.empty((b, n), dtype=torch.float32, device=xyz1.device)
    idx1 = torch.empty((b, n), dtype=torch.int32, device=xyz1.device)
    dist2, idx2 = torch.empty_like(dist1), torch.empty_like(idx1)
    with torch.cuda.device(xyz1.device):
        st = _lib.lib().geoadv_nn_distance_paired(b, n, _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(dist1), _lib.ptr(idx1),
                                                  _lib.ptr(dist2), _lib.ptr(idx2), _lib.stream_handle())
    _lib.check(st, "nn_distance_paired")
    return dist1, idx1, dist2, idx2


def nn_distance_grad(xyz1, xyz2, grad_dist1, idx1, grad_dist2, idx2):
    """The NnDistanceGrad op behind tf_nndistance.py:35-41 -> (grad_xyz1, grad_xyz2)."""
    xyz1, xyz2 = _xyz_pair(xyz1, xyz2, "NnDistanceGrad")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    grad_dist1 = _f32(grad_dist1, "grad_dist1", 2)
    grad_dist2 = _f32(grad_dist2, "grad_dist2", 2)
    if tuple(grad_dist1.shape) != (b, n):
        raise ValueError("NnDistanceGrad requires grad_dist1 be of shape(batch,#points)")   # tf_nndistance.cpp:102
    if tuple(grad_dist2.shape) != (b, m):
        raise ValueError("NnDistanceGrad requires grad_dist2 be of shape(batch,#points)")   # tf_nndistance.cpp:104
    idx1 = _i32(idx1, "idx1", (b, n))
    idx2 = _i32(idx2, "idx2", (b, m))
    g1 = torch.empty_like(xyz1)
    g2 = torch.empty_like(xyz2)
    with torch.cuda.device(xyz1.device):
        st = _lib.lib().geoadv_nn_distance_grad(b, n, _lib.ptr(xyz1), m, _lib.ptr(xyz2), _lib.ptr(grad_dist1),
                                                _lib.ptr(idx1), _lib.ptr(grad_dist2), _lib.ptr(idx2), _lib.ptr(g1),
                                                _lib.ptr(g2), _lib.stream_handle())
    _lib.check(st, "nn_distance_grad")
    return g1, g2


class _NnDistanceFn(torch.autograd.Function):
    """Autograd glue equivalent to @ops.RegisterGradient('NnDistance') (tf_nndistance.py:35-41)."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        d1, i1, d2, i2 = nn_distance(xyz1, xyz2)
        ctx.save_for_backward(xyz1, xyz2, i1, i2)
        ctx.mark_non_differentiable(i1, i2)
        return d1, i1, d2, i2

    @staticmethod
    def backward(ctx, gd1, _gi1, gd2, _gi2):
        xyz1, xyz2, i1, i2 = ctx.saved_tensors
        g1, g2 = nn_distance_grad(xyz1, xyz2, gd1.contiguous(), i1, gd2.contiguous(), i2)
        return g1, g2


def nn_distance_autograd(xyz1, xyz2):
    """nn_distance with the registered gradient, for callers that differentiate through it."""
    return _NnDistanceFn.apply(xyz1, xyz2)


# ---------------------------------------------------------------------------------------------
# external/grouping/tf_grouping.py
# ---------------------------------------------------------------------------------------------
def _call(name, *args):
    st = getattr(_lib.lib(), name)(*args, _lib.stream_handle())
    _lib.check(st, name.replace("geoadv_", ""))


def query_ball_point(radius, nsample, xyz1, xyz2):
    """tf_grouping.py:8-20.  xyz1 (b,n,3) dataset, xyz2 (b,m,3) queries ->
    idx (b,m,nsample) int32 (first nsample points within radius, padded with the first hit),
    pts_cnt (b,m) int32."""
    xyz1, xyz2 = _xyz_pair(xyz1, xyz2, "QueryBallPoint")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = torch.zeros((b, m, int(nsample)), dtype=torch.int32, device=xyz1.device)
    cnt = torch.zeros((b, m), dtype=torch.int32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        _call("geoadv_query_ball_point", b, n, m, C.c_float(radius), int(nsample), _lib.ptr(xyz1), _lib.ptr(xyz2),
              _lib.ptr(idx), _lib.ptr(cnt))
    return idx, cnt


def select_top_k(k, dist):
    """tf_grouping.py:22-31.  dist (b,m,n) -> idx (b,m,n) int32, dist_out (b,m,n); the first k
    entries of every row are the k smallest in ascending order (the reference's swap order)."""
    dist = _f32(dist, "dist", 3)
    b, m, n = dist.shape
    outi = torch.empty((b, m, n), dtype=torch.int32, device=dist.device)
    out = torch.empty_like(dist)
    with torch.cuda.device(dist.device):
        _call("geoadv_selection_sort", b, n, m, int(k), _lib.ptr(dist), _lib.ptr(outi), _lib.ptr(out))
    return outi, out


def group_point(points, idx):
    """tf_grouping.py:33-41.  points (b,n,c), idx (b,m,nsample) -> out (b,m,nsample,c)."""
    points = _f32(points, "points", 3)
    b, n, c = points.shape
    if idx.dim() != 3 or idx.shape[0] != b:
        raise ValueError("GroupPoint expects idx of shape (batch, npoint, nsample)")
    idx = _i32(idx, "idx", idx.shape)
    m, ns = idx.shape[1], idx.shape[2]
    out = torch.empty((b, m, ns, c), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        _call("geoadv_group_point", b, n, c, m, ns, _lib.ptr(points), _lib.ptr(idx), _lib.ptr(out))
    return out


def group_point_grad(points, idx, grad_out):
    """The GroupPointGrad op behind tf_grouping.py:42-46 -> grad_points (b,n,c)."""
    points = _f32(points, "points", 3)
    b, n, c = points.shape
    idx = _i32(idx, "idx", idx.shape)
    m, ns = idx.shape[1], idx.shape[2]
    grad_out = _f32(grad_out, "grad_out", 4)
    if tuple(grad_out.shape) != (b, m, ns, c):
        raise ValueError("GroupPointGrad expects grad_out of shape (batch, npoint, nsample, channel)")
    gp = torch.empty_like(points)
    with torch.cuda.device(points.device):
        wb = int(_lib.lib().geoadv_group_point_grad_workspace_bytes(b, n, c, m, ns))
        ws = torch.empty(wb, dtype=torch.uint8, device=points.device)       # caller-owned scratch (torch's caching allocator)
        _call("geoadv_group_point_grad_ws", b, n, c, m, ns, _lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(gp), _lib.ptr(ws), C.c_size_t(wb))
    return gp


KNN_KERNELS = {"auto": 0, "all_points": 1, "grid": 2, "grid_shells": 3}     # include/geoadv.h GEOADV_KNN_*
_knn_default = "auto"


def _knn_kernel(kernel):
    k = _knn_default if kernel is None else kernel
    if k not in KNN_KERNELS:
        raise ValueError("kernel must be one of %s" % sorted(KNN_KERNELS))
    return KNN_KERNELS[k]


def knn_point(k, xyz1, xyz2, kernel=None):
    """tf_grouping.py:48-75.  xyz1 (b,n,3) dataset, xyz2 (b,m,3) queries -> val (b,m,k) squared
    distances ascending, idx (b,m,k) int32 -- fused (no (b,m,n) matrix), same order among equal
    distances as the reference's SelectionSort.
    kernel: which kernel answers THIS call -- "auto" (by size), "all_points", "grid", "grid_shells" (same results; None = the
    default set by knn_grid_mode).  The scratch is a torch tensor (caller-owned workspace of the C ABI's _ws form)."""
    xyz1, xyz2 = _xyz_pair(xyz1, xyz2, "knn_point")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    val = torch.empty((b, m, int(k)), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, m, int(k)), dtype=torch.int32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        wb = int(_lib.lib().geoadv_knn_workspace_bytes(b, n, m, int(k)))
        ws = torch.empty(wb, dtype=torch.uint8, device=xyz1.device)
        _call("geoadv_knn_point_ws", _knn_kernel(kernel), b, n, m, int(k), _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(val), _lib.ptr(idx),
              _lib.ptr(ws), C.c_size_t(wb))
    return val, idx


def knn_grid_mode(mode):
    """TESTS / MEASUREMENTS: the default k-NN kernel of this module's calls that pass no `kernel` ("auto" by size, "all_points",
    "grid", "grid_shells" = the grid search without its lane-private first pass) -- and of the C ABI's reference-shaped entry
    points (geoadv_knn_grid_mode).  Concurrent callers pass `kernel=` per call instead."""
    global _knn_default
    if mode not in KNN_KERNELS:
        raise ValueError("mode must be one of %s" % sorted(KNN_KERNELS))
    _lib.check(_lib.lib().geoadv_knn_grid_mode(KNN_KERNELS[mode]), "knn_grid_mode")
    _knn_default = mode


def knn_dists(pc, num_knn, kernel=None):
    """The graph of defender/get_knn_dists_per_point.py:78-81 fused: distances (not squared) from
    every point to its num_knn nearest neighbours, self dropped.  pc (b,n,3) -> (b,n,num_knn).  kernel: as knn_point."""
    pc = _f32(pc, "pc", 3)
    if pc.shape[2] != 3:
        raise ValueError("knn_dists only accepts 3d point sets")
    b, n, _ = pc.shape
    out = torch.empty((b, n, int(num_knn)), dtype=torch.float32, device=pc.device)
    with torch.cuda.device(pc.device):
        wb = int(_lib.lib().geoadv_knn_workspace_bytes(b, n, n, int(num_knn) + 1))
        ws = torch.empty(wb, dtype=torch.uint8, device=pc.device)
        _call("geoadv_knn_dists_ws", _knn_kernel(kernel), b, n, int(num_knn), _lib.ptr(pc), _lib.ptr(out), _lib.ptr(ws), C.c_size_t(wb))
    return out


# ---------------------------------------------------------------------------------------------
# the defenses' per-cloud bookkeeping (src/adversary_utils.py:149-178, src/ae_utils.py:12-80), device side
# ---------------------------------------------------------------------------------------------
def outlier_filter(point_clouds, knn_dists, knn_dist_thresh, top_k=None, want_outliers=True):
    """get_outlier_pc_inlier_pc(point_clouds, knn_dists, knn_dist_thresh) (adversary_utils.py:149-178) on GPU tensors ->
    (outlier_pc (b,n,3), outlier_idx (b,n) int16, outlier_num (b,) int16, inlier_pc (b,n,3)).
    knn_dists (b,n): the per-point scalar the reference function thresholds; or (b,n,k) with top_k: the kernel then forms
    the mean of the first top_k distances itself, as run_defense_surface.py:187-191 does with numpy.
    want_outliers=False skips the three outlier outputs (None)."""
    pc = _f32(point_clouds, "point_clouds", 3)
    if pc.shape[2] != 3:
        raise ValueError("outlier_filter only accepts 3d point sets")
    b, n, _ = pc.shape
    if knn_dists.dim() == 2:
        kd, stride, top_k = _f32(knn_dists, "knn_dists", 2), 1, 1
    else:
        kd = _f32(knn_dists, "knn_dists", 3)
        stride = kd.shape[2]
        top_k = stride if top_k is None else int(top_k)
    if tuple(kd.shape[:2]) != (b, n):
        raise ValueError("knn_dists must be of shape (batch, num_points[, k]); got %s for clouds %s" % (tuple(kd.shape), tuple(pc.shape)))
    inlier = torch.empty_like(pc)
    o_pc = torch.empty_like(pc) if want_outliers else None
    o_idx = torch.empty((b, n), dtype=torch.int16, device=pc.device) if want_outliers else None
    o_num = torch.empty((b,), dtype=torch.int16, device=pc.device) if want_outliers else None
    with torch.cuda.device(pc.device):
        _call("geoadv_outlier_filter", b, n, _lib.ptr(pc), _lib.ptr(kd), int(stride), int(top_k), C.c_float(float(knn_dist_thresh)),
              _lib.ptr(o_pc), _lib.ptr(o_idx), _lib.ptr(o_num), _lib.ptr(inlier))
    return o_pc, o_idx, o_num, inlier


def critical_split(point_clouds, max_val, max_idx):
    """get_critical_pc_non_critical_pc (ae_utils.py:51-80) from (np.max, np.argmax)(pre_symmetry, axis=1) on GPU tensors ->
    (critical_points (b,c,3), critical_idx (b,c) int16, critical_num (b,) int16, critical_pc (b,n,3), non_critical_pc (b,n,3)).
    Critical points owning equally many channels come in descending point index (include/geoadv.h)."""
    pc = _f32(point_clouds, "point_clouds", 3)
    if pc.shape[2] != 3:
        raise ValueError("critical_split only accepts 3d point sets")
    b, n, _ = pc.shape
    mv = _f32(max_val, "max_val", 2)
    c = mv.shape[1]
    if mv.shape[0] != b:
        raise ValueError("max_val must be of shape (batch, channels)")
    mi = _i32(max_idx, "max_idx", (b, c))
    cp = torch.empty((b, c, 3), dtype=torch.float32, device=pc.device)
    ci = torch.empty((b, c), dtype=torch.int16, device=pc.device)
    cn = torch.empty((b,), dtype=torch.int16, device=pc.device)
    cpc, ncpc = torch.empty_like(pc), torch.empty_like(pc)
    with torch.cuda.device(pc.device):
        _call("geoadv_critical_split", b, n, c, _lib.ptr(pc), _lib.ptr(mv), _lib.ptr(mi), _lib.ptr(cp), _lib.ptr(ci), _lib.ptr(cn),
              _lib.ptr(cpc), _lib.ptr(ncpc))
    return cp, ci, cn, cpc, ncpc


# ---------------------------------------------------------------------------------------------
# external/structural_losses/tf_approxmatch.py
# ---------------------------------------------------------------------------------------------
EMD_FAST, EMD_REFERENCE = 0, 1          # include/geoadv.h: how the pair weight expf(level * d2) is evaluated
EMD_DENSE_LEVELS = 0x100                # ... | this flag: every sweep of the call dense (no cell-grid form of the first three levels)


def emd_sparse_levels(on):
    """TESTS / MEASUREMENTS: the process default of EMD calls that pass no `dense_levels` (include/geoadv.h)."""
    _lib.check(_lib.lib().geoadv_emd_sparse_levels(int(bool(on))), "emd_sparse_levels")


def _emd_mode(reference_weights, dense_levels):
    return (EMD_REFERENCE if reference_weights else EMD_FAST) | (EMD_DENSE_LEVELS if dense_levels else 0)


def approx_match(xyz1, xyz2, reference_weights=False, dense_levels=False):
    """tf_approxmatch.py:10-18.  xyz1 (b,n,3), xyz2 (b,m,3) -> match (b,m,n): match[b,l,k] is the
    soft assignment between xyz2 point l and xyz1 point k (the reference GPU op's layout; the CPU
    op writes the transpose into the same declared shape).  Level schedule of the CPU op.
    reference_weights: every pair weight bit for bit the CPU op's (GEOADV_EMD_REFERENCE: every plan entry within ~2 float
    ulps of the CPU op, ~3.7x the time) instead of the fp32 fast form (typically 1e-6, rare entries 1e-4 relative).
    dense_levels: every sweep of THIS call dense (per call; the results differ only in the order of fp64 additions)."""
    xyz1, xyz2 = _xyz_pair(xyz1, xyz2, "ApproxMatch")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    match = torch.empty((b, m, n), dtype=torch.float32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        nf = _lib.lib().geoadv_approx_match_temp_floats(b, n, m)
        temp = torch.empty(int(nf), dtype=torch.float32, device=xyz1.device)
        _call("geoadv_approx_match_mode", _emd_mode(reference_weights, dense_levels), b, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2),
              _lib.ptr(match), _lib.ptr(temp))
    return match


def _match_args(xyz1, xyz2, match, op):
    xyz1, xyz2 = _xyz_pair(xyz1, xyz2, op)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    match = _f32(match, "match", 3)
    if tuple(match.shape) != (b, m, n):
        raise ValueError("%s expects (batch_size,#query,#dataset) match shape" % op)      # tf_approxmatch.cpp
    return xyz1, xyz2, match, b, n, m


def match_cost(xyz1, xyz2, match):
    """tf_approxmatch.py:21-32 -> cost (b,): sum over pairs of match * euclidean distance."""
    xyz1, xyz2, match, b, n, m = _match_args(xyz1, xyz2, match, "MatchCost")
    cost = torch.empty((b,), dtype=torch.float32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        wf = int(_lib.lib().geoadv_match_cost_workspace_floats(b, n, m))
        ws = torch.empty(wf, dtype=torch.float32, device=xyz1.device)
        _call("geoadv_match_cost_ws", b, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(match), _lib.ptr(cost), _lib.ptr(ws), C.c_size_t(wf))
    return cost


def match_cost_grad(xyz1, xyz2, match):
    """The MatchCostGrad op behind tf_approxmatch.py:38-50 -> (grad1 (b,n,3), grad2 (b,m,3)),
    before the scaling by the upstream grad_cost[b]."""
    xyz1, xyz2, match, b, n, m = _match_args(xyz1, xyz2, match, "MatchCostGrad")
    g1 = torch.empty_like(xyz1)
    g2 = torch.empty_like(xyz2)
    with torch.cuda.device(xyz1.device):
        _call("geoadv_match_cost_grad", b, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(match), _lib.ptr(g1), _lib.ptr(g2))
    return g1, g2


def emd_cost_grad1(xyz1, xyz2, reference_weights=False, dense_levels=False):
    """match_cost(xyz1, xyz2, approx_match(xyz1, xyz2)) and its gradient w.r.t. xyz1 with the plan held constant, without
    materialising the (b,m,n) plan -- the fused form the attack loop uses.  -> (cost (b,), grad1 (b,n,3))."""
    xyz1, xyz2 = _xyz_pair(xyz1, xyz2, "ApproxMatch")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    cost = torch.empty((b,), dtype=torch.float32, device=xyz1.device)
    g1 = torch.empty_like(xyz1)
    with torch.cuda.device(xyz1.device):
        nf = _lib.lib().geoadv_emd_cost_grad1_temp_floats(b, n, m)
        temp = torch.empty(int(nf), dtype=torch.float32, device=xyz1.device)
        _call("geoadv_emd_cost_grad1_mode", _emd_mode(reference_weights, dense_levels), b, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2),
              _lib.ptr(cost), _lib.ptr(g1), _lib.ptr(temp))
    return cost, g1


class _MatchCostFn(torch.autograd.Function):
    """@tf.RegisterGradient('MatchCost') (tf_approxmatch.py:38-50): grads scaled by grad_cost[b],
    no gradient w.r.t. match (ApproxMatch is registered NoGradient, :19)."""

    @staticmethod
    def forward(ctx, xyz1, xyz2, match):
        ctx.save_for_backward(xyz1, xyz2, match)
        return match_cost(xyz1, xyz2, match)

    @staticmethod
    def backward(ctx, grad_cost):
        xyz1, xyz2, match = ctx.saved_tensors
        g1, g2 = match_cost_grad(xyz1, xyz2, match)
        gc = grad_cost.reshape(-1, 1, 1)
        return g1 * gc, g2 * gc, None


def match_cost_autograd(xyz1, xyz2, match):
    return _MatchCostFn.apply(xyz1, xyz2, match)


# ---------------------------------------------------------------------------------------------
# attacker/prepare_indices_for_attack.py
# ---------------------------------------------------------------------------------------------
def chamfer_dist_matrix(pcs_a, pcs_b, max_workspace_bytes=2 << 30):
    """All-pairs Chamfer distance: out[i, j] = mean(dist1) + mean(dist2) of nn_distance(pcs_a[i], pcs_b[j])
    (the `chamfer_dist` tensor of prepare_indices_for_attack.py:113-114 for every pair), computed without
    tiling the clouds.  pcs_a (na,n,3), pcs_b (nb,m,3) GPU tensors -> (na, nb)."""
    a = _f32(pcs_a, "pcs_a", 3)
    b = _f32(pcs_b, "pcs_b", 3)
    if a.shape[2] != 3 or b.shape[2] != 3:
        raise ValueError("chamfer_dist_matrix only accepts 3d point sets")
    na, n, _ = a.shape
    nb, m, _ = b.shape
    out = torch.empty((na, nb), dtype=torch.float32, device=a.device)
    if na == 0 or nb == 0:
        return out
    with torch.cuda.device(a.device):
        want = _lib.lib().geoadv_chamfer_matrix_workspace_floats(na, nb, n, m)
        nfl = int(min(want, max(max_workspace_bytes // 4, _lib.lib().geoadv_chamfer_matrix_workspace_floats(1, 1, n, m))))
        ws = torch.empty(nfl, dtype=torch.float32, device=a.device)
        st = _lib.lib().geoadv_chamfer_matrix(na, nb, n, m, _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), _lib.ptr(ws),
                                              C.c_size_t(nfl), _lib.stream_handle())
    _lib.check(st, "chamfer_matrix")
    return out
