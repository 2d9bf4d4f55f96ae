"""The two defenses (defender/get_knn_dists_per_point.py + defender/run_defense_surface.py:187-207 with
src/adversary_utils.get_outlier_pc_inlier_pc; defender/run_defense_critical.py:180-196 with src/ae_utils.py:12-80), logic
only -- the file / CLI plumbing is out of scope.

Everything between the adversarial clouds and the defended reconstructions runs on the GPU (round 4): the fused kNN kernel,
the outlier / critical-point packing kernels (csrc/defense.hip, the reference does them with per-cloud numpy loops), the
victim AE and the Chamfer score.  `*_device` functions take and return GPU tensors; the functions with the reference's names
wrap them for numpy callers.
"""
import numpy as np
import torch

from . import ops


KNN_BATCH = 100            # get_knn_dists_per_point.py:74


def _device(device):
    """None = the CURRENT CUDA device (a rank of a multi-GPU run works on its own GPU, not on cuda:0)."""
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def _dev(a, device):
    t = torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)) if not isinstance(a, torch.Tensor) else a
    return t.to(device=_device(device), dtype=torch.float32).contiguous()


def get_knn_dists(point_clouds, num_knn=8, device=None, batch=None):
    """knn_dists [num_pc, num_points, num_knn] (get_knn_dists_per_point.py:70-83).  The reference feeds 100 clouds at a time
    because its graph holds a (100, n, n) matrix; the fused kernel holds nothing of that size, so by default one launch serves
    up to 65535 clouds (the kernel's grid limit) and larger sets are chunked.  batch: chunk size (the reference's 100 =
    KNN_BATCH), for callers that want its memory profile."""
    n_pc = len(point_clouds)
    step = int(batch) if batch else 65535
    if n_pc <= step:
        return ops.knn_dists(_dev(point_clouds, device), num_knn).cpu().numpy()
    out = np.empty(tuple(point_clouds.shape[:2]) + (int(num_knn),), np.float32)
    for s0 in range(0, n_pc, step):
        out[s0:s0 + step] = ops.knn_dists(_dev(point_clouds[s0:s0 + step], device), num_knn).cpu().numpy()
    return out


def get_outlier_pc_inlier_pc(point_clouds, knn_dists, knn_dist_thresh, device=None):
    """adversary_utils.py:149-178 for numpy callers.  knn_dists is the per-point scalar the caller thresholds."""
    o_pc, o_idx, o_num, i_pc = ops.outlier_filter(_dev(point_clouds, device), _dev(knn_dists, device), knn_dist_thresh)
    return o_pc.cpu().numpy(), o_idx.cpu().numpy(), o_num.cpu().numpy(), i_pc.cpu().numpy()


def defend_surface_device(ae, adversarial_pc, source_pc, num_knn=8, top_k=2, knn_dist_thresh=0.04):
    """run_defense_surface.py:187-207 for one set of clouds, GPU tensors in and out: filter off-surface points, reconstruct the
    defended clouds with the victim AE, score them against the sources."""
    adv, src = ae._as_dev(adversarial_pc), ae._as_dev(source_pc)
    knn = ops.knn_dists(adv, num_knn)
    _, o_idx, o_num, inlier = ops.outlier_filter(adv, knn, knn_dist_thresh, top_k=top_k)
    recon, _ = ae.forward(inlier)
    err = ae.loss_per_pc_tensor(recon, src)
    return dict(knn_dists=knn, outlier_idx=o_idx, outlier_num=o_num, defended_pc=inlier, defended_recon=recon,
                recon_error_vs_source=err)


def get_critical_pc_non_critical_pc(point_clouds, max_val, max_idx, device=None):
    """src/ae_utils.py:51-80 for numpy callers, with (max_val, max_idx) = (np.max, np.argmax)(pre_symmetry_data, axis=1)
    supplied by the fused encoder (PointNetAE.max_and_argmax) instead of the (num_pc, n, 128) tensor itself."""
    mi = torch.as_tensor(np.ascontiguousarray(max_idx, dtype=np.int32)).to(_device(device))
    out = ops.critical_split(_dev(point_clouds, device), _dev(max_val, device), mi)
    return tuple(t.cpu().numpy() for t in out)


def get_critical_points(point_clouds, max_val, max_idx, device=None):
    """src/ae_utils.py:12-48 (without its file saving) -> (critical_points [num_pc, bneck, 3], idx_critical int16, num_critical
    int16): the first three outputs of the device kernel behind get_critical_pc_non_critical_pc."""
    return get_critical_pc_non_critical_pc(point_clouds, max_val, max_idx, device)[:3]


def get_complementary_idx(idx, n):
    """src/general_utils.py:84-91: the indices of range(n) that are not in idx, ascending."""
    comp = np.full(n, True)
    comp[np.asarray(idx, dtype=int)] = False
    return np.arange(n, dtype=int)[comp]


def defend_critical_device(ae, adversarial_pc, source_pc):
    """run_defense_critical.py:180-196 for one set of clouds, GPU tensors in and out: drop the critical points of each
    adversarial cloud, reconstruct what is left and score it against the source."""
    adv, src = ae._as_dev(adversarial_pc), ae._as_dev(source_pc)
    mv, mi = ae.max_and_argmax(adv)
    crit_pts, crit_idx, crit_num, pc_critical, pc_defended = ops.critical_split(adv, mv, mi)
    recon, _ = ae.forward(pc_defended)
    err = ae.loss_per_pc_tensor(recon, src)
    return dict(critical_points=crit_pts, critical_idx=crit_idx, critical_num=crit_num, critical_pc=pc_critical,
                defended_pc=pc_defended, defended_recon=recon, recon_error_vs_source=err)


def _to_numpy(d):
    return {k: v.cpu().numpy() for k, v in d.items()}


def defend_surface(ae, adversarial_pc, source_pc, num_knn=8, top_k=2, knn_dist_thresh=0.04):
    """defend_surface_device for numpy callers (one download per result array at the end)."""
    out = defend_surface_device(ae, adversarial_pc, source_pc, num_knn, top_k, knn_dist_thresh)
    ae.status()                                  # (the f16x2 encoder's range guard: raises instead of returning +inf-born numbers)
    return _to_numpy(out)


def defend_critical(ae, adversarial_pc, source_pc):
    """defend_critical_device for numpy callers."""
    out = defend_critical_device(ae, adversarial_pc, source_pc)
    ae.status()
    return _to_numpy(out)
