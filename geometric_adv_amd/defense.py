"""Off-surface defense (defender/get_knn_dists_per_point.py + defender/run_defense_surface.py
+ src/adversary_utils.get_outlier_pc_inlier_pc), logic only -- the file/CLI plumbing is out of scope.

kNN distances come from the fused gfx950 kernel (ops.knn_dists); the outlier/inlier packing keeps
the reference's numpy semantics (stable compaction, last point duplicated as padding, int16
bookkeeping) and runs on the host exactly like the reference does.
"""
import numpy as np
import torch

from . import ops

KNN_BATCH = 100            # get_knn_dists_per_point.py:74


def get_knn_dists(point_clouds, num_knn=8, device="cuda:0", batch=KNN_BATCH):
    """knn_dists [num_pc, num_points, num_knn] for numpy clouds, in chunks of 100 like the reference."""
    out = np.empty(point_clouds.shape[:2] + (num_knn,), np.float32)
    for s in range(0, len(point_clouds), batch):
        pc = torch.as_tensor(np.ascontiguousarray(point_clouds[s:s + batch], dtype=np.float32)).to(device)
        out[s:s + batch] = ops.knn_dists(pc, num_knn).cpu().numpy()
    return out


def get_outlier_pc_inlier_pc(point_clouds, knn_dists, knn_dist_thresh):
    """adversary_utils.py:149-178.  knn_dists here is the per-point scalar the caller thresholds
    (run_defense_surface.py:187-191 passes the mean of the first two kNN distances)."""
    num_pc, num_points, _ = point_clouds.shape
    outlier_pc = np.zeros_like(point_clouds)
    outlier_idx = np.zeros([num_pc, num_points], dtype=np.int16)
    outlier_num = np.zeros(num_pc, dtype=np.int16)
    inlier_pc = np.zeros_like(point_clouds)
    for l in range(num_pc):
        d = knn_dists[l]
        o_idx = np.where(d > knn_dist_thresh)[0]
        o_pts = point_clouds[l, o_idx, :]
        outlier_idx[l, :len(o_idx)] = o_idx
        outlier_num[l] = len(o_idx)
        outlier_pc[l, :len(o_idx)] = o_pts
        if 0 < len(o_idx) < num_points:
            outlier_pc[l, len(o_idx):] = o_pts[-1]
        i_idx = np.where(d <= knn_dist_thresh)[0]
        i_pts = point_clouds[l, i_idx, :]
        inlier_pc[l, :len(i_idx), :] = i_pts
        if 0 < len(i_idx) < num_points:
            inlier_pc[l, len(i_idx):, :] = i_pts[-1]
    return outlier_pc, outlier_idx, outlier_num, inlier_pc


def defend_surface(ae, adversarial_pc, source_pc, num_knn=8, top_k=2, knn_dist_thresh=0.04):
    """run_defense_surface.py:187-207 for one set of clouds: filter off-surface points, reconstruct
    the defended clouds with the victim AE, and score them against the sources.
    Returns dict(knn_dists, outlier_num, defended_pc, defended_recon, recon_error_vs_source)."""
    knn = get_knn_dists(adversarial_pc, num_knn, ae.device)
    score = knn[:, :, :top_k].mean(axis=2)
    _, o_idx, o_num, inlier = get_outlier_pc_inlier_pc(adversarial_pc, score, knn_dist_thresh)
    recon, _ = ae.forward(inlier)
    err = ae.loss_per_pc_tensor(recon, ae._as_dev(source_pc)).cpu().numpy()
    return dict(knn_dists=knn, outlier_idx=o_idx, outlier_num=o_num, defended_pc=inlier,
                defended_recon=recon.cpu().numpy(), recon_error_vs_source=err)


# ---------------------------------------------------------------------------------------------
# Critical-points defense (defender/run_defense_critical.py:180-196, src/ae_utils.py:12-80) -- SURVEY 8f-3
# ---------------------------------------------------------------------------------------------
def get_complementary_idx(idx, n):
    """src/general_utils.py:84-91."""
    comp = np.full(n, True)
    comp[idx] = False
    return np.arange(n, dtype=int)[comp]


def get_critical_points(point_clouds, max_val, max_idx):
    """src/ae_utils.py:12-48 with (max_val, max_idx) = (np.max, np.argmax)(pre_symmetry_data, axis=1) supplied by
    the fused encoder (PointNetAE.max_and_argmax) instead of the (num_pc, n, 128) tensor itself."""
    num_pc, bottleneck_size = max_val.shape
    critical_points = np.zeros([num_pc, bottleneck_size, 3], dtype=point_clouds.dtype)
    idx_critical = np.zeros([num_pc, bottleneck_size], dtype=np.int16)
    num_critical = np.zeros(num_pc, dtype=np.int16)
    for i in range(num_pc):
        max_idx_non_zero = max_idx[i][max_val[i] > 0.0]          # drop channels that are 0 for the whole cloud
        idx_critical_pc, counts = np.unique(max_idx_non_zero, return_counts=True)
        num_critical_pc = idx_critical_pc.shape[0]
        num_critical[i] = num_critical_pc
        idx_sort = np.argsort(counts)[::-1]                        # most critical points first
        idx_sorted = idx_critical_pc[idx_sort]
        critical_points[i, :num_critical_pc, :] = point_clouds[i][idx_sorted]
        idx_critical[i, :num_critical_pc] = idx_sorted
    return critical_points, idx_critical, num_critical


def get_critical_pc_non_critical_pc(point_clouds, max_val, max_idx):
    """src/ae_utils.py:51-80."""
    critical_points, critical_idx, critical_num = get_critical_points(point_clouds, max_val, max_idx)
    critical_pc = np.zeros_like(point_clouds)
    non_critical_pc = np.zeros_like(point_clouds)
    n = point_clouds.shape[1]
    for k in range(len(point_clouds)):
        idx_pc = critical_idx[k, :critical_num[k]]
        pts = point_clouds[k, idx_pc, :]
        critical_pc[k, :critical_num[k], :] = pts
        critical_pc[k, critical_num[k]:, :] = pts[-1]               # duplicated last point: same latent vector
        comp = get_complementary_idx(idx_pc, n)
        non_crit = point_clouds[k, comp, :]
        non_critical_pc[k, :len(non_crit)] = non_crit
        non_critical_pc[k, len(non_crit):] = non_crit[-1]
    return critical_points, critical_idx, critical_num, critical_pc, non_critical_pc


def defend_critical(ae, adversarial_pc, source_pc):
    """run_defense_critical.py:180-196 for one set of clouds: drop the critical points of each adversarial cloud,
    reconstruct what is left and score it against the source."""
    mv, mi = ae.max_and_argmax(adversarial_pc)
    mv, mi = mv.cpu().numpy(), mi.cpu().numpy()
    crit_pts, crit_idx, crit_num, pc_critical, pc_defended = get_critical_pc_non_critical_pc(adversarial_pc, mv, mi)
    recon, _ = ae.forward(pc_defended)
    err = ae.loss_per_pc_tensor(recon, ae._as_dev(source_pc)).cpu().numpy()
    return dict(critical_points=crit_pts, critical_idx=crit_idx, critical_num=crit_num, critical_pc=pc_critical,
                defended_pc=pc_defended, defended_recon=recon.cpu().numpy(), recon_error_vs_source=err)
