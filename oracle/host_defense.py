"""TEST INFRASTRUCTURE -- numpy restatement of the two defenses' per-cloud bookkeeping, the checker of the device kernels in
geometric_adv_amd/csrc/defense.hip (only tests/ and bench.py's checks import this; the product never does).

  outlier_inlier          src/adversary_utils.py:149-178  get_outlier_pc_inlier_pc
  critical_points         src/ae_utils.py:12-48           get_critical_points
  critical_and_rest       src/ae_utils.py:51-80           get_critical_pc_non_critical_pc (+ src/general_utils.py:84-91)

Pinned by tests/golden/host_logic.npz, which oracle/make_golden_host.py produced by exec'ing the reference's own function
bodies (tests/test_host_golden.py).  One freedom is left open by the reference itself: ae_utils.py:34 orders the critical
points with `np.argsort(counts)[::-1]`, numpy's default sort is not stable (the AVX-512 builds reorder ties even among five
elements), so points owning equally many channels come in a build-dependent order.  `critical_points` here uses the stable
sort (count descending, then point index descending) -- the order the device kernel defines -- and `same_critical_sets`
compares two results up to the order inside a group of equal counts.
"""
import numpy as np


def _pack(cloud, keep):
    """Rows of `cloud` with keep, in order, padded to the cloud's length with the last kept row (zeros if none)."""
    out = np.zeros_like(cloud)
    rows = cloud[keep]
    out[:len(rows)] = rows
    if 0 < len(rows) < len(cloud):
        out[len(rows):] = rows[-1]
    return out


def outlier_inlier(point_clouds, score, thresh):
    """-> outlier_pc, outlier_idx (int16), outlier_num (int16), inlier_pc.  score (num_pc, n): the per-point scalar."""
    num_pc, n, _ = point_clouds.shape
    outlier_pc, inlier_pc = np.zeros_like(point_clouds), np.zeros_like(point_clouds)
    outlier_idx = np.zeros((num_pc, n), np.int16)
    outlier_num = np.zeros(num_pc, np.int16)
    for c in range(num_pc):
        out = score[c] > thresh
        where = np.flatnonzero(out)
        outlier_idx[c, :len(where)] = where
        outlier_num[c] = len(where)
        outlier_pc[c] = _pack(point_clouds[c], out)
        inlier_pc[c] = _pack(point_clouds[c], score[c] <= thresh)         # (a NaN score is in neither set)
    return outlier_pc, outlier_idx, outlier_num, inlier_pc


def critical_points(point_clouds, max_val, max_idx, kind="stable"):
    """-> critical_points (num_pc, c, 3), idx_critical (int16), num_critical (int16), counts (num_pc, c) per listed point.
    kind: the sort behind np.argsort(counts)[::-1]; None = numpy's default (what the reference runs)."""
    num_pc, c = max_val.shape
    pts = np.zeros((num_pc, c, 3), point_clouds.dtype)
    idx = np.zeros((num_pc, c), np.int16)
    num = np.zeros(num_pc, np.int16)
    cnt = np.zeros((num_pc, c), np.int64)
    for i in range(num_pc):
        owners, counts = np.unique(max_idx[i][max_val[i] > 0.0], return_counts=True)
        order = (np.argsort(counts, kind=kind) if kind else np.argsort(counts))[::-1]
        k = len(owners)
        num[i] = k
        idx[i, :k] = owners[order]
        cnt[i, :k] = counts[order]
        pts[i, :k] = point_clouds[i][owners[order]]
    return pts, idx, num, cnt


def critical_and_rest(point_clouds, max_val, max_idx, kind="stable"):
    """-> critical_points, critical_idx, critical_num, critical_pc, non_critical_pc."""
    pts, idx, num, _ = critical_points(point_clouds, max_val, max_idx, kind)
    n = point_clouds.shape[1]
    crit_pc, rest_pc = np.zeros_like(point_clouds), np.zeros_like(point_clouds)
    for i in range(len(point_clouds)):
        own = idx[i, :num[i]].astype(np.int64)
        if num[i]:
            crit_pc[i, :num[i]] = point_clouds[i][own]
            crit_pc[i, num[i]:] = point_clouds[i][own[-1]]
        rest = np.ones(n, bool)
        rest[own] = False
        rest_pc[i] = _pack(point_clouds[i], rest)
    return pts, idx, num, crit_pc, rest_pc


def same_critical_sets(idx_a, num_a, idx_b, num_b, max_val, max_idx):
    """True if two (critical_idx, critical_num) results list the same points with non-increasing channel counts and differ at
    most in the order inside a group of equal counts."""
    if not np.array_equal(num_a, num_b):
        return False
    for i in range(len(num_a)):
        owners, counts = np.unique(max_idx[i][max_val[i] > 0.0], return_counts=True)
        count_of = dict(zip(owners.tolist(), counts.tolist()))
        for idx in (idx_a, idx_b):
            row = idx[i, :num_a[i]].astype(np.int64).tolist()
            if sorted(row) != sorted(owners.tolist()):
                return False
            c = [count_of[p] for p in row]
            if any(c[j] < c[j + 1] for j in range(len(c) - 1)):
                return False
    return True
