"""Timing of the kNN grouping operators and the two defenses at BASELINE config 3's shape (B = 256, N = 2048).
    python tools/defense_time.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geometric_adv_amd import ops, defense, weights as W
from geometric_adv_amd.autoencoder import PointNetAE
B, N = 256, 2048
rng = np.random.default_rng(3)
x = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5)
xs = torch.as_tensor(x).cuda()


def timed(f, reps=5):
    """ms per call: median of three windows of `reps` calls after one untimed call."""
    f(); torch.cuda.synchronize()
    w = []
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize(); w.append((time.perf_counter() - t) / reps * 1e3)
    return sorted(w)[1]


out = {"batch": B, "n_points": N}
for mode in ("all_points", "grid", "grid_shells"):
    ops.knn_grid_mode(mode)
    out["knn_dists_k8_ms_" + mode] = timed(lambda: ops.knn_dists(xs, 8))
    out["knn_point_k8_ms_" + mode] = timed(lambda: ops.knn_point(8, xs, xs))
    out["knn_point_k9_ms_" + mode] = timed(lambda: ops.knn_point(9, xs, xs))
ops.knn_grid_mode("auto")
v = rng.standard_normal((B, N, 3)).astype(np.float32)
sh = torch.as_tensor((0.4 * v / np.linalg.norm(v, axis=2, keepdims=True)).astype(np.float32)).cuda()
sh[:, :100] = torch.as_tensor(x[:, :100]).cuda()          # a sphere shell with 5 % uniform outliers
out["knn_dists_k8_ms_shell_with_outliers"] = timed(lambda: ops.knn_dists(sh, 8))
out["query_ball_point_r0.1_ns32_ms"] = timed(lambda: ops.query_ball_point(0.1, 32, xs, xs))
idx = ops.knn_point(8, xs, xs)[1]
out["group_point_k8_ms"] = timed(lambda: ops.group_point(xs, idx))
w = W.synthetic_weights(N, seed=7); ae = PointNetAE(w, N)
out["defend_surface_device_ms"] = timed(lambda: defense.defend_surface_device(ae, xs, xs), reps=5)      # GPU tensors in and out
out["defend_critical_device_ms"] = timed(lambda: defense.defend_critical_device(ae, xs, xs), reps=5)
kn = ops.knn_dists(xs, 8)
out["outlier_filter_ms"] = timed(lambda: ops.outlier_filter(xs, kn, 0.04, top_k=2))
mv, mi = ae.max_and_argmax(xs)
out["critical_split_ms"] = timed(lambda: ops.critical_split(xs, mv, mi))
out["defend_surface_numpy_in_out_ms"] = timed(lambda: defense.defend_surface(ae, x, x), reps=2)        # + upload, + 6 result downloads
out["defend_critical_numpy_in_out_ms"] = timed(lambda: defense.defend_critical(ae, x, x), reps=2)
out["pairs_G"] = B * N * N / 1e9
print(json.dumps(out))
