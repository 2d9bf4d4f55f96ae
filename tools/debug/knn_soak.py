"""One-off soak: knn_dists / knn_point, grid search against the all-points kernel on many random shapes and cloud kinds (bit for bit)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from geometric_adv_amd import ops
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    b = int(rng.integers(1, 9)); n = int(rng.integers(512, 4097)); k = int(rng.integers(1, 16))
    kind = trial % 6
    x = rng.random((b, n, 3), dtype=np.float32)
    if kind == 1: x *= rng.uniform(0.001, 100.0, size=(b, 1, 3)).astype(np.float32)
    if kind == 2: x[:, : n // 3] = x[:, : n // 3] * np.float32(0.01) + np.float32(0.5)
    if kind == 3:
        v = rng.standard_normal((b, n, 3)).astype(np.float32); x = (0.4 * v / np.linalg.norm(v, axis=2, keepdims=True)).astype(np.float32)
        x[:, :64] = rng.standard_normal((b, 64, 3)).astype(np.float32) * 3
    if kind == 4: x[:, 100:200] = x[:, 300:400]
    if kind == 5: x[:, :, 2] = 0.25
    t = torch.as_tensor(x).cuda()
    out = {}
    for mode in ("all_points", "grid"):
        ops.knn_grid_mode(mode)
        out[mode] = (ops.knn_dists(t, k),) + tuple(ops.knn_point(min(k + 1, 16), t, t))
    ops.knn_grid_mode("auto")
    ok = all(torch.equal(a, g) for a, g in zip(out["all_points"], out["grid"]))
    if not ok:
        bad += 1
        print("MISMATCH", trial, b, n, k, kind)
print("trials done, mismatches:", bad)
