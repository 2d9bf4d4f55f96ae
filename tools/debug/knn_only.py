"""knn_dists at config 2's shape with both kernels (for rocprofv3 passes)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from geometric_adv_amd import ops
B, N = 256, 2048
x = torch.as_tensor(np.random.default_rng(3).random((B, N, 3), dtype=np.float32) - np.float32(0.5)).cuda()
out = {}
for mode in (("all_points", "grid") if not os.environ.get("GEOADV_KNN_MODE") else (os.environ["GEOADV_KNN_MODE"],)):
    ops.knn_grid_mode(mode)
    ops.knn_dists(x, 8); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): ops.knn_dists(x, 8)
    for _ in range(5): ops.knn_point(8, x, x)
    for _ in range(5): ops.knn_point(9, x, x)
    torch.cuda.synchronize(); out[mode] = (time.perf_counter() - t) / 5 * 1e3
print(json.dumps(out))
