"""Block timeline of the grid k-NN search from in-kernel stamps (-DGA_STAMPS variant): when blocks start / end, phase medians."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from geometric_adv_amd import ops, _lib
B, N = 256, 2048
x = torch.as_tensor(np.random.default_rng(3).random((B, N, 3), dtype=np.float32) - np.float32(0.5)).cuda()
ops.knn_grid_mode("grid")
lib = _lib.lib()
for _ in range(2):
    ops.knn_dists(x, 8); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 1024 * 8))()      # (slot 0 = knn_grid_kernel; the first 1024 workgroups)
assert lib.geoadv_debug_stamps_grouping(buf) == 0
a = np.array(buf, dtype=np.uint64).reshape(8, 1024, 8)[0].astype(np.float64) / 100.0     # us (100 MHz clock)
t0 = a[:, 0].min()
start, end = a[:, 0] - t0, a[:, 7] - t0
print(json.dumps({"blocks": 1024, "first_start_us": float(start.min()), "last_start_us": float(start.max()), "last_end_us": float(end.max()),
                  "block_duration_us_median": float(np.median(end - start)), "block_duration_us_max": float((end - start).max()),
                  "phase_median_us": {"staging": float(np.median(a[:, 1] - a[:, 0])), "barrier": float(np.median(a[:, 2] - a[:, 1])),
                                      "tasks_of_wave_0": float(np.median(a[:, 7] - a[:, 2]))},
                  "starts_us_percentiles": [float(np.percentile(start, p)) for p in (10, 25, 50, 75, 90)]}))
