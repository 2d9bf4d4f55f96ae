"""Work counters of the grid k-NN search (diagnostic variant -DKG_DIAG): points visited, shells, drain steps per wave."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from geometric_adv_amd import ops, _lib
B, N = 64, 2048
rng = np.random.default_rng(3)
x = torch.as_tensor(rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5)).cuda()
ops.knn_grid_mode(os.environ.get("GEOADV_KNN_MODE", "grid"))
buf = (ctypes.c_ulonglong * 16)()
lib = _lib.lib()
ops.knn_dists(x, 8); torch.cuda.synchronize()
lib.geoadv_debug_knn_diag(buf, 1)
ops.knn_dists(x, 8); torch.cuda.synchronize()
lib.geoadv_debug_knn_diag(buf, 1)
v = list(buf)
waves = max(v[4], 1)
print(json.dumps({"waves": v[4], "points_per_wave": v[0] / waves, "frac_of_cloud": v[0] / waves / N, "shells_per_wave": v[1] / waves,
                  "drain_steps_per_wave": v[2] / waves, "drains_per_wave": v[5] / waves, "groups_of_4_per_wave": v[3] / waves,
                  "rows_per_wave": v[6] / waves, "whole_grid_waves": v[7], "lane_tasks": v[8], "lane_steps_per_task": v[9] / max(v[8], 1),
                  "leftover_lanes_per_task": v[10] / max(v[8], 1), "leftover_chunks": v[11]}))

a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
if hasattr(lib, "geoadv_debug_knn_occupancy"):
    lib.geoadv_debug_knn_occupancy(N, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    print(json.dumps({"occupancy_blocks_per_cu": {"grid_256": a.value, "grid_512": b.value, "all_points_256": c.value}}))
