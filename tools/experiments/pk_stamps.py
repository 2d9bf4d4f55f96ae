"""In-kernel timeline of chamfer_pk_kernel (diagnostic build -DPK_VARIANT=20, swapped in by tools/debug/ab_cmd.sh): per
workgroup s_memrealtime stamps (100 MHz) at entry / row loads issued / columns staged / sweep done / exit.
    python tools/debug/pk_stamps.py B"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from geometric_adv_amd import ops, _lib
B, N = int(sys.argv[1]), 2048
rng = np.random.default_rng(B)
x = torch.as_tensor(rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5)).cuda()
y = torch.as_tensor(rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5)).cuda()
for _ in range(20):
    ops.nn_distance_symmetric(x, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ops.nn_distance_symmetric(x, y)
torch.cuda.synchronize()
nwg = 4096
buf = (C.c_ulonglong * (8 * nwg))()
assert _lib.lib().geoadv_debug_pk_stamps(buf, 8 * nwg) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 8).astype(np.int64)
live = s[:, 4] > 0
s = s[live]
t0 = s[:, 0].min()
us = (s[:, :5] - t0) / 100.0
print(json.dumps({"batch": B, "workgroups": int(live.sum()),
                  "start_us_min_med_max": [round(float(v), 2) for v in (us[:, 0].min(), np.median(us[:, 0]), us[:, 0].max())],
                  "phase_us_median": {"entry->rows_requested": round(float(np.median(us[:, 1] - us[:, 0])), 2),
                                      "->columns_staged": round(float(np.median(us[:, 2] - us[:, 1])), 2),
                                      "->sweep_done": round(float(np.median(us[:, 3] - us[:, 2])), 2),
                                      "->exit": round(float(np.median(us[:, 4] - us[:, 3])), 2)},
                  "wg_total_us_med_max": [round(float(np.median(us[:, 4] - us[:, 0])), 2), round(float((us[:, 4] - us[:, 0]).max()), 2)],
                  "last_exit_us": round(float(us[:, 4].max()), 2)}))
