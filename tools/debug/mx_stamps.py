"""In-kernel stamps of the matrix-screened scan as an operator (diagnostic build: bash tools/debug/build_variants.sh chamfer_sym.hip
stamps:-DGA_STAMPS, swapped in by tools/debug/ab_cmd.sh).   python tools/debug/mx_stamps.py B N"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from geometric_adv_amd import ops, _lib
B = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
x = torch.rand((B, N, 3), device="cuda") - 0.5
y = torch.rand((B, N, 3), device="cuda") - 0.5
for _ in range(20):
    ops.nn_distance_sym(x, y)
torch.cuda.synchronize()
NB = 1024
buf = (C.c_ulonglong * (8 * NB * 8))()
fn = getattr(_lib.lib(), "geoadv_debug_stamps_chamfer_sym", None)
if fn is None:
    raise SystemExit("this libgeoadv.so was not built with -DGA_STAMPS")
assert fn(buf) == 0
allst = np.frombuffer(buf, dtype=np.uint64).reshape(8, NB, 8).astype(np.int64)
cnt = allst[1]
calls = np.maximum(cnt[:, 3], 1) / 8.0
print(json.dumps({"uncertified_rows_per_workgroup_mean": float((cnt[:, 1] / calls)[cnt[:, 3] > 0].mean()) if (cnt[:, 3] > 0).any() else None,
                  "deferred_columns_per_workgroup_mean": float((cnt[:, 2] / calls)[cnt[:, 3] > 0].mean()) if (cnt[:, 3] > 0).any() else None}))
t = allst[0]
cyc = allst[2][:, 1]
ok = (cyc > 0) & (t[:, 6] > 0) & (t[:, 0] > 0)
print(json.dumps({"core_cycles_per_us_median_(s_memtime_over_s_memrealtime)": float(np.median(cyc[ok] / ((t[ok, 6] - t[ok, 0]) / 100.0)))}))
live = (t[:, 0] > 0) & (t[:, 7] > 0)
t = t[live]
u = (t - t[:, 0].min()) / 100.0
out = {"workgroups": int(len(t)), "first_start": float(u[:, 0].min()), "last_start": float(u[:, 0].max()), "first_end": float(u[:, 7].min()), "last_end": float(u[:, 7].max())}
prev = 0
for i in range(1, 8):
    if (t[:, i] > 0).all():
        out["%d->%d" % (prev, i)] = [round(float(np.percentile(u[:, i] - u[:, prev], q)), 2) for q in (10, 50, 90)]
        prev = i
print(json.dumps(out))
