"""Where does the x3 encoder's launch end late?  Per-workgroup start / end from the stamps build, grouped by XCD (linear block id % 8)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from geometric_adv_amd import weights as W, _lib
from geometric_adv_amd.autoencoder import PointNetAE
B, n = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 2048
ae = PointNetAE(W.randomized_weights(n, seed=3), n, encoder_arith=os.environ.get("ARITH"))     # None: the library default
pc = torch.rand(B, n, 3, device="cuda") - 0.5
for _ in range(20):
    ae.forward(pc, want_recon=False)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 1024 * 8))()
assert _lib.lib().geoadv_debug_stamps_encoder_x3(buf) == 0
blocks = min(1024, B * (n // 128))
a = np.array(buf, dtype=np.uint64).reshape(8, 1024, 8)[0][:blocks].astype(np.float64) / 100.0
t0 = a[:, 0].min()
dur = a[:, 7] - a[:, 0]
print(json.dumps({"blocks": blocks, "start_spread_us": float(a[:, 0].max() - t0), "duration_us_percentiles": {p: round(float(np.percentile(dur, p)), 2) for p in (0, 10, 50, 90, 100)},
                  "end_us_percentiles": {p: round(float(np.percentile(a[:, 7] - t0, p)), 2) for p in (0, 10, 50, 90, 100)}}))
lin = np.arange(blocks)          # blockIdx.x + gridDim.x * blockIdx.y
for x in range(8):
    m = (lin % 8) == x
    print(json.dumps({"xcd": x, "workgroups": int(m.sum()), "duration_median": round(float(np.median(dur[m])), 2), "duration_max": round(float(dur[m].max()), 2),
                      "phase_medians": [round(float(np.median(a[m, i + 1] - a[m, i])), 2) for i in range(7)]}))
