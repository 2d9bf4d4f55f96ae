"""Phase timeline of the x3 encoder's split form from in-kernel stamps (-DGA_STAMPS variant of encoder_x3.hip)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from geometric_adv_amd import weights as W, _lib
from geometric_adv_amd.autoencoder import PointNetAE
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = 2048
ae = PointNetAE(W.randomized_weights(n, seed=3), n)
pc = torch.rand(B, n, 3, device="cuda") - 0.5
for _ in range(5):
    ae.forward(pc, want_recon=False)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 1024 * 8))()
assert _lib.lib().geoadv_debug_stamps_encoder_x3(buf) == 0
blocks = min(1024, B * (n // 32))
a = np.array(buf, dtype=np.uint64).reshape(8, 1024, 8)[1][:blocks].astype(np.float64) / 100.0     # us (100 MHz clock)
t0 = a[:, 0].min()
names = ["loads+consts+barrier", "layer0+layer1", "exchange1", "layer2", "exchange2", "layers3a..4b (2 exchanges)", "pool+masks"]
print(json.dumps({"B": B, "blocks": blocks, "first_start_us": 0.0, "last_start_us": float((a[:, 0] - t0).max()), "last_end_us": float((a[:, 7] - t0).max()),
                  "block_us_median": float(np.median(a[:, 7] - a[:, 0])),
                  "phase_median_us": {names[i]: round(float(np.median(a[:, i + 1] - a[:, i])), 2) for i in range(7)}}))
