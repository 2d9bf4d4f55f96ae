"""Diagnostic: where one encoder-forward tile spends its cycles (s_memtime stamps at the phase boundaries)."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometric_adv_amd import _lib, weights as W
from geometric_adv_amd.autoencoder import PointNetAE

B, N = 32, 2048
ae = PointNetAE(W.synthetic_weights(N), N)
x = torch.rand((B, N, 3), device="cuda:0") - 0.5
ae.forward(x)
tiles = N // 64
st = torch.zeros((B, tiles, 12), dtype=torch.int64, device="cuda:0")
lib = _lib.lib()
for _ in range(3):
    _lib.check(lib.geoadv_debug_encoder_stamps(ae.handle, B, _lib.ptr(x), _lib.ptr(ae._ws), _lib.ptr(st), _lib.stream_handle()), "stamps")
torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.int64).reshape(-1, 12)[:, :9]
d = np.diff(s, axis=1)
names = ["points+L0", "L1", "L2", "L3a", "L4a", "L3b", "L4b", "pool"]
med = np.median(d, axis=0)
tot = np.median(s[:, 8] - s[:, 0])
# MFMAs per wave per phase; 4 waves share a SIMD when 2 workgroups are resident
mfma = [0, 32, 64, 64, 64, 64, 64, 0]
out = {"total_cycles_median": float(tot), "phases": {}}
for nme, m, k in zip(names, med, mfma):
    out["phases"][nme] = {"cycles": float(m), "mfma_cycles_own_wg": k * 64 * 2}
    print("%-10s %8.0f cycles   (this WG's MFMA work on one SIMD: %5d cycles)" % (nme, m, k * 64 * 2))
print("total %.0f cycles; start spread %.0f" % (tot, s[:, 0].max() - s[:, 0].min()))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/encoder_phases.json", "w"), indent=1)
