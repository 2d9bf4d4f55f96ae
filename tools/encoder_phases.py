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
for _ in range(300):      # ~30 ms of back-to-back launches so that the clock settles under load
    _lib.check(lib.geoadv_debug_encoder_stamps(ae.handle, B, _lib.ptr(x), _lib.ptr(ae._ws), _lib.ptr(st), _lib.stream_handle()), "stamps")
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    _lib.check(lib.geoadv_debug_encoder_stamps(ae.handle, B, _lib.ptr(x), _lib.ptr(ae._ws), _lib.ptr(st), _lib.stream_handle()), "stamps")
e1.record(); torch.cuda.synchronize()
print("stamped kernel back to back: %.1f us per launch" % (e0.elapsed_time(e1) * 1000 / 200))
raw = st.cpu().numpy().astype(np.int64).reshape(-1, 12)
s = raw[:, :9]
clk = (raw[:, 8] - raw[:, 0]) / np.maximum(raw[:, 10] - raw[:, 9], 1) * 100e6 / 1e9
print("in-kernel shader clock (s_memtime / s_memrealtime x 100 MHz): median %.3f GHz (min %.3f, max %.3f)" % (np.median(clk), clk.min(), clk.max()))
d = np.diff(s, axis=1)
names = ["points+L0", "L1", "L2", "L3a", "L4a", "L3b", "L4b", "pool"]
med = np.median(d, axis=0)
tot = np.median(s[:, 8] - s[:, 0])
# MFMAs per wave per phase; 4 waves share a SIMD when 2 workgroups are resident
mfma = [0, 32, 64, 64, 64, 64, 64, 0]
out = {"total_cycles_median": float(tot), "in_kernel_clock_GHz_median": float(np.median(clk)), "phases": {}}
for nme, m, k in zip(names, med, mfma):
    out["phases"][nme] = {"cycles": float(m), "mfma_cycles_own_wg": k * 64 * 2}
    print("%-10s %8.0f cycles   (this WG's MFMA work on one SIMD: %5d cycles)" % (nme, m, k * 64 * 2))
st0 = s[:, 0] - s[:, 0].min()
en = s[:, 8] - s[:, 0].min()
order = np.argsort(st0)
first, second = order[:len(order) // 2], order[len(order) // 2:]
print("total %.0f cycles per tile" % tot)
print("kernel span %.0f cycles = %.1f us at the in-kernel clock" % (en.max(), en.max() / np.median(clk) / 1e3))
print("first-round tiles : start %6.0f .. %6.0f   end %6.0f .. %6.0f" % (st0[first].min(), st0[first].max(), en[first].min(), en[first].max()))
print("second-round tiles: start %6.0f .. %6.0f   end %6.0f .. %6.0f" % (st0[second].min(), st0[second].max(), en[second].min(), en[second].max()))
print("tile duration: min %.0f median %.0f max %.0f" % ((s[:, 8] - s[:, 0]).min(), tot, (s[:, 8] - s[:, 0]).max()))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/encoder_phases.json", "w"), indent=1)
