"""MFMA issue-rate probe: what fp32 MFMA rate does the box sustain with NO memory traffic, for the encoder's chain shape
(one accumulator, every MFMA dependent on the previous) and for independent accumulators?  (geoadv_probe_microbench 5..9)
    python tools/mfma_probe.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_adv_amd import ops  # noqa: F401
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe"))
import probe
names = {5: "32x32x2 f32, 1 accumulator (dependent chain)", 6: "32x32x2 f32, 2 accumulators", 7: "32x32x2 f32, 4 accumulators",
         8: "16x16x4 f32, 1 accumulator", 9: "16x16x4 f32, 4 accumulators",
         10: "encoder loop shape: 4 MFMA + 1 ds_read_b128 per k-group", 11: "encoder loop shape: 4 MFMA + 1 KiB global ring refill per k-group",
         12: "encoder loop shape: 4 MFMA + LDS read + global refill", 13: "as 11 with 16 KB of weights (L1 hits)",
         14: "as 11 with 4 MB of weights", 15: "4 dependent MFMA + 4 independent v_add_f32", 16: "4 dependent MFMA + 8 v_add_f32",
         17: "4 dependent MFMA + 16 v_add_f32"}
iters = 500
out = {}
for w, nm in names.items():
    ms = min(probe.microbench(w, iters) for _ in range(3))
    flop_per = 2 * 16 * 16 * 4 if w in (8, 9) else 2 * 32 * 32 * 2
    waves = 4096 * 4 if (w <= 9 or w >= 15) else 1024 * 8
    tf = waves * 16 * iters * flop_per / ms / 1e9
    out[nm] = {"ms": ms, "TFLOP_per_s": tf, "frac_of_157.3": tf / 157.3}
    print(nm, round(ms, 3), "ms", round(tf, 1), "TFLOP/s")
print(json.dumps(out))
