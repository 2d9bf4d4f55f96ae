"""MFMA issue-rate probe: what fp32 MFMA rate does the box sustain with NO memory traffic, for the encoder's chain shape
(one accumulator, every MFMA dependent on the previous) and for independent accumulators?  (geoadv_microbench 5..9)
    python tools/mfma_probe.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_adv_amd import ops
names = {5: "32x32x2 f32, 1 accumulator (dependent chain)", 6: "32x32x2 f32, 2 accumulators", 7: "32x32x2 f32, 4 accumulators",
         8: "16x16x4 f32, 1 accumulator", 9: "16x16x4 f32, 4 accumulators"}
iters = 500
out = {}
for w, nm in names.items():
    ms = min(ops.microbench(w, iters) for _ in range(3))
    flop_per = 2 * 32 * 32 * 2 if w <= 7 else 2 * 16 * 16 * 4
    waves = 4096 * 4
    tf = waves * 16 * iters * flop_per / ms / 1e9
    out[nm] = {"ms": ms, "TFLOP_per_s": tf, "frac_of_157.3": tf / 157.3}
    print(nm, round(ms, 3), "ms", round(tf, 1), "TFLOP/s")
print(json.dumps(out))
