"""Scratch GPU probe: VALU calibration + Chamfer kernel timing (writes gpurun_out/probe.json)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometric_adv_amd import ops  # noqa: F401
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe"))
import probe

out = {}
dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count)
names = ["mul+add", "pk_mul+pk_add", "min", "fma", "pk_fma"]
iters = 4000
for w, nm in enumerate(names):
    ms = probe.microbench(w, iters)
    inst = 2048 * 256 * 16 * iters
    out["valu_" + nm] = {"ms": ms, "Tinstr_lane_per_s": inst / ms / 1e9}
    print(nm, ms, "ms", inst / ms / 1e9, "T lane-instr/s")

for w, nm in zip(range(18, 40), ["add", "mul", "sub", "min_u32", "min3_f32", "min3_u32", "max_f32", "cndmask_vcc", "cmp_lt_f32", "mov",
                            "cndmask_sgpr_mask", "writelane", "fma", "lshl_add_u32+add_u32", "add_f64", "mul_f64", "fma_f64",
                            "cvt_f64_f32", "cvt_f32_f64", "exp_f32", "ldexp_f32", "rndne_f32"]):
    ms = min(probe.microbench(w, iters) for _ in range(2))
    inst = 2048 * 256 * 16 * iters
    cyc = ms * 1e-3 * 2.4e9 * 1024 / (2048 * 4 * 16 * iters)      # SIMD cycles per wave instruction at 2.4 GHz
    out["valu_" + nm] = {"ms": ms, "Tinstr_lane_per_s": inst / ms / 1e9, "cycles_per_wave_instr": cyc}
    print(nm, round(ms, 3), "ms", round(inst / ms / 1e9, 1), "T lane-instr/s", round(cyc, 2), "cycles/instr")

for (b, n) in [(32, 2048), (4, 2048), (32, 8192), (256, 2048)]:
    x = torch.rand((b, n, 3), device=dev) - 0.5
    y = torch.rand((b, n, 3), device=dev) - 0.5
    for _ in range(3):
        ops.nn_distance(x, y)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        ops.nn_distance(x, y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    pairs = 2.0 * b * n * n
    out["chamfer_b%d_n%d" % (b, n)] = {"ms": ms, "Gpairs_per_s": pairs / ms / 1e6}
    print("nn_distance b=%d n=%d: %.3f ms  %.1f Gpair/s" % (b, n, ms, pairs / ms / 1e6))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/probe.json", "w"), indent=1)
