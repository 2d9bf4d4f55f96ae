"""Timeline of ONE attack iteration from in-kernel stamps (diagnostic build: bash tools/debug/build_variants.sh all stamps:-DGA_STAMPS,
swapped in by tools/debug/ab_cmd.sh).  Per kernel: when its first / last workgroup started and ended (us, 100 MHz s_memrealtime,
relative to the first stamp of the iteration) and the median time between its phase stamps.
    python tools/debug/iter_timeline.py B [N]          (GEOADV_TOOL_CFG='{"chamfer_kernel": "symmetric"}' overrides Configuration fields)"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from geometric_adv_amd import weights as W, _lib
from geometric_adv_amd.adv_ae import AdvAE, Configuration
from geometric_adv_amd.autoencoder import PointNetAE
B = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
rng = np.random.default_rng(B)
x = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5); gt = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5)
w = W.synthetic_weights(N, seed=7); ae = PointNetAE(w, N)
at = AdvAE("a", Configuration(batch_size=B, n_points=N, weights=w, num_iterations=400, num_iterations_thresh=10**6, **json.loads(os.environ.get("GEOADV_TOOL_CFG", "{}"))), ae=ae)
at.set_inputs(x, gt, ae.transform(gt), 1.0); at.init_pert(None, reset_optimizer=True)
at.run(0, 60, 10**6)
torch.cuda.synchronize()
NB = 1024
names = {"decoder": {0: "latent_decode", 1: "grid_search blocks (in latent_decode)", 2: "decoder_fc2", 3: "decoder_fc2_bwd", 4: "decoder_bwd_tail"},
         "encoder": {0: "encoder_fwd", 1: "encoder_bwd blocks (masked / dense)", 2: "encoder_jac", 4: "decoder_bwd_tail blocks (in tail+dense)"}, "chamfer": {0: "chamfer_scan"},
         "encoder_x3": {0: "encoder_fwd3 (x3: constants | layer 0 | layer 1 | layer 2 | layers 3a + 4a | layers 3b + 4b | pool + masks)", 1: "encoder_fwd3s (x3 split form)"},
         "chamfer_sym": {0: "chamfer_sym", 1: "grid_search blocks (in chamfer_sym)", 2: "chamfer_sym_finish", 3: "pool Jacobian blocks (in chamfer_sym)", 4: "loss + gradient riders (in chamfer_sym): entry | released | end"}, "attack": {0: "loss_cgrad (all blocks)", 1: "loss_cgrad: gradient blocks"}}
rows = []
for tu, slots in names.items():
    buf = (C.c_ulonglong * (8 * NB * 8))()
    fn = getattr(_lib.lib(), "geoadv_debug_stamps_" + tu, None)
    if fn is None:
        raise SystemExit("this libgeoadv.so was not built with -DGA_STAMPS")
    assert fn(buf) == 0
    s = np.frombuffer(buf, dtype=np.uint64).reshape(8, NB, 8).astype(np.int64)
    for k, nm in slots.items():
        t = s[k]
        live = (t[:, 0] > 0) & (t[:, 7] > 0)
        if live.any():
            rows.append((nm, t[live]))
t0 = min(r[1][:, 0].min() for r in rows)
rows.sort(key=lambda r: r[1][:, 0].min())
for nm, t in rows:
    u = (t - t0) / 100.0
    ph = {}
    prev = 0
    for i in range(1, 8):
        ok = t[:, i] > 0
        if ok.all():
            ph["%d->%d" % (prev, i)] = round(float(np.median(u[:, i] - u[:, prev])), 2)
            prev = i
    print(json.dumps({"kernel": nm, "workgroups": int(len(t)), "first_start": round(float(u[:, 0].min()), 2), "last_start": round(float(u[:, 0].max()), 2),
                      "first_end": round(float(u[:, 7].min()), 2), "last_end": round(float(u[:, 7].max()), 2), "phases_us_median": ph}))
