"""Per-class us of the all-pairs B = 32 loop with the screened scan forced in (loop rule) and off.   python tools/debug/mx_loop_classes.py [B]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from geometric_adv_amd import ops, weights as W
from geometric_adv_amd.adv_ae import AdvAE, Configuration
from geometric_adv_amd.autoencoder import PointNetAE
N = 2048; B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
w = W.synthetic_weights(N, seed=7); ae = PointNetAE(w, N)
rng = np.random.default_rng(B)
x = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5); gt = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5)
at = AdvAE("a", Configuration(batch_size=B, n_points=N, weights=w, num_iterations=3000, num_iterations_thresh=10**6, chamfer_prune=False), ae=ae)
at.set_inputs(x, gt, ae.transform(gt), 1.0); at.init_pert(None, reset_optimizer=True)
at.run(0, 200, 10**6); torch.cuda.synchronize()
for on in (True, False, True, False):
    ops.chamfer_screen(on)
    at.run(200, 50, 10**6); torch.cuda.synchronize()
    at.profile(True); at.run(250, 100, 10**6); torch.cuda.synchronize()
    br = {k: round(ms / max(c, 1) * 1e3, 2) for k, (c, ms) in at.profile_read().items()}
    at.profile(False)
    print(json.dumps({"screened": on, "us": br, "sum": round(sum(br.values()), 1)}))
