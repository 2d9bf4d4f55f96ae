"""Per-kernel-class time of one attack iteration (HIP events on the launch stream) for a given batch size.
    python tools/attack_breakdown.py [B ...]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geometric_adv_amd import weights as W
from geometric_adv_amd.adv_ae import AdvAE, Configuration
from geometric_adv_amd.autoencoder import PointNetAE
N = int(os.environ.get("GEOADV_TOOL_N", "2048"))
for B in [int(a) for a in sys.argv[1:]] or [1, 32]:
    rng = np.random.default_rng(B)
    x = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5); gt = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5)
    w = W.synthetic_weights(N, seed=7); ae = PointNetAE(w, N)
    at = AdvAE("a", Configuration(batch_size=B, n_points=N, weights=w, num_iterations=400, num_iterations_thresh=10**6), ae=ae)
    at.set_inputs(x, gt, ae.transform(gt), 1.0); at.init_pert(None, reset_optimizer=True)
    at.run(0, 20, 10**6)
    at.profile(True)
    at.run(20, 100, 10**6)
    torch.cuda.synchronize()
    br = {k: round(1e3 * ms / max(n_, 1), 2) for k, (n_, ms) in at.profile_read().items()}
    print(json.dumps({"batch": B, "us_per_iteration": br, "sum_us": round(sum(br.values()), 1)}))
