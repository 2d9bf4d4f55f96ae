import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from geometric_adv_amd import ops, weights as W
from geometric_adv_amd.adv_ae import AdvAE, Configuration
from geometric_adv_amd.autoencoder import PointNetAE
B, N = 32, 2048
ops.chamfer_screen(os.environ.get("SCREEN", "1") == "1")
w = W.synthetic_weights(N, seed=7); ae = PointNetAE(w, N)
rng = np.random.default_rng(0)
x = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5); gt = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5)
at = AdvAE("a", Configuration(batch_size=B, n_points=N, weights=w, num_iterations=400, num_iterations_thresh=10**6, chamfer_prune=False), ae=ae)
at.set_inputs(x, gt, ae.transform(gt), 1.0); at.init_pert(None, reset_optimizer=True)
at.run(0, 300, 10**6); torch.cuda.synchronize()
