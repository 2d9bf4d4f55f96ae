"""One-off soak: 3000 attack iterations at B = 32 x 2048 in one call; status() raises if a hand-off timed out."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from geometric_adv_amd import weights as W
from geometric_adv_amd.adv_ae import AdvAE, Configuration
from geometric_adv_amd.autoencoder import PointNetAE
N, B = 2048, 32
w = W.synthetic_weights(N, seed=7)
ae = PointNetAE(w, N, device="cuda:0")
rng = np.random.default_rng(5)
x = rng.random((B, N, 3), dtype=np.float32) - 0.5; gt = rng.random((B, N, 3), dtype=np.float32) - 0.5
at = AdvAE("adversary", Configuration(batch_size=B, n_points=N, weights=w, num_iterations=3000, num_iterations_thresh=2400), device="cuda:0", ae=ae)
at.set_inputs(x, gt, ae.transform(gt), 1.0)
at.init_pert(None, reset_optimizer=True)
t = time.time()
at.run(0, 3000, 2400)
torch.cuda.synchronize()
print("3000 iterations in %.3f s; status:" % (time.time() - t), at.status())
m = at.get_best(np.ones(B, dtype=np.float32))
print("best metrics finite:", bool(torch.isfinite(m[0]).all()))
