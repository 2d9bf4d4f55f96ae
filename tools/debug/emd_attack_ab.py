"""Chamfer + EMD attack iteration at B = 128 with the sparse EMD levels on / off, same process, interleaved."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from geometric_adv_amd import weights as W, ops
from geometric_adv_amd.adv_ae import AdvAE, Configuration
from geometric_adv_amd.autoencoder import PointNetAE
N, B = 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.default_rng(B)
x = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5); gt = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5)
w = W.synthetic_weights(N, seed=7); ae = PointNetAE(w, N)
at = AdvAE("a", Configuration(batch_size=B, n_points=N, weights=w, num_iterations=400, num_iterations_thresh=10**6, emd_weight=1.0), ae=ae)
at.set_inputs(x, gt, ae.transform(gt), 1.0); at.init_pert(None, reset_optimizer=True)
at.run(0, 3, 10**6); torch.cuda.synchronize()
it = 3
for rep in range(3):
    for sparse in (False, True):
        ops.emd_sparse_levels(sparse)
        at.run(it, 2, 10**6); it += 2; torch.cuda.synchronize()
        t0 = time.perf_counter(); at.run(it, 10, 10**6); it += 10; torch.cuda.synchronize()
        print(json.dumps({"sparse": sparse, "ms_per_iteration": (time.perf_counter() - t0) / 10 * 1e3}))
p = at.peek()
r = p["recon"]
print("recon extent", (r.amax(dim=(0, 1)) - r.amin(dim=(0, 1))).tolist(), "std", r.std(dim=(0, 1)).tolist())
