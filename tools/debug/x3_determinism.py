"""The x3 forward run many times on the same input: every run must give the bits of the first (LDS-DMA ring, counted waits,
exchanges -- a race would show as a flipped bit sooner or later).  All three forms (B = 3, 8, 40) with and without masks."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from geometric_adv_amd import weights as W
from geometric_adv_amd.autoencoder import PointNetAE
from geometric_adv_amd.adv_ae import AdvAE, Configuration
from conftest import cloud
n = 2048
w = W.randomized_weights(n, seed=3)
ae = PointNetAE(w, n)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for B in (3, 8, 40):
    pc = torch.as_tensor(cloud(7, B, n)).cuda()
    z0, i0 = ae.max_and_argmax(pc)
    z0, i0 = torch.as_tensor(z0).clone(), torch.as_tensor(i0).clone()
    bad = 0
    for _ in range(reps):
        z, i = ae.max_and_argmax(pc)
        bad += int(not (torch.equal(torch.as_tensor(z), z0) and torch.equal(torch.as_tensor(i), i0)))
    # with masks: the attack's forward (perturbation fixed by a zero learning rate)
    at = AdvAE("a", Configuration(batch_size=B, n_points=n, weights=w, num_iterations=10, num_iterations_thresh=10 ** 6, learning_rate=0.0), ae=ae)
    x, gt = cloud(8, B, n), cloud(9, B, n)
    at.set_inputs(x, gt, None, 1.0)
    at.init_pert((1e-3 * np.random.default_rng(1).standard_normal((B, n, 3))).astype(np.float32), reset_optimizer=True)
    at.run(0, 2, 10 ** 6)
    ref = {k: v.clone() for k, v in at.peek().items()}
    badm = 0
    for r in range(reps // 3):
        at.run(2 + 3 * r, 3, 10 ** 6)
        cur = at.peek()
        badm += int(not all(torch.equal(cur[k], ref[k]) for k in ("latent", "recon", "grad", "idx_r1", "idx_a1")))
    print("B", B, "forward runs differing from the first:", bad, "of", reps, "| attack states differing (lr 0):", badm, "of", reps // 3, flush=True)
