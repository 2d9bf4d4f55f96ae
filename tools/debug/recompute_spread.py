"""Masked against recomputing encoder backward (tests/test_gpu_attack.py::test_masked_backward_equals_recomputing_backward) over seeds and
arithmetics: per iteration, gradient difference / largest component and perturbation difference / largest."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from conftest import cloud
from geometric_adv_amd import weights as W
from geometric_adv_amd.autoencoder import PointNetAE
from geometric_adv_amd.adv_ae import AdvAE, Configuration
for n in (65, 257):
    w = W.randomized_weights(n)
    for arith in ("f16x2", "bf16x3", "f32"):
        ae = PointNetAE(w, n, encoder_arith=arith)
        for seed in (71, 73, 75):
            x, gt = cloud(seed, 3, n), cloud(seed + 1, 3, n)
            ats = []
            for rec in (False, True):
                at = AdvAE("a", Configuration(batch_size=3, n_points=n, weights=w, num_iterations=6, num_iterations_thresh=3, recompute_backward=rec), ae=ae)
                at.set_inputs(x, gt, None, 1.0); at.init_pert(None, reset_optimizer=True)
                ats.append(at)
            line = []
            for it in range(6):
                pk = []
                for at in ats:
                    at.run(it, 1, 3); pk.append({k: v.clone() for k, v in at.peek().items()})
                g0, g1 = pk[0]["grad"], pk[1]["grad"]
                dg = float(((g0 - g1).abs() / g1.abs().amax((1, 2), keepdim=True)).max())
                dp = float((pk[0]["pert"] - pk[1]["pert"]).abs().max() / pk[1]["pert"].abs().max())
                dz = float((pk[0]["latent"] - pk[1]["latent"]).abs().max())
                line.append("it%d g %.1e p %.1e z %.1e" % (it, dg, dp, dz))
            print(n, arith, seed, " | ".join(line), flush=True)
