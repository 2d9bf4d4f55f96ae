"""The latent-attack parity of tests/test_gpu_configs.py (configs[2]) under both encoder arithmetics: error against the fp64 model."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from geometric_adv_amd import weights as W
from geometric_adv_amd.adv_ae import AdvAE, Configuration
from conftest import cloud
from oracle.attack_model import AEModel, AttackModel
n, b = 2048, 64
w = W.synthetic_weights(n)
x, gt = cloud(31, b, n), cloud(32, b, n)
for arith in ("f32", "bf16x3"):
    conf = Configuration(batch_size=b, n_points=n, weights=w, loss_adv_type="latent", loss_dist_type="chamfer",
                         dist_weight_list=[150.0], num_iterations=30, num_iterations_thresh=25)
    from geometric_adv_amd.autoencoder import PointNetAE
    at = AdvAE("adversary", conf, ae=PointNetAE(w, n, encoder_arith=arith))
    tz = at.ae.transform(gt)
    ref = at.ae.get_loss_per_pc(gt)
    metrics, adv, recon = at.attack(x, tz, gt, ref, conf)
    h = at.last_history[0]
    s = {k: v.cpu().numpy() for k, v in at.peek().items()}
    sel = [0, 7, 30, 63]
    am = AttackModel(AEModel(W.canonical(w, n), n, np.float64), x[sel], gt[sel], tz[sel].astype(np.float64), 150.0 * np.ones(len(sel)), loss_adv_type="latent")
    am.pert = s["pert"][sel].astype(np.float64)
    f = am.forward()
    print(arith, "loss_adv rel err", np.abs(h[-1, 0][sel] / f["loss_adv"] - 1), "loss_adv", f["loss_adv"], "latent max abs err", np.abs(s["latent"][sel] - f["z"]).max())
