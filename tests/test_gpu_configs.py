"""GPU: the BASELINE.json configurations as parity cases (config 1 is bench.py's workload)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_config0_trajectory_matches_model_golden():
    """configs[0]: single source/target pair, N=1024, 10 attack iterations -- every iteration's six
    metric vectors against the fp64 model trajectory (tests/golden/attack_trajectory.npz), plus the
    B=2 latent-space case.  Ten Adam steps from sigma=1e-7 noise stay within 1e-4 relative."""
    import torch
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd.adversary import init_pert_value
    g = np.load(os.path.join(GOLDEN, "attack_trajectory.npz"))
    for name in g["cases"]:
        n, hist_want = int(g[f"{name}_n"]), g[f"{name}_hist"]
        b = g[f"{name}_x"].shape[0]
        w = W.randomized_weights(n, seed=int(g[f"{name}_wseed"]))
        conf = Configuration(batch_size=b, n_points=n, weights=w, loss_adv_type=str(g[f"{name}_adv_type"]),
                             loss_dist_type=str(g[f"{name}_dist_type"]), num_iterations=10, num_iterations_thresh=8)
        at = AdvAE("adversary", conf)
        at.set_inputs(g[f"{name}_x"], g[f"{name}_gt"], g[f"{name}_tz"], float(g[f"{name}_dw"]))
        at.init_pert(init_pert_value(b, n), reset_optimizer=True)
        hist = torch.empty((10, 6, b), device="cuda:0")
        at.run(0, 10, 8, hist)
        got = hist.cpu().numpy()
        np.testing.assert_allclose(got, hist_want, rtol=1e-4, atol=1e-9, err_msg=str(name))
        s = at.peek()
        np.testing.assert_allclose(s["recon"].cpu().numpy(), g[f"{name}_recon"], atol=1e-5)
        # pert after ten steps of ~0.01: coordinates whose gradient is at the 1e-8 epsilon of Adam's
        # denominator are ill-conditioned (a 1e-7 change of g moves the step by 1e-4); a handful may differ
        dp = np.abs(s["pert"].cpu().numpy() - g[f"{name}_pert"])
        assert dp.max() < 5e-4 and (dp < 2e-5).mean() > 0.99


def test_config4_shape_n8192_lds_stress(oracle):
    """configs[4]: N = 8192 dense clouds (4 Chamfer LDS stages, 128-tile encoder grid, 24576-wide
    decoder, sorted-gradient fallback): forward vs model, exact indices, gradient vs model."""
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from oracle.attack_model import AEModel, AttackModel
    from conftest import cloud
    n, b = 8192, 2
    w = W.randomized_weights(n, seed=8)
    model = AEModel(W.canonical(w, n), n)
    x, gt = cloud(81, b, n), cloud(82, b, n)
    conf = Configuration(batch_size=b, n_points=n, weights=w, num_iterations=2, num_iterations_thresh=1)
    at = AdvAE("adversary", conf)
    at.set_inputs(x, gt, None, 1.0)
    p0 = (1e-3 * np.random.default_rng(1).standard_normal((b, n, 3))).astype(np.float32)
    at.init_pert(p0, reset_optimizer=True)
    s = {k: v.cpu().numpy() for k, v in at.peek().items()}
    am = AttackModel(model, x, gt, None, np.ones(b))
    am.init_pert(p0)
    f = am.forward()
    np.testing.assert_allclose(s["recon"], f["recon"], atol=2e-6)
    _, i1, _, i2 = oracle.nn_distance(s["recon"], gt)
    assert np.array_equal(s["idx_r1"], i1) and np.array_equal(s["idx_r2"], i2)
    _, i1, _, i2 = oracle.nn_distance(s["adv"], x)
    assert np.array_equal(s["idx_a1"], i1) and np.array_equal(s["idx_a2"], i2)
    g = am.gradient(am.forward(idx_override=(s["idx_r1"], s["idx_r2"], s["idx_a1"], s["idx_a2"])))
    at.run(0, 1, 1)
    got = at.peek()["grad"].cpu().numpy()
    sc = np.abs(g).reshape(b, -1).max(1)[:, None, None]
    np.testing.assert_allclose(got / sc, g / sc, atol=1e-4)


def test_config2_latent_attack_then_knn_defense_b256():
    """configs[2]: B = 256, N = 2048, latent-space attack (weight 150) followed by the k-NN
    off-surface defense.  Full size, so properties: the latent loss decreases; adv stays a valid
    output of the loop (adv == source + pert); defended clouds keep only points whose 2-NN mean is
    within the threshold; reconstructing the defended cloud is invariant to the padding."""
    import torch
    from geometric_adv_amd import ops, weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd.defense import defend_surface
    from conftest import cloud
    n, b = 2048, 256
    w = W.synthetic_weights(n)
    conf = Configuration(batch_size=b, n_points=n, weights=w, loss_adv_type="latent", loss_dist_type="chamfer",
                         dist_weight_list=[150.0], num_iterations=30, num_iterations_thresh=25)
    at = AdvAE("adversary", conf)
    x, gt = cloud(31, b, n), cloud(32, b, n)
    tz = at.ae.transform(gt)
    ref = at.ae.get_loss_per_pc(gt)
    metrics, adv, recon = at.attack(x, tz, gt, ref, conf)
    h = at.last_history[0]
    assert h[-1, 0].mean() < h[0, 0].mean()                      # ||z - z_target|| went down
    assert metrics.shape == (1, b, 5) and np.isfinite(metrics).all()
    out = defend_surface(at.ae, adv[0], x, num_knn=8, top_k=2, knn_dist_thresh=0.04)
    knn = out["knn_dists"]
    assert knn.shape == (b, n, 8) and (np.diff(knn, axis=2) >= 0).all()
    score = knn[:, :, :2].mean(2)
    for j in (0, 100, 255):
        keep = score[j] <= 0.04
        assert out["outlier_num"][j] == (~keep).sum()
        assert np.array_equal(out["defended_pc"][j, :keep.sum()], adv[0, j][keep])
    assert out["recon_error_vs_source"].shape == (b,) and np.isfinite(out["recon_error_vs_source"]).all()
