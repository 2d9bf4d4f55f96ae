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
    # parity on sampled clouds of the full batch, from the GPU's own final state: source-distance indices exact (pinned oracle), the
    # latent loss and the distance loss of the last iteration 1e-5 against the fp64 model evaluated at the GPU's perturbation
    from oracle.attack_model import AEModel, AttackModel
    from oracle.cpu_oracle import Oracle
    s = {k: v.cpu().numpy() for k, v in at.peek().items()}
    sel = [0, 77, 130, 255]
    _, j1, _, j2 = Oracle().nn_distance(s["adv"][sel], x[sel])
    assert np.array_equal(s["idx_a1"][sel], j1) and np.array_equal(s["idx_a2"][sel], j2)
    am = AttackModel(AEModel(W.canonical(w, n), n, np.float64), x[sel], gt[sel], tz[sel].astype(np.float64), 150.0 * np.ones(len(sel)),
                     loss_adv_type="latent")
    am.pert = s["pert"][sel].astype(np.float64)
    f = am.forward()
    # the latent loss is a sum of squared DIFFERENCES of latents (|z - t| ~ 0.005 against |z| ~ 0.2): the latent's own fp32
    # rounding (2e-7 absolute, checked two lines down at 2e-6) is amplified ~40 x in relative terms -- 2e-6 ... 1e-5 under either
    # encoder arithmetic (tools/debug/x3_cfg2.py), so 3e-5 here; the Chamfer loss keeps the path's 1e-5
    np.testing.assert_allclose(h[-1, 0][sel], f["loss_adv"], rtol=3e-5)
    np.testing.assert_allclose(h[-1, 1][sel], f["loss_dist"], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(s["latent"][sel], f["z"], atol=2e-6)
    out = defend_surface(at.ae, adv[0], x, num_knn=8, top_k=2, knn_dist_thresh=0.04)
    knn = out["knn_dists"]
    assert knn.shape == (b, n, 8) and (np.diff(knn, axis=2) >= 0).all()
    score = knn[:, :, :2].mean(2)
    for j in (0, 100, 255):
        keep = score[j] <= 0.04
        assert out["outlier_num"][j] == (~keep).sum()
        assert np.array_equal(out["defended_pc"][j, :keep.sum()], adv[0, j][keep])
    assert out["recon_error_vs_source"].shape == (b,) and np.isfinite(out["recon_error_vs_source"]).all()


@pytest.mark.parametrize("prune", [True, False])
def test_config1_full_shape_b32_n2048(oracle, prune):
    """configs[1] at its OWN shape (B = 32, N = 2048, chamfer/chamfer -- bench.py's workload: 1024 encoder tiles in two rounds,
    8 row tiles in the symmetric Chamfer kernel, the 2048-point grid-search instantiation riding in the latent_decode launch):
    three iterations, each checked from the GPU's own state -- reconstruction vs the fp64 model 2e-6, all four nearest-neighbour
    index arrays np.array_equal to the pinned oracle on the GPU's own clouds, the six metric rows 1e-5 relative -- once with the
    paired grid search and once with the all-pairs kernel for nn_distance(adv, x)."""
    import torch
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd.autoencoder import PointNetAE
    from oracle.attack_model import AEModel, AttackModel
    from conftest import cloud
    b, n = 32, 2048
    w = W.synthetic_weights(n, seed=7)                    # bench.py's weights and seeds
    ae = PointNetAE(w, n)
    model = AEModel(W.canonical(w, n), n, np.float64)
    x, gt = cloud(1002, b, n), cloud(2002, b, n)
    at = AdvAE("adversary", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=4, num_iterations_thresh=2,
                                          chamfer_prune=prune), ae=ae)
    at.set_inputs(x, gt, None, 1.0)
    at.init_pert(None, reset_optimizer=True)
    am = AttackModel(model, x, gt, None, np.ones(b))
    hist = torch.empty((1, 6, b), device=ae.device)
    for it in range(3):
        at.run(it, 1, 2, hist)
        s = {k: v.cpu().numpy() for k, v in at.peek().items()}
        am.pert = s["pert"].astype(np.float64)
        f = am.forward()
        np.testing.assert_allclose(s["adv"], f["adv"], atol=1e-7)
        np.testing.assert_allclose(s["recon"], f["recon"], atol=2e-6)
        _, i1, _, i2 = oracle.nn_distance(s["recon"], gt)
        assert np.array_equal(s["idx_r1"], i1) and np.array_equal(s["idx_r2"], i2), it
        _, i1, _, i2 = oracle.nn_distance(s["adv"], x)
        assert np.array_equal(s["idx_a1"], i1) and np.array_equal(s["idx_a2"], i2), it
        h = hist.cpu().numpy()[0]
        want = [f["loss_adv"], f["loss_dist"], f["loss_pert"], f["max_dist"], f["input_dist"], f["loss_ae"]]
        for k, wv in enumerate(want):
            np.testing.assert_allclose(h[k], wv, rtol=1e-5, atol=1e-12, err_msg="iteration %d metric %d" % (it, k))
    m, adv, recon = at.get_best(ae.get_loss_per_pc(gt))      # keep-best took iterations 2 and 3 (thresh 2)
    assert np.isfinite(m.cpu().numpy()).all() and (m[:, 4] > 0).all()


@pytest.mark.parametrize("prune", [True, False])
def test_config1_full_length_500_iterations_thresh_400(oracle, prune):
    """configs[1] at its own shape AND length (run_attack.py:34-35: 500 iterations, keep-best from iteration 400; B = 32,
    N = 2048, chamfer/chamfer), once with the paired grid search and once all-pairs.  After iterations 100 / 400 / 500 -- points
    hundreds of Adam steps away from their sources -- from the GPU's own state: all four nearest-neighbour index arrays
    np.array_equal to the pinned oracle for all 32 clouds, the six metric rows 1e-5 relative against the fp64 model evaluated at the
    GPU's pert, reconstruction 2e-6.  Then keep-best (adv_ae.py:234-246): get_best's metrics equal the bookkeeping of the reference
    loop fed the GPU's own per-iteration rows (strict '<' on loss_ae over iterations 400..500), and the kept clouds are the ones
    of the winning iteration (their Chamfer loss, recomputed by the oracle, is the kept error)."""
    import torch
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd.autoencoder import PointNetAE
    from oracle.attack_model import AEModel, AttackModel
    from conftest import cloud
    b, n, iters, thresh = 32, 2048, 500, 400
    w = W.synthetic_weights(n, seed=7)                    # bench.py's weights and seeds
    ae = PointNetAE(w, n)
    model = AEModel(W.canonical(w, n), n, np.float64)
    x, gt = cloud(1002, b, n), cloud(2002, b, n)
    at = AdvAE("adversary", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=iters, num_iterations_thresh=thresh,
                                          learning_rate=0.01, chamfer_prune=prune), ae=ae)
    at.set_inputs(x, gt, None, 1.0)
    at.init_pert(None, reset_optimizer=True)
    am = AttackModel(model, x, gt, None, np.ones(b))
    hist = torch.empty((iters, 6, b), device=ae.device)
    first = 0
    for upto in (100, 400, 500):
        at.run(first, upto - first, thresh, hist[first:upto])
        first = upto
        s = {k: v.cpu().numpy() for k, v in at.peek().items()}
        assert np.abs(s["pert"]).max() > (0.02 if upto >= 400 else 0.005)       # the clouds did move
        am.pert = s["pert"].astype(np.float64)
        f = am.forward()
        np.testing.assert_allclose(s["adv"], f["adv"], atol=1e-7)
        np.testing.assert_allclose(s["recon"], f["recon"], atol=2e-6)
        _, i1, _, i2 = oracle.nn_distance(s["recon"], gt)
        assert np.array_equal(s["idx_r1"], i1) and np.array_equal(s["idx_r2"], i2), upto
        _, i1, _, i2 = oracle.nn_distance(s["adv"], x)
        assert np.array_equal(s["idx_a1"], i1) and np.array_equal(s["idx_a2"], i2), upto
        h = hist[upto - 1].cpu().numpy()
        want = [f["loss_adv"], f["loss_dist"], f["loss_pert"], f["max_dist"], f["input_dist"], f["loss_ae"]]
        for k, wv in enumerate(want):
            np.testing.assert_allclose(h[k], wv, rtol=1e-5, atol=1e-12, err_msg="iteration %d metric %d" % (upto, k))
    ref = ae.get_loss_per_pc(gt)
    m, adv, recon = at.get_best(ref)
    at.status()
    m, adv, recon = m.cpu().numpy(), adv.cpu().numpy(), recon.cpu().numpy()
    h = hist.cpu().numpy()
    assert np.isfinite(h).all()
    # the reference's bookkeeping (adv_ae.py:197-200, 234-246) on the GPU's own rows
    best_err = np.full(b, 1e10, np.float32)
    best = np.zeros((b, 4), np.float32)
    best_it = np.full(b, -1)
    for it in range(iters):
        if it + 1 < thresh:
            continue
        take = h[it, 5] < best_err
        best_err[take] = h[it, 5][take]
        best[take] = np.stack([h[it, 0], h[it, 1], h[it, 4], h[it, 5]], axis=1)[take]
        best_it[take] = it
    assert (best_it >= thresh - 1).all()
    refv = np.asarray(ref.cpu().numpy() if hasattr(ref, "cpu") else ref, np.float32)
    assert np.array_equal(m[:, 0], best[:, 0]) and np.array_equal(m[:, 1], best[:, 1]) and np.array_equal(m[:, 2], best[:, 2])
    assert np.array_equal(m[:, 3], (best[:, 3] / refv).astype(np.float32)) and np.array_equal(m[:, 4], best_err)
    # the kept clouds belong to the kept error
    r1, _, r2, _ = oracle.nn_distance(recon, gt)
    np.testing.assert_allclose(r1.mean(1, dtype=np.float64) + r2.mean(1, dtype=np.float64), best_err, rtol=1e-5)
    a1, _, a2, _ = oracle.nn_distance(adv, x)
    np.testing.assert_allclose(a1.mean(1, dtype=np.float64) + a2.mean(1, dtype=np.float64), m[:, 2], rtol=1e-5)
    np.testing.assert_allclose(recon, model.reconstruct(adv.astype(np.float64))[0], atol=2e-6)


def test_config3_per_gpu_shape_b128_chamfer_plus_emd(oracle):
    """configs[3] at its per-GPU shape (B = 1024 over 8 GPUs => B = 128, N = 2048, loss_adv = Chamfer + EMD/N): one step.
    Gradient vs the fp64 model (pinned C approx_match / match_cost_grad inside) on 4 sampled clouds -- the batch is a sum of
    independent clouds, so a cloud's gradient does not depend on the others --; transport-plan properties of the GPU's own
    match on ALL 128 clouds: non-negative, every target column receives at most its capacity 1, every source row ships at
    most 1, total mass within 2 % of N (tf_approxmatch.cpp:23-84: remainders decay geometrically over the 10 levels);
    and all 537 M plan entries of the fast weight mode against the reference mode (the CPU op's weights bit for bit, itself
    held to 2 ulps of the oracle in test_gpu_emd.py): the statistical bound of the fast mode at the full shape."""
    import torch
    from geometric_adv_amd import ops, weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd.autoencoder import PointNetAE
    from oracle.attack_model import AEModel, AttackModel
    from conftest import cloud
    b, n = 128, 2048
    w = W.synthetic_weights(n, seed=7)
    ae = PointNetAE(w, n)
    x, gt = cloud(1003, b, n), cloud(2003, b, n)
    at = AdvAE("adversary", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=2, num_iterations_thresh=1,
                                          emd_weight=1.0), ae=ae)
    at.set_inputs(x, gt, None, 1.0)
    p0 = (1e-3 * np.random.default_rng(3).standard_normal((b, n, 3))).astype(np.float32)
    at.init_pert(p0, reset_optimizer=True)
    s = {k: v.cpu().numpy() for k, v in at.peek().items()}
    hist = torch.empty((1, 6, b), device=ae.device)
    at.run(0, 1, 1, hist)
    got = at.peek()["grad"].cpu().numpy()
    sel = [0, 37, 90, 127]
    model = AEModel(W.canonical(w, n), n, np.float64)
    am = AttackModel(model, x[sel], gt[sel], None, np.ones(len(sel)), emd_weight=1.0)
    am.init_pert(p0[sel])
    f = am.forward(idx_override=tuple(s[k][sel] for k in ("idx_r1", "idx_r2", "idx_a1", "idx_a2")))
    np.testing.assert_allclose(s["recon"][sel], f["recon"], atol=2e-6)
    g = am.gradient(f)
    sc = np.abs(g).reshape(len(sel), -1).max(1)[:, None, None]
    np.testing.assert_allclose(got[sel] / sc, g / sc, atol=2e-4)
    # the plan the loop used = approx_match of the forward's reconstruction (recomputed every forward: NoGradient op)
    recon = torch.as_tensor(s["recon"]).to(ae.device)
    gtd = torch.as_tensor(gt).to(ae.device)
    for lo in range(0, b, 32):                               # 32 x 2048 x 2048 floats = 537 MB per slice
        match = ops.approx_match(recon[lo:lo + 32], gtd[lo:lo + 32])       # (b, m, n) GPU layout
        assert (match >= 0).all()
        per_target, per_source = match.sum(2), match.sum(1)
        assert per_target.max() <= 1.0 + 1e-4 and per_source.max() <= 1.0 + 1e-4
        total = match.sum((1, 2))
        assert (total > 0.98 * n).all() and (total <= n * (1 + 1e-5)).all()
        cost = ops.match_cost(recon[lo:lo + 32], gtd[lo:lo + 32], match)
        assert torch.isfinite(cost).all() and (cost > 0).all()
        ref = ops.approx_match(recon[lo:lo + 32], gtd[lo:lo + 32], reference_weights=True)
        err = (match - ref).abs()
        outside = (err > 2e-6 + 2e-5 * ref.abs()).sum().item()
        assert outside <= 1e-5 * ref.numel() and err.max().item() < 1e-3, (lo, outside, err.max().item())
        del ref, err


def test_config4_full_batch_b32_n8192_indices(oracle):
    """configs[4] at its per-GPU shape (B = 256 over 8 GPUs => B = 32, N = 8192): two iterations of the loop, then all four
    nearest-neighbour index arrays of 4 sampled clouds against the pinned oracle on the GPU's own clouds, and the six metric
    rows of those clouds against fp64 means of the oracle's distances (1e-5 relative)."""
    import torch
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from conftest import cloud
    b, n = 32, 8192
    w = W.synthetic_weights(n, seed=7)
    x, gt = cloud(1004, b, n), cloud(2004, b, n)
    at = AdvAE("adversary", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=3, num_iterations_thresh=1))
    at.set_inputs(x, gt, None, 1.0)
    at.init_pert(None, reset_optimizer=True)
    hist = torch.empty((2, 6, b), device=at.device)
    at.run(0, 2, 1, hist)
    s = {k: v.cpu().numpy() for k, v in at.peek().items()}
    h = hist.cpu().numpy()[-1]
    sel = [0, 11, 20, 31]
    r1, i1, r2, i2 = oracle.nn_distance(s["recon"][sel], gt[sel])
    assert np.array_equal(s["idx_r1"][sel], i1) and np.array_equal(s["idx_r2"][sel], i2)
    a1, j1, a2, j2 = oracle.nn_distance(s["adv"][sel], x[sel])
    assert np.array_equal(s["idx_a1"][sel], j1) and np.array_equal(s["idx_a2"][sel], j2)
    np.testing.assert_allclose(h[5][sel], r1.mean(1, dtype=np.float64) + r2.mean(1, dtype=np.float64), rtol=1e-5)
    np.testing.assert_allclose(h[4][sel], a1.mean(1, dtype=np.float64) + a2.mean(1, dtype=np.float64), rtol=1e-5)


def _full_size_checks(oracle, at, x, gt, sel, hist, emd):
    """Sampled clouds of a full-size run against the pinned oracle on the GPU's own clouds: the four index arrays exact, the
    Chamfer metric rows within 1e-5 of fp64 means of the oracle's distances; status OK, every metric finite."""
    at.status()
    s = at.peek()
    h = hist.cpu().numpy()[-1]
    assert np.isfinite(hist.cpu().numpy()).all()
    recon, adv = s["recon"][sel].cpu().numpy(), s["adv"][sel].cpu().numpy()
    r1, i1, r2, i2 = oracle.nn_distance(recon, gt[sel])
    assert np.array_equal(s["idx_r1"][sel].cpu().numpy(), i1) and np.array_equal(s["idx_r2"][sel].cpu().numpy(), i2)
    a1, j1, a2, j2 = oracle.nn_distance(adv, x[sel])
    assert np.array_equal(s["idx_a1"][sel].cpu().numpy(), j1) and np.array_equal(s["idx_a2"][sel].cpu().numpy(), j2)
    cham = r1.mean(1, dtype=np.float64) + r2.mean(1, dtype=np.float64)
    dist = a1.mean(1, dtype=np.float64) + a2.mean(1, dtype=np.float64)
    np.testing.assert_allclose(h[4][sel], dist, rtol=1e-5)        # input_dist (adv_ae.py:132)
    np.testing.assert_allclose(h[5][sel], cham, rtol=1e-5)        # loss_ae (adv_ae.py:121)
    if not emd:
        np.testing.assert_allclose(h[0][sel], cham, rtol=1e-5)    # loss_adv = loss_ae for the Chamfer attack
    return s, h, cham


def test_config3_full_size_b1024_on_one_gpu(oracle):
    """configs[3] WHOLE on one GPU (B = 1024 x N = 2048, loss_adv = Chamfer + approx-EMD / N; b*n*m = 2^32 pair weights per level
    -- past every 32-bit product): two iterations.  4 sampled clouds' indices against the pinned oracle, the input-distance row
    within 1e-5, the adversarial-loss row = Chamfer (oracle) + match_cost / N of the GPU's own plan on those clouds (the plan
    held to the CPU op in test_gpu_emd.py), transport-plan properties on one 32-cloud slice, attack_status OK."""
    import torch
    from geometric_adv_amd import ops, weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd.autoencoder import PointNetAE
    from conftest import cloud
    b, n = 1024, 2048
    w = W.synthetic_weights(n, seed=7)
    ae = PointNetAE(w, n)
    x, gt = cloud(1013, b, n), cloud(2013, b, n)
    at = AdvAE("adversary", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=3, num_iterations_thresh=1,
                                          emd_weight=1.0), ae=ae)
    at.set_inputs(x, gt, None, 1.0)
    at.init_pert(None, reset_optimizer=True)
    hist = torch.empty((2, 6, b), device=ae.device)
    at.run(0, 2, 1, hist)
    sel = [0, 333, 800, 1023]
    s, h, cham = _full_size_checks(oracle, at, x, gt, sel, hist, emd=True)
    gtd = torch.as_tensor(gt).to(ae.device)
    sl = slice(992, 1024)                                     # the LAST slice: its offsets are the largest of the batch
    match = ops.approx_match(s["recon"][sl], gtd[sl])
    assert (match >= 0).all()
    assert match.sum(2).max() <= 1.0 + 1e-4 and match.sum(1).max() <= 1.0 + 1e-4
    total = match.sum((1, 2))
    assert (total > 0.98 * n).all() and (total <= n * (1 + 1e-5)).all()
    cost = ops.match_cost(s["recon"][sl], gtd[sl], match).cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(h[0][1023], cham[3] + cost[31] / n, rtol=2e-5)      # loss_adv = Chamfer + EMD cost / N


def test_config4_full_size_b256_n8192_on_one_gpu(oracle):
    """configs[4] WHOLE on one GPU (B = 256 x N = 8192, output-space attack; b*n*m = 2^34 pair distances per nn_distance): two
    iterations, 4 sampled clouds' four index arrays against the pinned oracle on the GPU's own clouds, both Chamfer metric
    rows within 1e-5, attack_status OK."""
    import torch
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from conftest import cloud
    b, n = 256, 8192
    w = W.synthetic_weights(n, seed=7)
    x, gt = cloud(1014, b, n), cloud(2014, b, n)
    at = AdvAE("adversary", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=3, num_iterations_thresh=1))
    at.set_inputs(x, gt, None, 1.0)
    at.init_pert(None, reset_optimizer=True)
    hist = torch.empty((2, 6, b), device=at.device)
    at.run(0, 2, 1, hist)
    _full_size_checks(oracle, at, x, gt, [0, 100, 201, 255], hist, emd=False)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` from a bare shell (no torchrun, WORLD_SIZE unset): the parent spawns the two ranks before
    touching the GPU and relays rank 0's line.  The box has one GPU, so GEOADV_BENCH_SHARE_GPU=1 puts both ranks on cuda:0 with
    a gloo group (RCCL refuses two ranks on one device): the N > 1 code path -- weak leg, all-pairs leg, strong-scaling leg,
    the gather of the final scalars -- runs for real."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["GEOADV_BENCH_SHARE_GPU"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--windows", "3"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and out["config"]["global_batch"] == 64
    assert out["scaling"] == "weak" and out["value"] > 0 and out["value_all_pairs"] > 0
    s = out["strong_scaling"]
    assert s["global_batch"] == 32 and s["batch_per_gpu"] == 16 and s["value"] > 0
    assert out["roofline"]["launches_timed"] >= 50 and 0 < out["roofline"]["frac"] < 1
