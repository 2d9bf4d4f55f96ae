"""GPU parity of the network and the attack iteration against the numpy model
(oracle/attack_model.py, fp64) and the pinned C oracle.

Tolerances (fp32 GPU arithmetic vs fp64 model; north_star: 1e-5 relative on the Chamfer loss,
bit-exact indices):
  latent / reconstruction : 2e-6 abs   (the reference's own sanity tolerance is 1e-6, run_defense_critical.py:121-123)
  per-cloud losses        : 1e-5 rel
  gradient w.r.t. pert    : 1e-4 rel of the per-cloud max (sums of ~1e3 fp32 products)
  NN indices              : exact, checked by re-running the oracle on the GPU's own clouds
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 256


@pytest.fixture(scope="module")
def setup():
    import torch
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.autoencoder import PointNetAE
    from oracle.attack_model import AEModel
    assert torch.cuda.is_available()
    w = W.randomized_weights(N)
    ae = PointNetAE(w, N)
    model = AEModel(W.canonical(w, N), N, np.float64)
    return w, ae, model


def _clouds(seed, b, n=N):
    from conftest import cloud
    return cloud(seed, b, n), cloud(seed + 1, b, n)


def test_ae_forward_matches_model(setup):
    w, ae, model = setup
    for b in (1, 5, 40):
        x, _ = _clouds(10 + b, b)
        recon, z = ae.forward(x)
        want_recon, want_z = model.reconstruct(x)
        np.testing.assert_allclose(z.cpu().numpy(), want_z, atol=2e-6, rtol=1e-6)
        np.testing.assert_allclose(recon.cpu().numpy(), want_recon, atol=2e-6, rtol=1e-6)


def test_ae_forward_ragged_point_count():
    """n not a multiple of the 64-point tile / 32-column block."""
    import torch
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.autoencoder import PointNetAE
    from oracle.attack_model import AEModel
    n = 200
    w = W.randomized_weights(n, seed=5)
    ae = PointNetAE(w, n)
    model = AEModel(W.canonical(w, n), n)
    x, _ = _clouds(3, 3, n)
    recon, z = ae.forward(x)
    want_recon, want_z = model.reconstruct(x)
    np.testing.assert_allclose(z.cpu().numpy(), want_z, atol=2e-6)
    np.testing.assert_allclose(recon.cpu().numpy(), want_recon, atol=2e-6)


def test_latent_is_permutation_invariant_and_duplicate_invariant(setup):
    """Properties of the symmetric max-pool the defense relies on (adversary_utils.py:166:
    'duplication of last point does not change the latent vector due to global pooling')."""
    import torch
    w, ae, model = setup
    x, _ = _clouds(77, 4)
    _, z = ae.forward(x)
    perm = np.random.default_rng(0).permutation(N)
    _, zp = ae.forward(x[:, perm])
    assert torch.equal(z, zp)
    xd = x.copy(); xd[:, N // 2:] = xd[:, N // 2 - 1:N // 2]
    xs = x.copy(); xs[:, N // 2:] = x[:, :1]                    # different padding point, same prefix ...
    xs[:, N // 2:] = xs[:, N // 2 - 1:N // 2]                   # ... then the same duplication
    _, z1 = ae.forward(xd); _, z2 = ae.forward(xs)
    assert torch.equal(z1, z2)


def _mk_attack(w, ae, b, adv_type="chamfer", dist_type="chamfer", **kw):
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    conf = Configuration(batch_size=b, n_points=N, weights=w, loss_adv_type=adv_type, loss_dist_type=dist_type,
                         num_iterations=10, num_iterations_thresh=5, **kw)
    return AdvAE("adversary", conf, ae=ae)


@pytest.mark.parametrize("adv_type,dist_type,kw", [
    ("chamfer", "chamfer", {}),
    ("latent", "chamfer", {}),
    ("chamfer", "pert", {}),
    ("latent", "pert", {"max_point_pert_weight": 0.5}),
    ("chamfer", "chamfer", {"max_point_dist_weight": 2.0}),
])
def test_single_iterations_match_model(setup, oracle, adv_type, dist_type, kw):
    """Three consecutive iterations, each checked from the GPU's own state: forward values,
    exact NN indices, gradient, Adam update."""
    import torch
    from oracle.attack_model import AttackModel
    w, ae, model = setup
    b = 3
    x, gt = _clouds(21, b)
    tz = model.encode(gt)
    dw = np.array([1.0, 150.0, 0.3], np.float32)
    at = _mk_attack(w, ae, b, adv_type, dist_type, **kw)
    at.set_inputs(x, gt, tz.astype(np.float32), dw)
    rng = np.random.default_rng(5)
    p0 = (1e-3 * rng.standard_normal((b, N, 3))).astype(np.float32)   # big enough that 'pert' losses are well conditioned
    at.init_pert(p0, reset_optimizer=True)
    am = AttackModel(model, x, gt, tz, dw, adv_type, dist_type, lr=0.01,
                     max_point_pert_weight=kw.get("max_point_pert_weight", 0.0),
                     max_point_dist_weight=kw.get("max_point_dist_weight", 0.0))
    am.init_pert(p0)
    hist = torch.empty((1, 6, b), device=ae.device)
    for it in range(3):
        s = {k: v.cpu().numpy() for k, v in at.peek().items()}       # forward of the current pert
        am.pert = s["pert"].astype(np.float64)                       # re-sync the model to the GPU state
        f = am.forward()
        np.testing.assert_allclose(s["adv"], f["adv"], atol=1e-7)
        np.testing.assert_allclose(s["latent"], f["z"], atol=2e-6)
        np.testing.assert_allclose(s["recon"], f["recon"], atol=2e-6)
        # exact indices: the pinned oracle on the GPU's own clouds
        _, i1, _, i2 = oracle.nn_distance(s["recon"], gt)
        assert np.array_equal(s["idx_r1"], i1) and np.array_equal(s["idx_r2"], i2)
        _, i1, _, i2 = oracle.nn_distance(s["adv"], x)
        assert np.array_equal(s["idx_a1"], i1) and np.array_equal(s["idx_a2"], i2)
        # gradient with the matches pinned to the GPU's
        f = am.forward(idx_override=(s["idx_r1"], s["idx_r2"], s["idx_a1"], s["idx_a2"]))
        g = am.gradient(f)
        at.run(it, 1, 1, hist)
        s2 = {k: v.cpu().numpy() for k, v in at.peek().items()}
        scale = np.abs(g).reshape(b, -1).max(1)[:, None, None]
        np.testing.assert_allclose(s2["grad"] / scale, g / scale, atol=1e-4)
        # Adam from the GPU's own gradient (isolates the update rule; the model's slots follow
        # the GPU's gradients so that only the rule itself is compared)
        am.adam(s2["grad"].astype(np.float64))
        np.testing.assert_allclose(s2["pert"], am.pert, rtol=2e-6, atol=2e-8)
        # metrics row of this iteration = losses of the UPDATED pert
        am.pert = s2["pert"].astype(np.float64)
        f2 = am.forward(idx_override=(s2["idx_r1"], s2["idx_r2"], s2["idx_a1"], s2["idx_a2"]))
        h = hist.cpu().numpy()[0]
        fourth = f2["loss_max"] if dist_type == "pert" else f2["max_dist"]
        for row, want in zip(h, [f2["loss_adv"], f2["loss_dist"], f2["loss_pert"], fourth, f2["input_dist"], f2["loss_ae"]]):
            np.testing.assert_allclose(row, want, rtol=1e-5, atol=1e-12)


def test_maxpool_exact_ties_split_gradient(setup):
    """Duplicated points give exact positive ties in the max-pool; TF's reduce_max gradient splits
    equally among them (the dense fallback path of the encoder backward)."""
    from oracle.attack_model import AttackModel
    w, ae, model = setup
    b = 2
    x, gt = _clouds(31, b)
    x[0, 1::2] = x[0, 0::2]                                         # cloud 0: every point twice -> ties everywhere
    tz = model.encode(gt)
    at = _mk_attack(w, ae, b, "latent", "chamfer")
    at.set_inputs(x, gt, tz.astype(np.float32), 1.0)
    p0 = np.zeros((b, N, 3), np.float32)
    at.init_pert(p0, reset_optimizer=True)
    am = AttackModel(model, x, gt, tz, np.ones(b), "latent", "chamfer")
    am.init_pert(p0)
    s = {k: v.cpu().numpy() for k, v in at.peek().items()}
    f = am.forward(idx_override=(s["idx_r1"], s["idx_r2"], s["idx_a1"], s["idx_a2"]))
    g = am.gradient(f)
    at.run(0, 1, 1)
    got = at.peek()["grad"].cpu().numpy()
    scale = np.abs(g).reshape(b, -1).max(1)[:, None, None]
    np.testing.assert_allclose(got / scale, g / scale, atol=1e-4)
    # the tied cloud really exercised the split: duplicates carry identical encoder gradients
    assert np.abs(g[0]).max() > 0


def test_keep_best_bookkeeping(setup):
    """adv_ae.py:234-249: among iterations with index+1 >= thresh keep the FIRST minimum of the
    target reconstruction error, its metrics row and its clouds."""
    import torch
    w, ae, model = setup
    b, iters, thresh = 4, 12, 6
    x, gt = _clouds(41, b)
    ref = ae.get_loss_per_pc(ae.forward(gt)[0], gt) * 0 + ae.get_loss_per_pc(gt)      # target_ae_loss_ref
    at = _mk_attack(w, ae, b)
    at.set_inputs(x, gt, None, 1.0)
    at.init_pert(None, reset_optimizer=True)
    hist = torch.empty((iters, 6, b), device=ae.device)
    snaps = []
    for it in range(iters):                                              # one at a time so every state can be snapshotted
        at.run(it, 1, thresh, hist[it:it + 1])
        s = at.peek()
        snaps.append((s["adv"].clone(), s["recon"].clone()))
    metrics, adv, recon = at.get_best(ref)
    m_only, no_adv, no_recon = at.get_best(ref, clouds=False)           # the metrics alone (what bench.py's timed window fetches)
    assert no_adv is None and no_recon is None and torch.equal(m_only, metrics)
    h = hist.cpu().numpy()
    metrics = metrics.cpu().numpy()
    for j in range(b):
        err = h[thresh - 1:, 5, j]
        k = int(np.argmin(err)) + thresh - 1                             # np.argmin = first minimum = strict '<'
        assert metrics[j, 4] == h[k, 5, j]
        assert metrics[j, 0] == h[k, 0, j] and metrics[j, 1] == h[k, 1, j] and metrics[j, 2] == h[k, 4, j]
        np.testing.assert_allclose(metrics[j, 3], h[k, 5, j] / ref[j], rtol=1e-6)
        assert torch.equal(adv[j], snaps[k][0][j]) and torch.equal(recon[j], snaps[k][1][j])


def test_run_in_one_call_equals_run_in_pieces(setup):
    """The loop is deterministic: 8 iterations in one call == 8 calls of one iteration, bit for bit."""
    import torch
    w, ae, model = setup
    b = 3
    x, gt = _clouds(51, b)
    outs = []
    for pieces in (1, 8):
        at = _mk_attack(w, ae, b)
        at.set_inputs(x, gt, None, 1.0)
        at.init_pert(None, reset_optimizer=True)
        per = 8 // pieces
        for p in range(pieces):
            at.run(p * per, per, 4)
        outs.append(at.peek()["pert"].clone())
    assert torch.equal(outs[0], outs[1])


def test_attack_api_shapes_and_progress(setup):
    """AdvAE.attack (adv_ae.py:155-189): output shapes, and the attack makes progress: the target
    reconstruction error after the attack is below the error of the un-attacked source."""
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    w, ae, model = setup
    b, n_ex = 2, 4
    x, gt = _clouds(61, n_ex)
    conf = Configuration(batch_size=b, n_points=N, weights=w, dist_weight_list=[0.5, 2.0], num_iterations=40,
                         num_iterations_thresh=30)
    at = AdvAE("adversary", conf, ae=ae)
    ref = ae.get_loss_per_pc(gt)
    tz = ae.transform(gt)
    metrics, adv, recon = at.attack(x, tz, gt, ref, conf)
    assert metrics.shape == (2, n_ex, 5) and adv.shape == (2, n_ex, N, 3) and recon.shape == (2, n_ex, N, 3)
    before = ae.get_loss_per_pc(x, gt)                                  # chamfer(recon(source), target)
    assert (metrics[:, :, 4] < before[None, :]).all()
    with pytest.raises(AssertionError):
        at.attack(x[:3], tz[:3], gt[:3], ref[:3], conf)                 # 3 % 2 != 0 (adv_ae.py:162)


def test_batch_slots_equal_one_handle_per_run_of_batches(setup):
    """Configuration.batch_slots = 2: two batches in flight on one GPU (own handle, stream and host thread each).  Slot k
    takes the k-th contiguous run of batches, so the result must equal two separate handles walking those runs one
    after the other -- bit for bit, Adam slots carried from batch to batch included -- and the log keeps batch order."""
    import io
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    w, ae, model = setup
    b, n_ex = 2, 10                                                     # 5 batches: runs of 3 and 2
    x, gt = _clouds(71, n_ex)
    ref = ae.get_loss_per_pc(gt)
    tz = ae.transform(gt)
    kw = dict(batch_size=b, n_points=N, weights=w, dist_weight_list=[0.5, 2.0], num_iterations=25, num_iterations_thresh=20)
    log = io.StringIO()
    got = AdvAE("adversary", Configuration(batch_slots=2, **kw), ae=ae).attack(x, tz, gt, ref, log_file=log)
    want = []
    for lo, hi in ((0, 6), (6, 10)):
        want.append(AdvAE("adversary", Configuration(**kw), ae=ae).attack(x[lo:hi], tz[lo:hi], gt[lo:hi], ref[lo:hi]))
    for k in range(3):
        assert np.array_equal(got[k], np.concatenate([want[0][k], want[1][k]], axis=1))
    marks = [ln for ln in log.getvalue().splitlines() if ln.startswith("Batch ")]
    assert [int(ln.split()[1]) for ln in marks] == [1, 2, 3, 4, 5]


def test_odd_sizes_all_code_paths_agree():
    """Ragged and extreme cloud sizes (1 point, sizes around the 64-row encoder tile, the 256-row Chamfer tile, the 2048 /
    4096-point limits of the grid search's instantiations) through every alternative code path the library keeps behind a
    geoadv_attack_config switch: the whole trajectory -- perturbation, nearest-neighbour indices, keep-best metrics -- must be
    bit-identical to the default path.  Each path runs in its own process.  (The recomputing
    encoder backward is not in the list: it sums the same products in another MFMA shape's order and is held to a
    tolerance in test_masked_backward_equals_recomputing_backward, odd sizes included.)"""
    import json, os, subprocess, sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_paths_child.py")
    sizes = ["1", "2", "31", "64", "65", "255", "257", "1023", "2049", "4097"]
    got = {}
    M = {"encoder_backward": "masked"}        # (the two backward forms agree to rounding, not bit for bit: each group fixes one)
    J = {"encoder_backward": "jacobian"}
    for name, cfg in [("default", {"chamfer_prune": "always", **M}), ("all-pairs source distance", {"chamfer_prune": False, **M}),
                      ("Adam in its own launch", {"separate_adam": True, "chamfer_prune": "always", **M}),
                      ("two-scan Chamfer", {"chamfer_kernel": "two_scan", "chamfer_prune": "always", **M}),
                      ("two-scan Chamfer, all-pairs source distance", {"chamfer_kernel": "two_scan", "chamfer_prune": False, **M}),
                      ("symmetric Chamfer + grid search", {"chamfer_kernel": "symmetric", "chamfer_prune": "always", **M}),
                      ("auto (tiny batch: all-pairs)", M),
                      ("symmetric Chamfer, all-pairs source distance", {"chamfer_kernel": "symmetric", "chamfer_prune": False, **M}),
                      ("jacobian: default", {"chamfer_kernel": "symmetric", "chamfer_prune": "always", **J}),
                      ("jacobian: in a launch of its own", {"chamfer_kernel": "two_scan", "chamfer_prune": False, **J}),
                      ("jacobian: beside the scan, Adam in its own launch", {"chamfer_kernel": "symmetric", "separate_adam": True, **J})]:
        o = subprocess.run([sys.executable, child, json.dumps(cfg)] + sizes, capture_output=True, text=True, timeout=600)
        lines = [ln for ln in o.stdout.splitlines() if ln.startswith("HASHES ")]
        assert o.returncode == 0 and lines, (name, o.stderr[-400:])
        got[name] = json.loads(lines[-1][7:])
    for name, h in got.items():
        assert h == got["jacobian: default" if name.startswith("jacobian") else "default"], name


@pytest.mark.parametrize("reference_weights", [False, True])
def test_emd_combined_loss_step(setup, reference_weights):
    """configs[3]: Chamfer + EMD combined adversarial loss (build-defined: loss_adv = chamfer +
    emd_weight * match_cost(recon, gt) / N, match held constant in the backward like the reference's
    NoGradient registration): loss value and gradient of one iteration against the model, with the plan's pair weights in
    either mode (Configuration.emd_reference_weights)."""
    import torch
    from oracle.attack_model import AttackModel
    w, ae, model = setup
    b = 2
    x, gt = _clouds(71, b)
    at = _mk_attack(w, ae, b, "chamfer", "chamfer", emd_weight=0.5, emd_reference_weights=reference_weights)
    at.set_inputs(x, gt, None, 1.0)
    p0 = (1e-3 * np.random.default_rng(2).standard_normal((b, N, 3))).astype(np.float32)
    at.init_pert(p0, reset_optimizer=True)
    am = AttackModel(model, x, gt, None, np.ones(b), emd_weight=0.5)
    am.init_pert(p0)
    s = {k: v.cpu().numpy() for k, v in at.peek().items()}
    f = am.forward(idx_override=(s["idx_r1"], s["idx_r2"], s["idx_a1"], s["idx_a2"]))
    g = am.gradient(f)
    hist = torch.empty((1, 6, b), device=ae.device)
    at.run(0, 1, 1, hist)
    got = at.peek()["grad"].cpu().numpy()
    sc = np.abs(g).reshape(b, -1).max(1)[:, None, None]
    np.testing.assert_allclose(got / sc, g / sc, atol=2e-4)
    # metrics of the updated state: loss_adv includes the EMD term, loss_ae (the keep-best key) does not
    am.pert = at.peek()["pert"].cpu().numpy().astype(np.float64)
    f2 = am.forward()
    h = hist.cpu().numpy()[0]
    np.testing.assert_allclose(h[5], f2["loss_ae"], rtol=1e-5)
    np.testing.assert_allclose(h[0], f2["loss_adv"], rtol=2e-5)
    assert (h[0] > h[5]).all()


@pytest.mark.parametrize("n", [N, 200, 1, 65, 257])
def test_masked_backward_equals_recomputing_backward(setup, n):
    """The sparse encoder backward reads the ReLU masks the forward left behind (16-row tiles on the 16x16x4 MFMA shape); with
    Configuration(recompute_backward=True) it re-runs the forward for the critical rows instead (32-row tiles, 32x32x2).  The same products in
    two summation orders: from ONE state -- the perturbation of the masked run after 0 ... 5 Adam steps -- the two gradients agree to
    2e-6 of their largest component (also for a ragged point count, whose last tile is partly empty), and the perturbations after
    that step to 1e-3 of their size.  (Two six-step TRAJECTORIES are not compared: they differ by ~1e-5 after a step, and one
    nearest-neighbour or critical-point switch that only one of them sees by iteration 5 is a 1e-3 difference under any of the
    encoder arithmetics -- tools/debug/recompute_spread.py.)"""
    import torch
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.autoencoder import PointNetAE
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    w = W.randomized_weights(n)
    ae = PointNetAE(w, n)
    b = 3
    x, gt = _clouds(71, b, n)
    make = lambda recompute: AdvAE("adversary", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=6, num_iterations_thresh=3,
                                                              recompute_backward=recompute), ae=ae)
    lead = make(False)
    lead.set_inputs(x, gt, None, 1.0)
    lead.init_pert(None, reset_optimizer=True)
    for it in range(6):
        state = lead.peek()["pert"].cpu().numpy()
        grads, outs = [], []
        for recompute in (False, True):
            at = make(recompute)
            at.set_inputs(x, gt, None, 1.0)
            at.init_pert(state, reset_optimizer=True)
            at.run(0, 1, 3)
            pk = at.peek()
            grads.append(pk["grad"].clone())
            outs.append(pk["pert"].clone())
        sc = grads[1].abs().amax((1, 2), keepdim=True)
        assert (sc > 0).all()
        torch.testing.assert_close(grads[0] / sc, grads[1] / sc, rtol=0, atol=2e-6)
        # (the step itself is Adam's first, lr * g / (|g| + eps'): a component whose gradient is ~0 moves by anything up to lr
        # on a 1e-6 difference, so the perturbations are held to 1e-3 of their size only)
        assert outs[0].abs().max() > 0
        torch.testing.assert_close(outs[0], outs[1], rtol=0, atol=1e-3 * outs[1].abs().max().item())
        lead.run(it, 1, 3)


@pytest.mark.parametrize("n,form", [(2048, "auto"), (256, "masked"), (256, "jacobian")])
def test_trajectory_of_a_cloud_does_not_depend_on_its_batch(n, form):
    """which batch a cloud sits in must not change its trajectory.  Every kernel of the loop sums a cloud's numbers
    in an order fixed by the cloud alone -- the forward's 32- and 64-row tiles run the same chains, the backward (masked or
    pool Jacobian) uses 16-row tiles at every batch size, the Chamfer results are exact -- so the perturbation of a cloud after
    8 iterations is bit-identical whether it is attacked in a batch of 8, 32 or 40.  (What does change the bits: the backward
    FORM -- masked vs Jacobian agree to rounding, test_jacobian_backward_equals_masked_backward -- which `auto` picks by
    batch size -- batch * n_points below / from GEOADV_SYM_MIN_POINTS = 4096 (include/geoadv.h), i.e. one cloud / from two clouds
    of 2048 points; the bit-for-bit statements of dist.py / run_attack.py hold at equal
    configuration.)  Batches of 2 and 4 ride along: the symmetric scan's atomic row-minimum form (narrow column slices)."""
    import torch
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.autoencoder import PointNetAE
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    w = W.randomized_weights(n)
    ae = PointNetAE(w, n)
    x, gt = _clouds(91, 40, n)
    p0 = (1e-3 * np.random.default_rng(3).standard_normal((40, n, 3))).astype(np.float32)
    outs = {}
    for b in (2, 4, 8, 32, 40):
        at = AdvAE("adversary", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=8, num_iterations_thresh=3,
                                              encoder_backward=form), ae=ae)
        at.set_inputs(x[:b], gt[:b], None, 1.0)
        at.init_pert(p0[:b], reset_optimizer=True)
        at.run(0, 8, 3)
        outs[b] = at.peek()["pert"][:8].clone()
    assert outs[8].abs().max() > 0
    assert torch.equal(outs[8], outs[32]) and torch.equal(outs[8], outs[40])
    assert torch.equal(outs[2], outs[8][:2]) and torch.equal(outs[4], outs[8][:4])


@pytest.mark.parametrize("kernel", ["symmetric", "two_scan"])
@pytest.mark.parametrize("n,b", [(N, 3), (N, 32), (200, 3), (1, 2), (65, 5), (257, 3)])
def test_jacobian_backward_equals_masked_backward(kernel, n, b):
    """The encoder backward as the pool Jacobian evaluated beside the forward (encoder_jac.h: J[c] = d z[c] / d adv[crit[c]], then
    g = sum dz[c] J[c] in the decoder backward's tail) against back-propagating dz through the critical rows: the same
    products summed per channel first instead of per point first -- the first iteration's gradient agrees to 2e-6 of its
    largest component, the perturbation after 6 Adam steps to 1e-4 of its size (but for a few elements in ten thousand).  Both hosts: beside the symmetric scan and as
    a launch of its own (two-scan kernel); clouds with duplicated points (tied pool maxima) take the dense path in both."""
    import torch
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.autoencoder import PointNetAE
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    w = W.randomized_weights(n)
    ae = PointNetAE(w, n)
    x, gt = _clouds(71, b, n)
    if n >= 64:
        x = x.copy(); x[0, 1] = x[0, 0]                    # one cloud with a duplicated point: an exact tie in its pool
    grads, outs = [], []
    for form in ("masked", "jacobian"):
        at = AdvAE("adversary", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=6, num_iterations_thresh=3,
                                              encoder_backward=form, chamfer_kernel=kernel), ae=ae)
        at.set_inputs(x, gt, None, 1.0)
        at.init_pert(np.zeros((b, n, 3), np.float32), reset_optimizer=True)      # zero pert: the duplicate stays a tie in iteration 1
        at.run(0, 1, 3)
        grads.append(at.peek()["grad"].clone())
        at.run(1, 5, 3)
        outs.append(at.peek()["pert"].clone())
    sc = grads[0].abs().amax((1, 2), keepdim=True)
    assert (sc > 0).all()
    torch.testing.assert_close(grads[1] / sc, grads[0] / sc, rtol=0, atol=2e-6)
    top = outs[0].abs().max().item()
    assert top > 0
    diff = (outs[1] - outs[0]).abs()
    # six Adam steps amplify a rounding difference where a coordinate's gradient is itself rounding noise (the step is
    # lr * m / (sqrt(v) + 1e-8)): all but a few elements per ten thousand stay within 1e-4 of the largest perturbation
    assert (diff <= 1e-4 * top).float().mean().item() >= 0.9995 and diff.max().item() <= 0.05 * top


@pytest.mark.parametrize("sym", ["symmetric", "two_scan"])
@pytest.mark.parametrize("lr,n", [(0.01, N), (0.3, N), (0.01, 200), (0.01, 8192), (0.3, 5000)])
def test_pruned_source_distance_equals_all_pairs(setup, lr, n, sym):
    """nn_distance(adv, x) through the paired grid search (default) vs the all-pairs kernel (Configuration(chamfer_prune=False)): the
    whole loop must agree bit for bit -- also when a huge learning rate scatters the points so that clouds hand themselves
    back to the all-pairs kernel, for a point count that is not a multiple of anything, and for clouds of more than 4096
    points (the search's large instantiation, launched on its own).  In both forms of the all-pairs side: the symmetric
    scan (large batches) and the plain scans gated by the search's hand-back flags (small ones)."""
    import torch
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.autoencoder import PointNetAE
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    w = W.randomized_weights(n)
    ae = PointNetAE(w, n)
    b = 3
    x, gt = _clouds(81, b, n)
    outs = []
    for prune in ("always", False):
        at = AdvAE("adversary", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=12, num_iterations_thresh=3,
                                              learning_rate=lr, chamfer_kernel=sym, chamfer_prune=prune), ae=ae)
        at.set_inputs(x, gt, None, 1.0)
        at.init_pert(None, reset_optimizer=True)
        at.run(0, 12, 3)
        p = at.peek()
        outs.append((p["pert"].clone(), p["idx_a1"].clone(), p["idx_a2"].clone()))
        # geoadv_attack_search_state: is the paired search in use, and how many clouds does it currently hand back
        searched, handed_back = at.search_state()
        assert searched == (prune == "always") and 0 <= handed_back <= b
        if not searched:
            assert handed_back == 0
        elif lr >= 0.3:
            assert handed_back > 0, "a step of 0.3 per iteration scatters the points out of their cells: clouds must hand themselves back"
    assert outs[0][0].abs().max() > 0
    for a, c in zip(outs[0], outs[1]):
        assert torch.equal(a, c)


@pytest.mark.parametrize("case", ["output", "latent", "tied"])
def test_forward_and_gradient_match_torch_golden(case):
    """The second opinion (oracle/torch_model.py: torch library layers + autograd, fp64; vectors in
    tests/golden/torch_second_opinion.npz written by oracle/make_golden_torch.py): latent / reconstruction 2e-6, per-cloud
    losses 1e-5 relative, NN indices exact (when the GPU's reconstruction picks the same matches), gradient 1e-4 of its
    maximum -- incl. the tied max-pool case (every point duplicated), where the gradient splits equally (TF _MinOrMaxGrad)."""
    import os
    import torch
    from conftest import GOLDEN
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.autoencoder import PointNetAE
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    g = np.load(os.path.join(GOLDEN, "torch_second_opinion.npz"))
    n, x, gt, pert = int(g[f"{case}_n"]), g[f"{case}_x"], g[f"{case}_gt"], g[f"{case}_pert"]
    b = len(x)
    w = W.randomized_weights(n, seed=int(g[f"{case}_wseed"]))
    ae = PointNetAE(w, n)
    at = AdvAE("adversary", Configuration(batch_size=b, n_points=n, weights=w, loss_adv_type=str(g[f"{case}_adv_type"]),
                                          loss_dist_type=str(g[f"{case}_dist_type"]), num_iterations=2, num_iterations_thresh=1), ae=ae)
    at.set_inputs(x, gt, g[f"{case}_tz"], float(g[f"{case}_dw"]))
    at.init_pert(pert, reset_optimizer=True)
    s = {k: v.cpu().numpy() for k, v in at.peek().items()}
    np.testing.assert_allclose(s["latent"], g[f"{case}_z"], atol=2e-6)
    np.testing.assert_allclose(s["recon"], g[f"{case}_recon"], atol=2e-6)
    same = all(np.array_equal(s[k], g[f"{case}_idx{i}"]) for i, k in enumerate(("idx_r1", "idx_r2", "idx_a1", "idx_a2")))
    hist = torch.empty((1, 6, b), device=ae.device)
    at.run(0, 1, 1, hist)                                          # backward of that forward + Adam
    got = at.peek()["grad"].cpu().numpy()
    want = g[f"{case}_grad"]
    if same:                                                       # (a flipped near-tie in fp32 would legitimately move the gradient)
        sc = np.abs(want).reshape(b, -1).max(1)[:, None, None]
        np.testing.assert_allclose(got / sc, want / sc, atol=1e-4)
    else:
        assert case != "tied"
    if case == "tied":
        assert np.array_equal(got[:, :n // 2], got[:, n // 2:])    # duplicates receive identical gradient


_SHARD_WORKER = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from geometric_adv_amd import dist as gdist, weights as W
from geometric_adv_amd.adv_ae import AdvAE, Configuration
rank, world, local = gdist.init("gloo")                      # both ranks share the one GPU of the test box (RCCL refuses that)
d = np.load(sys.argv[1])
n = d["x"].shape[1]
conf = Configuration(batch_size=2, n_points=n, weights=W.randomized_weights(n), dist_weight_list=[0.5, 2.0],
                     num_iterations=8, num_iterations_thresh=5)
at = AdvAE("adversary", conf, device="cuda:0")
log = open(sys.argv[2] + ".rank%d.txt" % rank, "w")
m, a, r, sl = gdist.attack_sharded(at, d["x"], d["tz"], d["gt"], d["ref"], gather_clouds=True, log_file=log)
log.close()
torch.cuda.synchronize()
np.savez(sys.argv[2] + ".rank%d.npz" % rank, m=m, a=a, r=r, lo=sl.start, hi=sl.stop)
torch.distributed.destroy_process_group()
"""


def test_real_attack_sharded_over_two_ranks_on_one_gpu(setup, tmp_path):
    """The north-star path through dist.attack_sharded with the REAL AdvAE and world size 2 (gloo group, both ranks on cuda:0):
    metrics / clouds gathered in example order on every rank, equal bit for bit to (a) two single-process attacks on the two
    shards and (b) one process with Configuration.batch_slots = 2; every rank logs its own batches."""
    import os, socket, subprocess, sys
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    w, ae, model = setup
    n_ex = 6                                                  # 3 batches of 2: rank 0 takes two of them, rank 1 one
    x, gt = _clouds(91, n_ex)
    tz = ae.transform(gt)
    ref = ae.get_loss_per_pc(gt)
    np.savez(tmp_path / "in.npz", x=x, gt=gt, tz=tz, ref=ref)
    (tmp_path / "worker.py").write_text(_SHARD_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", port, str(tmp_path / "worker.py"), str(tmp_path / "in.npz"), str(tmp_path / "out")]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=os.getcwd())
    got = [np.load(str(tmp_path / "out") + ".rank%d.npz" % r) for r in range(2)]
    assert (int(got[0]["lo"]), int(got[0]["hi"]), int(got[1]["lo"]), int(got[1]["hi"])) == (0, 4, 4, 6)
    for k in ("m", "a", "r"):
        assert np.array_equal(got[0][k], got[1][k])           # every rank holds the gathered result
    kw = dict(batch_size=2, n_points=N, weights=w, dist_weight_list=[0.5, 2.0], num_iterations=8, num_iterations_thresh=5)
    shards = [AdvAE("adversary", Configuration(**kw), ae=ae).attack(x[lo:hi], tz[lo:hi], gt[lo:hi], ref[lo:hi])
              for lo, hi in ((0, 4), (4, 6))]
    slots = AdvAE("adversary", Configuration(batch_slots=2, **kw), ae=ae).attack(x, tz, gt, ref)
    for i, k in enumerate(("m", "a", "r")):
        assert np.array_equal(got[0][k], np.concatenate([shards[0][i], shards[1][i]], axis=1)), k
        assert np.array_equal(got[0][k], slots[i]), k
    assert got[0]["m"].shape == (2, n_ex, 5) and got[0]["a"].shape == (2, n_ex, N, 3)
    logs = [open(str(tmp_path / "out") + ".rank%d.txt" % r).read() for r in range(2)]
    assert logs[0].count("Batch ") == 2 and logs[1].count("Batch ") == 1 and "Dist weight" in logs[1]


def test_roctx_markers_and_kernel_timing_leave_results_alone(setup):
    """Tracing hooks: roctx ranges (libroctx64 looked up at run time) and per-class timing -- kernel begin/end stamps for the
    encoder forward, bracketing events for the rest -- must not change a bit of the trajectory."""
    import torch
    w, ae, model = setup
    b = 3
    x, gt = _clouds(61, b)
    outs = []
    for traced in (False, True):
        at = _mk_attack(w, ae, b)
        at.set_inputs(x, gt, None, 1.0)
        at.init_pert(None, reset_optimizer=True)
        if traced:
            at.markers(True)
            at.profile(True)
        at.run(0, 6, 3)
        if traced:
            prof = at.profile_read()
            assert prof["encoder_fwd"][0] == 7                                                     # 6 iterations + the first forward
            assert prof["adam"][0] == 0                       # (the step rides in the next forward's point loads: no launch of its own)
            assert all(0 < ms < 50 for name, (_, ms) in prof.items() if name != "adam")
            at.profile(False); at.markers(False)
        outs.append(at.peek()["pert"].clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("b,prune", [(64, True), (64, False), (40, True), (32, True), (8, False), (16, "always")])
def test_loss_riders_in_the_scan_launch_equal_their_own_launch(b, prune):
    """Configuration.loss_in_scan: from two rounds of scan workgroups on, the loss / keep-best / Chamfer-gradient workgroups ride as
    the last workgroups of the symmetric scan's launch and wait for their cloud's scan and search workgroups through a per-cloud
    counter (csrc/loss_cgrad.h).  Same bodies, same order of every sum: 60 iterations (keep-best from iteration 20 on) must agree
    with the launch of their own bit for bit -- metrics history, perturbation, indices, gradient, best clouds."""
    import torch
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd.autoencoder import PointNetAE
    from conftest import cloud
    n = 2048
    w = W.synthetic_weights(n, seed=7)
    ae = PointNetAE(w, n)
    x, gt = cloud(501, b, n), cloud(502, b, n)
    out = {}
    for on in (True, False):
        at = AdvAE("a", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=60, num_iterations_thresh=20, chamfer_prune=prune,
                                      loss_in_scan="always" if on else False), ae=ae)
        at.set_inputs(x, gt, ae.transform(gt), 1.0)
        at.init_pert(None, reset_optimizer=True)
        h = torch.empty((60, 6, b), device=ae.device)
        at.run(0, 60, 20, h)
        at.status()
        best = at.get_best(ae.get_loss_per_pc(gt))
        out[on] = (h.clone(), {k: v.clone() for k, v in at.peek().items()}, [t.clone() for t in best])
        del at
    assert torch.equal(out[True][0], out[False][0])
    for k in out[True][1]:
        assert torch.equal(out[True][1][k], out[False][1][k]), k
    for a_, b_ in zip(out[True][2], out[False][2]):
        assert torch.equal(a_, b_)
