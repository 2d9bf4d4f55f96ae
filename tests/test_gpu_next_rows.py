"""GPU: the SURVEY 8(f) rows built so far -- f-1 bulk Chamfer scorer, f-3 critical-points defense."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


@pytest.mark.parametrize("na,nb,n,m", [(3, 5, 300, 300), (2, 2, 1024, 1024), (4, 3, 130, 517), (1, 7, 2048, 2048),
                                     (3, 1, 130, 517)])   # odd pair count x odd n + m: the workspace of the float4 column partials is re-aligned
def test_chamfer_matrix_vs_oracle(oracle, na, nb, n, m):
    """Every entry equals mean(dist1) + mean(dist2) of the pinned oracle's nn_distance on that pair (sums in
    float64 on the oracle side; the kernel sums fp32 in a fixed order => 1e-6 relative)."""
    from geometric_adv_amd import ops
    from conftest import cloud
    a, b = cloud(1, na, n), cloud(2, nb, m)
    got = ops.chamfer_dist_matrix(_t(a), _t(b)).cpu().numpy()
    want = np.empty((na, nb))
    for i in range(na):
        d1, _, d2, _ = oracle.nn_distance(np.repeat(a[i:i + 1], nb, 0), b)
        want[i] = d1.mean(1, dtype=np.float64) + d2.mean(1, dtype=np.float64)
    np.testing.assert_allclose(got, want, rtol=2e-6)


def test_chamfer_matrix_chunking_and_symmetry():
    """A workspace that only fits a few pairs gives the same bits as one that fits all; the matrix of a set against
    itself is symmetric with a zero diagonal (both directions of one pair are the same distances)."""
    import torch
    from geometric_adv_amd import ops
    from conftest import cloud
    pcs = _t(cloud(3, 12, 512))
    full = ops.chamfer_dist_matrix(pcs, pcs)
    small = ops.chamfer_dist_matrix(pcs, pcs, max_workspace_bytes=4 * 3 * 200000)
    assert torch.equal(full, small)
    assert torch.equal(full, full.T) and not torch.diagonal(full).any()


def test_scorer_slice_orientation(oracle):
    from geometric_adv_amd.scorer import get_chamfer_dist_mat_slice, sort_dist_mat_rows
    from conftest import cloud
    pcs = cloud(5, 9, 256)
    sl = get_chamfer_dist_mat_slice(pcs, 2, 4, row_chunk=5)
    assert sl.shape == (9, 4)
    d1, _, d2, _ = oracle.nn_distance(pcs[3:4], pcs[7:8])             # source = current cloud 2+1, target = cloud 7
    np.testing.assert_allclose(sl[7, 1], d1.mean() + d2.mean(), rtol=2e-6)
    order = sort_dist_mat_rows(sl.T)
    assert order.shape == (4, 9) and (order[:, 0] == np.arange(2, 6)).all()        # a cloud is its own nearest neighbour


def test_critical_points_match_pre_symmetry_argmax():
    """f-3: (max, argmax) from the fused encoder == np.max / np.argmax of the model's pre-symmetry activations
    (src/ae_utils.py:19-20), and the host-side bookkeeping reproduces ae_utils' outputs."""
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.autoencoder import PointNetAE
    from oracle.attack_model import AEModel
    from conftest import cloud
    n = 512
    w = W.randomized_weights(n, seed=11)
    ae = PointNetAE(w, n)
    model = AEModel(W.canonical(w, n), n)
    pcs = cloud(7, 5, n)
    mv, mi = ae.max_and_argmax(pcs)
    mv, mi = mv.cpu().numpy(), mi.cpu().numpy()
    _, hs = model.encode(pcs, keep=True)
    pre = hs[-1]                                                       # (b, n, 128) pre-symmetry data
    np.testing.assert_allclose(mv, pre.max(1), atol=2e-6)
    want_idx = pre.argmax(1)
    top2 = np.sort(pre, axis=1)[:, -2:, :]
    live = pre.max(1) > 0                                              # channels that are 0 for a whole cloud carry no critical point
    clear = ((top2[:, 1] - top2[:, 0]) > 1e-5) & live                  # ... and skip near ties (fp32 vs fp64 may pick either)
    assert clear.sum() > 0.9 * live.sum() and live.mean() > 0.3
    assert np.array_equal(mv > 0, live)
    assert np.array_equal(mi[clear], want_idx[clear])
    # the device bookkeeping on (max, argmax) against the pinned numpy restatement, bit for bit (stable tie order)
    from geometric_adv_amd.defense import get_critical_pc_non_critical_pc
    from oracle.host_defense import critical_and_rest
    got = get_critical_pc_non_critical_pc(pcs, mv, mi)
    want = critical_and_rest(pcs, mv, mi)
    for a, b_ in zip(got, want):
        assert a.dtype == b_.dtype and np.array_equal(a, b_)
    assert (got[2] > 20).all()


def test_critical_cloud_reconstructs_identically_and_defense_runs():
    """The reference's own sanity check (run_defense_critical.py:186-189): the reconstruction of the critical points
    alone equals the reconstruction of the full cloud, exactly."""
    import torch
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.autoencoder import PointNetAE
    from geometric_adv_amd.defense import defend_critical
    from conftest import cloud
    n = 1024
    ae = PointNetAE(W.randomized_weights(n, seed=12), n)
    adv, src = cloud(8, 6, n), cloud(9, 6, n)
    out = defend_critical(ae, adv, src)
    r_full, z_full = ae.forward(adv)
    r_crit, z_crit = ae.forward(out["critical_pc"])
    assert torch.equal(z_full, z_crit) and torch.equal(r_full, r_crit)
    assert (out["critical_num"] > 0).all() and (out["critical_num"] <= 128).all()
    for k in range(len(adv)):                                          # defended = the complement, padded with its last point
        keep = np.setdiff1d(np.arange(n), out["critical_idx"][k, :out["critical_num"][k]])
        assert np.array_equal(out["defended_pc"][k, :len(keep)], adv[k][keep])
    assert out["recon_error_vs_source"].shape == (6,) and np.isfinite(out["recon_error_vs_source"]).all()


def test_run_attack_cli_end_to_end(tmp_path):
    """f-2: the on-disk contract of attacker/run_attack.py on a synthetic eval folder -- same input file names,
    same flags, same per-class outputs; the saved metrics equal a direct AdvAE.attack on the same pairs."""
    import os
    from geometric_adv_amd import run_attack, weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd.attack_data import prepare_data_for_attack
    from geometric_adv_amd.autoencoder import PointNetAE
    from conftest import cloud
    n = 128
    sizes = [4, 5, 4]
    slice_idx = np.concatenate([[0], np.cumsum(sizes)])
    pcs = cloud(17, int(slice_idx[-1]), n)
    w = W.synthetic_weights(n)
    top = tmp_path
    ev = top / "log" / "ae" / "eval"
    os.makedirs(ev)
    W.save_npz(str(top / "log" / "ae" / "weights.npz"), w)
    ae = PointNetAE(w, n)
    latent, loss = ae.transform(pcs), ae.get_loss_per_pc(pcs)
    rng = np.random.default_rng(0)
    nn_idx = np.zeros((len(pcs), len(pcs)), np.int16)
    for s in range(len(pcs)):
        for t in range(3):
            nn_idx[s, slice_idx[t]:slice_idx[t + 1]] = rng.permutation(sizes[t])
    classes = np.array(["chair", "table", "car"])
    attack_idx = np.stack([rng.permutation(4)[:2] for _ in sizes])
    np.save(ev / "point_clouds_test_set_3l.npy", pcs); np.save(ev / "latent_vectors_test_set_3l.npy", latent)
    np.save(ev / "pc_classes_3l.npy", classes); np.save(ev / "slice_idx_test_set_3l.npy", slice_idx)
    np.save(ev / "ae_loss_test_set_3l.npy", loss); np.save(ev / "chamfer_nn_idx_complete_test_set_3l.npy", nn_idx)
    np.save(ev / "sel_idx.npy", attack_idx)
    args = ["--top_dir", str(top), "--ae_folder", "log/ae", "--attack_pc_idx", "log/ae/eval/sel_idx.npy", "--batch_size", "2",
            "--num_iterations", "12", "--num_iterations_thresh", "8", "--num_pc_for_attack", "2", "--num_pc_for_target", "1",
            "--dist_weight_list", "0.5", "2.0", "--class_names", "chair", "car"]
    run_attack.main(args)
    out = ev / "attack_res"
    assert sorted(os.listdir(out)) == ["attack_configuration.json", "car", "chair"]
    for cls in ("chair", "car"):
        m = np.load(out / cls / "adversarial_metrics.npy")
        a = np.load(out / cls / "adversarial_pc_input.npy")
        r = np.load(out / cls / "adversarial_pc_recon.npy")
        assert m.shape == (2, 2, 5) and a.shape == (2, 2, n, 3) and r.shape == (2, 2, n, 3)      # 2 sources x 1 other class x 1 target
        assert np.array_equal(np.load(out / cls / "dist_weight.npy"), [0.5, 2.0])
        assert "Dist weight" in open(out / cls / "attack_stats.txt").read()
    # same numbers as the library call on the same pairs (a fresh AdvAE = fresh Adam state, like one graph per class)
    conf = Configuration(batch_size=2, n_points=n, weights=w, dist_weight_list=[0.5, 2.0], num_iterations=12, num_iterations_thresh=8)
    src, tgt = prepare_data_for_attack(classes, ["chair"], ["chair", "car"], pcs, slice_idx, attack_idx, 1, nn_idx, None)
    _, tl = prepare_data_for_attack(classes, ["chair"], ["chair", "car"], latent, slice_idx, attack_idx, 1, nn_idx, None)
    _, tr = prepare_data_for_attack(classes, ["chair"], ["chair", "car"], loss, slice_idx, attack_idx, 1, nn_idx, None)
    assert len(src) == 2
    m2, a2, r2 = AdvAE("adversary", conf).attack(src, tl, tgt, tr.reshape(-1), conf)
    assert np.array_equal(m2, np.load(out / "chair" / "adversarial_metrics.npy"))
    assert np.array_equal(a2, np.load(out / "chair" / "adversarial_pc_input.npy"))
    # f-1, second half: get_dists_per_point chained on those outputs WITH the reference's sanity check
    # (get_dists_per_point.py:114-115): the Chamfer distance the op recomputes from the saved adversarial clouds must be
    # np.array_equal to the source_chamfer_dist the loop recorded
    from geometric_adv_amd import get_dists_per_point, ops
    import torch
    get_dists_per_point.main(["--top_dir", str(top), "--ae_folder", "log/ae", "--attack_pc_idx", "log/ae/eval/sel_idx.npy",
                              "--do_sanity_checks", "1"])
    for cls in ("chair", "car"):
        d = np.load(out / cls / "adversarial_pc_input_dists.npy")
        a = np.load(out / cls / "adversarial_pc_input.npy")
        assert d.shape == a.shape[:3] and d.dtype == np.float32 and (d >= 0).all()
    s_pc, _ = prepare_data_for_attack(classes, ["chair"], ["chair", "car"], pcs, slice_idx, attack_idx, 1, nn_idx, None)
    first = oracle_first_dists(np.load(out / "chair" / "adversarial_pc_input.npy")[1], s_pc)
    assert np.array_equal(np.load(out / "chair" / "adversarial_pc_input_dists.npy")[1], np.sqrt(first))


def oracle_first_dists(adv, src):
    from oracle.cpu_oracle import Oracle
    return Oracle().nn_distance(adv, src)[0]


def test_chamfer_per_pc_equals_loop_metric_at_full_shape():
    """The reduction-order contract behind that sanity check at configs[1]'s shape (B = 32, N = 2048), where a private
    summation order would show: input_dist / loss_ae of the loop's history == ops.chamfer_per_pc(ops.nn_distance(...)) on the
    loop's own clouds, bit for bit; and within 1e-6 relative of the fp64 mean (north star: 1e-5)."""
    import torch
    from geometric_adv_amd import ops, weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd.autoencoder import PointNetAE
    from conftest import cloud
    b, n = 32, 2048
    w = W.synthetic_weights(n)
    ae = PointNetAE(w, n)
    x, gt = cloud(1002, b, n), cloud(2002, b, n)
    at = AdvAE("adversary", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=4, num_iterations_thresh=1), ae=ae)
    at.set_inputs(x, gt, None, 1.0)
    at.init_pert(None, reset_optimizer=True)
    hist = torch.empty((3, 6, b), device=ae.device)
    at.run(0, 3, 1, hist)
    s = at.peek()
    h = hist.cpu().numpy()[-1]
    xd, gd = torch.as_tensor(x).to(ae.device), torch.as_tensor(gt).to(ae.device)
    a1, _, a2, _ = ops.nn_distance(s["adv"], xd)
    r1, _, r2, _ = ops.nn_distance(s["recon"], gd)
    assert np.array_equal(ops.chamfer_per_pc(a1, a2).cpu().numpy(), h[4])           # input_dist = source_chamfer_dist
    assert np.array_equal(ops.chamfer_per_pc(r1, r2).cpu().numpy(), h[5])           # loss_ae = target_recon_error
    want = a1.double().mean(1) + a2.double().mean(1)
    np.testing.assert_allclose(h[4], want.cpu().numpy(), rtol=1e-6)
    # ragged sizes and n != m through the operator alone
    d1, d2 = torch.rand((3, 777), device=ae.device), torch.rand((3, 1300), device=ae.device)
    np.testing.assert_allclose(ops.chamfer_per_pc(d1, d2).cpu().numpy(), (d1.double().mean(1) + d2.double().mean(1)).cpu().numpy(), rtol=1e-6)


def test_train_checkpoint_attack_defend_pipeline(tmp_path):
    """The widened rows end to end, the way the reference's scripts chain them: train the victim (f-4) on synthetic
    shapes -> models.ckpt-N in TF V2 format (f-2) -> restore it by prefix in the attack class (a12) -> adversarial
    clouds -> off-surface defense (a16/a17) -> reconstruct.  Training must make reconstructions better than at
    initialisation, and the attack must pull the reconstruction of the source towards the target."""
    import torch
    from geometric_adv_amd import tf_checkpoint, defense
    from geometric_adv_amd.trainer import PointNetAETrainer, initial_weights
    from geometric_adv_amd.autoencoder import PointNetAE
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    n, b = 256, 8
    rng = np.random.default_rng(0)

    def shapes(count):                                   # points on randomly scaled ellipsoid shells and boxes
        u = rng.standard_normal((count, n, 3)).astype(np.float32)
        u /= np.linalg.norm(u, axis=2, keepdims=True)
        scale = rng.uniform(0.15, 0.45, size=(count, 1, 3)).astype(np.float32)
        box = rng.random((count, 1, 1)) < 0.5
        cube = np.clip(u * 3.0, -1.0, 1.0)
        return (np.where(box, cube, u) * scale).astype(np.float32)

    data = shapes(64)
    tr = PointNetAETrainer(initial_weights(n, seed=2), n, batch_size=b, learning_rate=0.002)
    first = tr._single_epoch_train(data)[0]
    for _ in range(14):
        last = tr._single_epoch_train(data)[0]
    assert last < 0.5 * first
    prefix = str(tmp_path / "models.ckpt-15")
    tf_checkpoint.write_checkpoint(prefix, tr.export_weights())

    ae = PointNetAE(prefix, n)                           # restore_ae_model by checkpoint prefix, TF-free
    src, tgt = shapes(b), shapes(b)
    conf = Configuration(batch_size=b, n_points=n, weights=prefix, dist_weight_list=[1.0], num_iterations=60,
                         num_iterations_thresh=40, learning_rate=0.01)
    at = AdvAE("adversary", conf, ae=ae)
    ref = ae.get_loss_per_pc(tgt)
    metrics, adv, recon = at.attack(src, ae.transform(tgt), tgt, ref, conf)
    before = ae.get_loss_per_pc(src, tgt)                # chamfer(recon(source), target) without the attack
    assert np.isfinite(metrics).all() and (metrics[0, :, 4] < before).all()
    out = defense.defend_surface(ae, adv[0], src)
    assert out["defended_pc"].shape == (b, n, 3) and np.isfinite(out["defended_recon"]).all()
    assert np.isfinite(out["recon_error_vs_source"]).all()


def test_full_chamfer_matrix_by_symmetry_equals_slices():
    """scorer.get_chamfer_dist_mat_full (upper-triangular block pairs, mirrored) == the reference-shaped column slices, bit for
    bit, for a set size that is not a multiple of the block; two 'ranks' sum to the same matrix."""
    from geometric_adv_amd.scorer import get_chamfer_dist_mat_full, get_chamfer_dist_mat_slice
    from conftest import cloud
    pcs = cloud(23, 21, 192)
    full = get_chamfer_dist_mat_full(pcs, block=8)
    want = np.concatenate([get_chamfer_dist_mat_slice(pcs, s, 7) for s in range(0, 21, 7)], axis=1)
    assert np.array_equal(full, want) and np.array_equal(full, full.T) and not np.diagonal(full).any()
    parts = [get_chamfer_dist_mat_full(pcs, block=8, rank=r, world=2) for r in range(2)]
    assert np.array_equal(parts[0] + parts[1], full)
