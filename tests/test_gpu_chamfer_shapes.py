"""GPU: every nn_distance kernel of the library on the cloud shapes real inputs have -- surfaces (sphere shell), clusters,
duplicated points and lattices (exact distance ties everywhere) -- not only the uniform cubes of the other tests (VERDICT
r02, weak #8): the public op against the pinned C oracle, the paired grid search against the public op, and the attack
loop's own kernels (symmetric scan + finish for (recon, target); grid search / all-pairs for (adv, source)) against the
public op on the loop's clouds.  Distances AND indices bit-equal throughout (tf_nndistance.cpp:21-43: strict '<', lowest
index wins ties)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def make_clouds(kind, seed, b, n):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return (rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5)).astype(np.float32)
    if kind == "sphere":                                   # a surface: points on the unit sphere shell
        v = rng.standard_normal((b, n, 3)).astype(np.float32)
        return (v / np.linalg.norm(v, axis=2, keepdims=True)).astype(np.float32) * np.float32(0.5)
    if kind == "clusters":                                 # two tight clusters far apart
        c = np.where(rng.random((b, n, 1)) < 0.5, np.float32(-0.4), np.float32(0.4))
        return (c + np.float32(0.01) * rng.standard_normal((b, n, 3))).astype(np.float32)
    if kind == "duplicates":                               # every point present 2-4 times: exact distance ties everywhere
        base = (rng.random((b, (n + 2) // 3, 3), dtype=np.float32) - np.float32(0.5)).astype(np.float32)
        idx = rng.integers(0, base.shape[1], (b, n))
        return np.take_along_axis(base, idx[:, :, None].repeat(3, 2), axis=1)
    if kind == "lattice":                                  # coarse lattice: many equidistant neighbours
        return (rng.integers(-4, 5, (b, n, 3)).astype(np.float32) * np.float32(0.125)).astype(np.float32)
    raise ValueError(kind)


KINDS = ["uniform", "sphere", "clusters", "duplicates", "lattice"]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("b,n,m", [(2, 2048, 2048), (1, 8192, 8192), (2, 300, 1000)])
def test_public_op_vs_oracle_on_shapes(oracle, kind, b, n, m):
    from geometric_adv_amd import ops
    a, c = make_clouds(kind, 11, b, n), make_clouds(kind, 12, b, m)
    if kind in ("duplicates", "lattice") and n == m:
        c = c.copy(); c[:, : n // 2] = a[:, : n // 2]                 # and exact zero distances across the two clouds
    got = [t.cpu().numpy() for t in ops.nn_distance(_t(a), _t(c))]
    want = oracle.nn_distance(a, c)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("n,scale", [(2048, 1e-3), (2048, 0.05), (8192, 1e-3)])
def test_paired_grid_search_on_shapes(kind, n, scale):
    """nn_distance_paired(x + pert, x): small perturbations (the search answers) and ones larger than the cells (far queue /
    whole clouds handed to the wave-per-query scan) give the public op's bits."""
    import torch
    from geometric_adv_amd import ops
    x = make_clouds(kind, 21, 3, n)
    pert = (np.random.default_rng(22).standard_normal(x.shape) * scale).astype(np.float32)
    p, q = _t(x + pert), _t(x)
    for g, w in zip(ops.nn_distance_paired(p, q), ops.nn_distance(p, q, kernel="scan")):
        assert torch.equal(g, w)


@pytest.mark.parametrize("kind", ["sphere", "clusters", "duplicates", "lattice"])
@pytest.mark.parametrize("b", [32, 16, 8, 4, 2])
@pytest.mark.parametrize("prune", ["always", False])
def test_attack_loop_indices_on_shapes(kind, b, prune):
    """The loop's four index arrays after a few iterations on non-uniform clouds equal ops.nn_distance on the loop's own clouds, and
    its loss_ae / input_dist rows the means of ops.nn_distance's distances: B = 32 runs the symmetric scan with its row partials
    merged by the loss launch (8 column slices), B = 16 / 8 / 4 the same scan with narrow slices and the row minima folded into
    packed words by 64-bit atomic minima, B = 2 the gated two-scan kernel; prune on / off = grid search / all-pairs."""
    import torch
    from geometric_adv_amd import ops, weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd.autoencoder import PointNetAE
    n = 2048
    w = W.randomized_weights(n)
    ae = PointNetAE(w, n)
    x, gt = make_clouds(kind, 31, b, n), make_clouds(kind, 32, b, n)
    at = AdvAE("a", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=4, num_iterations_thresh=2, learning_rate=0.01,
                                  chamfer_prune=prune), ae=ae)
    at.set_inputs(x, gt, ae.transform(gt), 1.0)
    at.init_pert(None, reset_optimizer=True)
    hist = torch.empty((4, 6, b), device=ae.device)
    at.run(0, 4, 2, hist)
    p = at.peek()
    d1, i1, d2, i2 = ops.nn_distance(p["recon"], _t(gt), kernel="scan")     # the public op's own kernel, not the loop's
    assert torch.equal(p["idx_r1"], i1) and torch.equal(p["idx_r2"], i2)
    e1, j1, e2, j2 = ops.nn_distance(p["adv"], _t(x), kernel="scan")
    assert torch.equal(p["idx_a1"], j1) and torch.equal(p["idx_a2"], j2)
    # the metric rows of the last iteration are sums of the same distances (fp32 sums in the loop's order: compare to fp64 means)
    h = hist[-1].cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(h[5], (d1.double().mean(1) + d2.double().mean(1)).cpu().numpy(), rtol=2e-6)
    np.testing.assert_allclose(h[4], (e1.double().mean(1) + e2.double().mean(1)).cpu().numpy(), rtol=2e-6, atol=1e-12)


@pytest.mark.parametrize("kernel", ["symmetric", "auto"])
@pytest.mark.parametrize("b,n,m", [(64, 1024, 8192), (1024, 256, 2048), (512, 1024, 1024), (128, 512, 4096), (40, 2048, 8192)])
def test_symmetric_scan_tie_rule_at_large_launches(oracle, kernel, b, n, m):
    """Launch shapes large enough for several column stages per workgroup and, below 1025 rows, several column-waves (ADVICE
    r05: a stage re-split over the column-waves broke 'lowest index wins'): duplicated points give exact ties in every
    direction; sampled clouds against the pinned oracle, indices and distance bits."""
    from geometric_adv_amd import ops
    a, c = make_clouds("duplicates", 41, b, n), make_clouds("duplicates", 42, b, m)
    got = [t.cpu().numpy() for t in ops.nn_distance(_t(a), _t(c), kernel=kernel)]
    pick = np.unique(np.r_[0, 1, b // 2, b - 1, np.random.default_rng(43).integers(0, b, 4)])
    want = oracle.nn_distance(a[pick], c[pick])
    for g, w in zip(got, want):
        assert np.array_equal(g[pick], w)
