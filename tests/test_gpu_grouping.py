"""GPU parity of the grouping ops against the golden vectors (reference CPU twins) and the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def test_selection_sort_golden(golden_grouping):
    from geometric_adv_amd import ops
    g = golden_grouping
    for name in ["kat", "rnd", "tie", "full", "swap"]:
        k = int(g[f"{name}_k"])
        idx, val = ops.select_top_k(k, _t(g[f"{name}_dist"]))
        assert np.array_equal(idx.cpu().numpy(), g[f"{name}_idx"]), name      # ALL n entries, not just the first k
        assert np.array_equal(val.cpu().numpy(), g[f"{name}_val"]), name


def test_query_ball_and_group_point_golden(golden_grouping):
    from geometric_adv_amd import ops
    g = golden_grouping
    idx, cnt = ops.query_ball_point(float(g["qb_radius"]), int(g["qb_nsample"]), _t(g["qb_xyz1"]), _t(g["qb_xyz2"]))
    assert np.array_equal(idx.cpu().numpy(), g["qb_idx"])
    out = ops.group_point(_t(g["gp_points"]), _t(g["qb_idx"]))
    assert np.array_equal(out.cpu().numpy(), g["gp_out"])
    gp = ops.group_point_grad(_t(g["gp_points"]), _t(g["qb_idx"]), _t(g["gp_grad_out"]))
    assert np.array_equal(gp.cpu().numpy(), g["gp_grad_points"])            # same accumulation order => same bits


@pytest.mark.parametrize("b,n,c,m,ns,skew", [(3, 300, 5, 77, 9, "uniform"), (2, 2048, 64, 2048, 32, "uniform"), (2, 1000, 3, 900, 16, "few"),
                                             (1, 130, 2, 4096, 16, "one"), (2, 63, 4, 50, 3, "uniform"), (1, 5000, 1, 333, 7, "runs")])
def test_group_point_grad_vs_oracle_bit_exact(oracle, b, n, c, m, ns, skew):
    """The counting-sort scatter against the CPU twin (test/query_ball_point.cpp:70-84), bit for bit: float sums in entry
    order.  Destinations uniform, concentrated on a few points (long per-point lists), all on ONE point (every entry of a
    64-entry group is a peer of every other), or in runs (query_ball_point's padding repeats the first hit); point counts
    that are not multiples of the 64-point wave ranges; more points than one scan pass of 1024."""
    from geometric_adv_amd import ops
    rng = np.random.default_rng(n + m)
    if skew == "uniform":
        idx = rng.integers(0, n, size=(b, m, ns))
    elif skew == "few":
        idx = rng.choice(np.array([0, 1, 63, 64, 65, n - 1]), size=(b, m, ns))
    elif skew == "one":
        idx = np.full((b, m, ns), n - 1)
    else:
        idx = np.repeat(rng.integers(0, n, size=(b, m, 1)), ns, axis=2)
    idx = idx.astype(np.int32)
    points = rng.standard_normal((b, n, c)).astype(np.float32)
    grad_out = rng.standard_normal((b, m, ns, c)).astype(np.float32)
    want = oracle.group_point_grad(points, idx, grad_out)
    got = ops.group_point_grad(_t(points), _t(idx), _t(grad_out)).cpu().numpy()
    assert np.array_equal(got, want)


def test_query_ball_vs_oracle_sparse_hits(oracle):
    from geometric_adv_amd import ops
    from conftest import cloud
    x1, x2 = cloud(1, 3, 300) + 0.5, cloud(2, 3, 50) + 0.5
    for radius, ns in [(0.05, 4), (0.2, 16), (2.0, 7)]:
        want_idx, want_cnt = oracle.query_ball_point(radius, ns, x1, x2)
        idx, cnt = ops.query_ball_point(radius, ns, _t(x1), _t(x2))
        cnt = cnt.cpu().numpy()
        assert np.array_equal(cnt, want_cnt)
        hit = want_cnt > 0                                                  # rows without any hit are left untouched by both
        assert np.array_equal(idx.cpu().numpy()[hit], want_idx[hit])


def test_query_ball_radius_boundary_and_large(oracle):
    """The kernel decides `max(sqrtf(d2), 1e-20) < radius` from d2 alone (largest float whose square root is below the
    radius): exact at the boundary -- lattice distances that EQUAL the radius are not hits, one ulp more radius makes them
    hits -- and for a dataset that spans several LDS tiles with early exits at nsample."""
    from geometric_adv_amd import ops
    from conftest import cloud
    rng = np.random.default_rng(5)
    lat = (rng.integers(-3, 4, size=(2, 400, 3)) * 0.25).astype(np.float32)
    for radius in (0.25, float(np.nextafter(np.float32(0.25), np.float32(1))), float(np.nextafter(np.float32(0.25), np.float32(0))),
                   0.4330127, 1e-21, 1e30):
        want_idx, want_cnt = oracle.query_ball_point(radius, 8, lat, lat[:, :60])
        idx, cnt = ops.query_ball_point(radius, 8, _t(lat), _t(np.ascontiguousarray(lat[:, :60])))
        assert np.array_equal(cnt.cpu().numpy(), want_cnt), radius
        hit = want_cnt > 0
        assert np.array_equal(idx.cpu().numpy()[hit], want_idx[hit]), radius
    x1, x2 = cloud(7, 2, 2500) + 0.5, cloud(8, 2, 300) + 0.5
    for radius, ns in [(0.1, 32), (0.3, 5)]:
        want_idx, want_cnt = oracle.query_ball_point(radius, ns, x1, x2)
        idx, cnt = ops.query_ball_point(radius, ns, _t(x1), _t(x2))
        assert np.array_equal(cnt.cpu().numpy(), want_cnt)
        hit = want_cnt > 0
        assert np.array_equal(idx.cpu().numpy()[hit], want_idx[hit])


@pytest.fixture(params=["all_points", "grid", "grid_shells"])
def knn_kernel(request):
    """The k-NN kernels on the same inputs: the all-points scan, the exact grid search (lane-private pass + leftovers) and the
    grid search's wave-uniform shell walk alone (ops.knn_grid_mode)."""
    from geometric_adv_amd import ops
    ops.knn_grid_mode(request.param)
    yield request.param
    ops.knn_grid_mode("auto")


@pytest.mark.parametrize("b,n,m,k", [(2, 200, 77, 9), (1, 2048, 64, 9), (3, 64, 64, 64), (2, 500, 10, 1), (2, 3000, 300, 16),
                                     (2, 1500, 257, 4), (1, 5, 5, 5), (2, 2048, 100, 2), (1, 1100, 40, 17)])
def test_knn_point_vs_oracle(oracle, knn_kernel, b, n, m, k):
    from geometric_adv_amd import ops
    from conftest import cloud
    x1, x2 = cloud(10 + n, b, n), cloud(20 + m, b, m)
    want_val, want_idx = oracle.knn_point(k, x1, x2)
    val, idx = ops.knn_point(k, _t(x1), _t(x2))
    assert np.array_equal(idx.cpu().numpy(), want_idx)
    assert np.array_equal(val.cpu().numpy(), want_val)


def test_knn_point_exact_ties_follow_the_swap_rule(oracle, knn_kernel):
    """Lattice clouds: many exactly equal distances; the order among them is the reference's."""
    from geometric_adv_amd import ops
    rng = np.random.default_rng(3)
    x = (rng.integers(-2, 3, size=(2, 150, 3)) * 0.25).astype(np.float32)
    want_val, want_idx = oracle.knn_point(9, x, x)
    val, idx = ops.knn_point(9, _t(x), _t(x))
    assert np.array_equal(idx.cpu().numpy(), want_idx)
    assert np.array_equal(val.cpu().numpy(), want_val)


def test_knn_mixed_ties_and_infinities(oracle, knn_kernel):
    """One launch, both paths: most queries have a unique answer (register top-k), the ones with duplicated neighbours or
    non-finite distances are handed to the selection-sort kernel -- the results are the reference's throughout."""
    from geometric_adv_amd import ops
    from conftest import cloud
    n = 700
    x = cloud(33, 3, n)
    x[:, 100:160] = x[:, 300:360]                 # 60 duplicated points: ties for them and their neighbours
    x[1, 5, 0] = np.inf                           # one point at infinity: NaN / inf distances
    q = np.concatenate([x[:, :200], cloud(34, 3, 50)], axis=1)
    for k in (3, 8):
        want_val, want_idx = oracle.knn_point(k, x, q)
        val, idx = ops.knn_point(k, _t(x), _t(q))
        assert np.array_equal(idx.cpu().numpy(), want_idx)
        assert np.array_equal(val.cpu().numpy(), want_val, equal_nan=True)
    x[1, 5, 0] = 0.25
    assert np.array_equal(ops.knn_dists(_t(x), 8).cpu().numpy(), oracle.knn_dists(x, 8))


def test_knn_dists_vs_oracle_and_defense(oracle, knn_kernel):
    from geometric_adv_amd import ops, weights as W
    from geometric_adv_amd.autoencoder import PointNetAE
    from geometric_adv_amd.defense import defend_surface, get_outlier_pc_inlier_pc
    from conftest import cloud
    n = 256
    pc = cloud(5, 6, n)
    got = ops.knn_dists(_t(pc), 8).cpu().numpy()
    assert np.array_equal(got, oracle.knn_dists(pc, 8))
    # a defended cloud: push 10 points far off the surface -> they are the outliers
    adv = pc.copy()
    adv[:, :10] += 3.0 * (np.arange(10, dtype=np.float32)[None, :, None] + 1)      # each far from everything else
    ae = PointNetAE(W.synthetic_weights(n), n)
    out = defend_surface(ae, adv, pc, knn_dist_thresh=0.5)
    assert (out["outlier_num"] == 10).all()
    assert np.array_equal(out["outlier_idx"][:, :10], np.tile(np.arange(10, dtype=np.int16), (6, 1)))
    assert np.array_equal(out["defended_pc"][:, :n - 10], adv[:, 10:])
    assert np.array_equal(out["defended_pc"][:, n - 10:], np.repeat(adv[:, -1:], 10, axis=1))   # last inlier duplicated
    assert out["recon_error_vs_source"].shape == (6,)
    # the fused score + packing kernel against the pinned numpy restatement fed with numpy's own mean
    from oracle.host_defense import outlier_inlier
    score = out["knn_dists"][:, :, :2].mean(2)
    want = outlier_inlier(adv, score, 0.5)
    assert np.array_equal(want[3], out["defended_pc"]) and np.array_equal(want[1], out["outlier_idx"])
    got = get_outlier_pc_inlier_pc(adv, score, 0.5)                    # the reference function's own signature (per-point scalar)
    for a, b_ in zip(got, want):
        assert a.dtype == b_.dtype and np.array_equal(a, b_)


def test_knn_full_size_properties(knn_kernel):
    """Config-3 size (n = 2048): properties instead of a CPU re-run -- the self distance is 0 and
    comes first, columns ascend, and the result is invariant to where in the batch a cloud sits."""
    import torch
    from geometric_adv_amd import ops
    from conftest import cloud
    pc = _t(cloud(9, 16, 2048))
    val, idx = ops.knn_point(9, pc, pc)
    ar = torch.arange(2048, device="cuda:0", dtype=torch.int32).expand(16, -1)
    assert torch.equal(idx[:, :, 0], ar) and not val[:, :, 0].any()
    assert (val[:, :, 1:] >= val[:, :, :-1]).all()
    v2, i2 = ops.knn_point(9, pc[3:5].contiguous(), pc[3:5].contiguous())
    assert torch.equal(v2, val[3:5]) and torch.equal(i2, idx[3:5])
    d = ops.knn_dists(pc, 8)
    torch.testing.assert_close(d, val[:, :, 1:].sqrt(), rtol=1e-6, atol=0)


@pytest.mark.parametrize("kind", ["surface", "planar", "line", "clustered_with_outliers", "far_queries", "shifted", "tiny_extent"])
def test_knn_grid_on_awkward_geometry(oracle, kind):
    """The grid search where a uniform grid is a poor fit -- bit-exact against the oracle all the same: a thin shell, a flat
    cloud (one degenerate axis: a single layer of cells), a line (two), one tight cluster plus a few far outliers (nearly all
    points in one cell: the search degenerates to the all-points scan), queries far outside the dataset's box, coordinates
    around 1000 (the slack on the face distances follows the magnitude), and an extent of 1e-4."""
    from geometric_adv_amd import ops
    rng = np.random.default_rng(len(kind))
    b, n, m = 2, 1500, 700
    x = rng.standard_normal((b, n, 3)).astype(np.float32)
    if kind == "surface":
        x = (0.4 * x / np.linalg.norm(x, axis=2, keepdims=True)).astype(np.float32)
    elif kind == "planar":
        x[:, :, 2] = np.float32(0.25)
    elif kind == "line":
        x[:, :, 1:] = np.float32(-0.5)
    elif kind == "clustered_with_outliers":
        x *= np.float32(0.01)
        x[:, :5] += np.float32(50.0) * (np.arange(5, dtype=np.float32)[None, :, None] + 1)
    elif kind == "shifted":
        x = (x * np.float32(0.2) + np.float32(1000.0)).astype(np.float32)
    elif kind == "tiny_extent":
        x = (x * np.float32(1e-4)).astype(np.float32)
    q = x[:, :m].copy() if kind != "far_queries" else (rng.standard_normal((b, m, 3)) * 30).astype(np.float32)
    q[:, ::2] += (rng.standard_normal((b, (m + 1) // 2, 3)) * 0.01).astype(np.float32)
    try:
        for mode in ("grid", "grid_shells"):
            ops.knn_grid_mode(mode)
            for k in (1, 9, 16):
                want_val, want_idx = oracle.knn_point(k, x, q)
                val, idx = ops.knn_point(k, _t(x), _t(q))
                assert np.array_equal(val.cpu().numpy(), want_val), (kind, k, mode)
                assert np.array_equal(idx.cpu().numpy(), want_idx), (kind, k, mode)
            assert np.array_equal(ops.knn_dists(_t(x), 8).cpu().numpy(), oracle.knn_dists(x, 8)), (kind, mode)
    finally:
        ops.knn_grid_mode("auto")


def test_knn_grid_equals_all_points_at_config_size():
    """B = 24 x N = 2048 (config 2's cloud size), uniform and surface-like clouds with a few far outliers -- the defense's
    input: the two kernels agree bit for bit (values and indices), and so does the 4096-point grid (16 cells per axis, 512-thread
    workgroups)."""
    import torch
    from geometric_adv_amd import ops
    from conftest import cloud
    for n, b in ((2048, 24), (4096, 5)):
        pc = cloud(77, b, n)
        v = np.random.default_rng(1).standard_normal((b // 2, n, 3)).astype(np.float32)
        pc[: b // 2] = 0.4 * v / np.linalg.norm(v, axis=2, keepdims=True)
        pc[:, :20] *= 3.0
        pc = _t(pc)
        out = {}
        for mode in ("all_points", "grid", "grid_shells"):
            ops.knn_grid_mode(mode)
            out[mode] = (ops.knn_dists(pc, 8), ) + tuple(ops.knn_point(9, pc, pc))
        ops.knn_grid_mode("auto")
        for other in ("grid", "grid_shells"):
            for a, g in zip(out["all_points"], out[other]):
                assert torch.equal(a, g), other


def test_knn_grid_many_leftover_queries():
    """knn_dists-style searches (values only: the lane-private walk) where most queries are NOT closed by the 27 cells around
    them -- queries far outside the cloud, a cloud that is two distant clusters plus stragglers --: more leftovers than a
    workgroup's list holds (the overflow goes to the exact redo kernel), several cooperative all-points chunks per workgroup.
    Same values as the all-points kernel, bit for bit."""
    import torch
    from geometric_adv_amd import ops
    rng = np.random.default_rng(99)
    b, n = 3, 2048
    x = rng.random((b, n, 3), dtype=np.float32)
    x[:, : n // 2] = x[:, : n // 2] * np.float32(0.05)                       # cluster A
    x[:, n // 2: n - 40] = x[:, n // 2: n - 40] * np.float32(0.05) + np.float32(5.0)   # cluster B, far away
    x[:, n - 40:] = rng.standard_normal((b, 40, 3)).astype(np.float32) * np.float32(20.0)   # stragglers
    out = {}
    for mode in ("all_points", "grid", "grid_shells"):
        ops.knn_grid_mode(mode)
        try:
            out[mode] = ops.knn_dists(_t(x), 8)
        finally:
            ops.knn_grid_mode("auto")
    assert torch.equal(out["all_points"], out["grid"]) and torch.equal(out["all_points"], out["grid_shells"])
    # queries far from everything, many more than KG_LEFT_CAP per workgroup (values + indices ride the shell walk; the values-only
    # path is reached through knn_dists only, so the far queries go in as a second "cloud" of the same call)
    y = np.concatenate([rng.random((b, 600, 3), dtype=np.float32), rng.standard_normal((b, n - 600, 3)).astype(np.float32) * np.float32(50.0)], axis=1)
    for mode in ("all_points", "grid"):
        ops.knn_grid_mode(mode)
        try:
            out[mode] = ops.knn_dists(_t(y), 8)
        finally:
            ops.knn_grid_mode("auto")
    assert torch.equal(out["all_points"], out["grid"])


def test_knn_grid_more_queries_than_the_build_kernel_keeps_in_registers():
    """5000 queries against 1000 / 4096 points: the grid's build kernel holds the first 4096 queries in registers and orders them
    through LDS, the rest goes the long way; values and indices as the all-points kernel's."""
    import torch
    from geometric_adv_amd import ops
    rng = np.random.default_rng(5)
    for n in (1000, 4096):
        x = rng.random((2, n, 3), dtype=np.float32)
        q = (rng.random((2, 5000, 3), dtype=np.float32) * np.float32(1.2) - np.float32(0.1))
        out = {}
        for mode in ("all_points", "grid"):
            ops.knn_grid_mode(mode)
            try:
                out[mode] = ops.knn_point(4, _t(x), _t(q))
            finally:
                ops.knn_grid_mode("auto")
        for a, g in zip(out["all_points"], out["grid"]):
            assert torch.equal(a, g), n


def test_grouping_argument_errors():
    import torch
    from geometric_adv_amd import ops
    x = torch.rand((2, 10, 3), device="cuda:0")
    with pytest.raises(ValueError):
        ops.query_ball_point(-1.0, 4, x, x)
    with pytest.raises(ValueError):
        ops.select_top_k(0, torch.rand((2, 3, 5), device="cuda:0"))
    with pytest.raises(ValueError):
        ops.knn_point(11, x, x)


def test_knn_grid_equals_all_points_randomised():
    """Thirty random problems -- batch, point counts (dataset != queries, not multiples of anything), k, anisotropic scales,
    clusters, duplicated points (ties: both kernels hand those queries to the same redo kernel), a NaN coordinate in one cloud --:
    the grid search and the all-points kernel return the same values and indices, bit for bit."""
    import torch
    from geometric_adv_amd import ops
    rng = np.random.default_rng(2024)
    for trial in range(30):
        b = int(rng.integers(1, 5)); n = int(rng.integers(70, 4097)); m = int(rng.integers(1, 1500)); k = int(rng.integers(1, min(16, n) + 1))
        x = rng.standard_normal((b, n, 3)).astype(np.float32) * rng.uniform(0.01, 3.0, size=(b, 1, 3)).astype(np.float32)
        if trial % 3 == 0:
            x[:, : n // 2] = (x[:, : n // 2] * 0.02 + rng.standard_normal((b, 1, 3))).astype(np.float32)      # a tight cluster
        if trial % 4 == 1:
            x[:, 10:30] = x[:, 40:60]                                                                       # duplicated points
        q = np.concatenate([x[:, : m // 2], rng.standard_normal((b, m - m // 2, 3)).astype(np.float32) * 2.0], axis=1)
        if trial == 7:
            x[0, 3, 1] = np.nan
        out = {}
        for mode in ("all_points", "grid", "grid_shells"):
            ops.knn_grid_mode(mode)
            try:
                out[mode] = ops.knn_point(k, _t(x), _t(q))
            finally:
                ops.knn_grid_mode("auto")
        for other in ("grid", "grid_shells"):
            for a, g in zip(out["all_points"], out[other]):
                assert torch.equal(a.nan_to_num(nan=-1.0), g.nan_to_num(nan=-1.0)), (trial, b, n, m, k, other)


def test_knn_point_keyed_lists_near_equal_distances(oracle):
    """knn_point's grid search keeps ONE word per candidate (distance bits with the low 12 mantissa bits replaced by the position in
    the sorted cloud) and recomputes the exact distances of the <= k + 1 positions a finished list names; a query whose k-th exact
    distance is not below the truncated value of the list's last key goes to the redo list.  Clouds built to land there: thin
    spherical shells around the queries (hundreds of neighbours within 1e-4 relative of each other), concentric shells a few
    float ulps apart, and a uniform cloud for contrast -- values and indices as the oracle's, bit for bit, in every kernel."""
    import torch
    from geometric_adv_amd import ops
    rng = np.random.default_rng(77)
    b, n, m, k = 2, 2048, 64, 9
    u = rng.standard_normal((b, n, 3)).astype(np.float32)
    u /= np.linalg.norm(u, axis=2, keepdims=True)
    centres = (rng.random((b, 8, 3), dtype=np.float32) - 0.5).astype(np.float32)
    x = np.empty((b, n, 3), np.float32)
    for j in range(8):                                          # 8 shells of 256 points, radius 0.2 * (1 +- 5e-5)
        r = (0.2 * (1.0 + 5e-5 * rng.standard_normal((b, 256, 1)))).astype(np.float32)
        x[:, 256 * j:256 * (j + 1)] = centres[:, j:j + 1] + r * u[:, 256 * j:256 * (j + 1)]
    q = np.concatenate([centres, (rng.random((b, m - 8, 3), dtype=np.float32) - 0.5).astype(np.float32)], axis=1)
    want_val, want_idx = oracle.knn_point(k, x, q)
    for kern in ("all_points", "grid", "grid_shells"):
        val, idx = ops.knn_point(k, _t(x), _t(q), kernel=kern)
        assert np.array_equal(idx.cpu().numpy(), want_idx), kern
        assert np.array_equal(val.cpu().numpy(), want_val), kern
    # a uniform cloud at the defense's shape, every kernel against the all-points one, k = 1 .. 16
    x = (rng.random((3, 2048, 3), dtype=np.float32) - 0.5).astype(np.float32)
    for k in (1, 2, 4, 8, 9, 10, 12, 16):
        ref = ops.knn_point(k, _t(x), _t(x), kernel="all_points")
        for kern in ("grid", "grid_shells"):
            got = ops.knn_point(k, _t(x), _t(x), kernel=kern)
            assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]), (k, kern)
