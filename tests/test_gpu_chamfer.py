"""GPU parity: nn_distance / nn_distance_grad through the C ABI vs the golden vectors (reference
CPU functions) and vs the oracle on seeded inputs.  Bit-exact: indices AND float outputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_nn_distance_golden(dev, golden_nn):
    from geometric_adv_amd import ops
    g = golden_nn
    for name in g["cases"]:
        d1, i1, d2, i2 = ops.nn_distance(_t(g[f"{name}_xyz1"], dev), _t(g[f"{name}_xyz2"], dev))
        assert np.array_equal(i1.cpu().numpy(), g[f"{name}_idx1"]), name
        assert np.array_equal(i2.cpu().numpy(), g[f"{name}_idx2"]), name
        assert np.array_equal(_bits(d1.cpu().numpy()), _bits(g[f"{name}_dist1"])), name
        assert np.array_equal(_bits(d2.cpu().numpy()), _bits(g[f"{name}_dist2"])), name


def test_nn_distance_grad_golden(dev, golden_nn):
    from geometric_adv_amd import ops
    g = golden_nn
    for name in g["cases"]:
        g1, g2 = ops.nn_distance_grad(_t(g[f"{name}_xyz1"], dev), _t(g[f"{name}_xyz2"], dev),
                                      _t(g[f"{name}_gd1"], dev), _t(g[f"{name}_idx1"], dev),
                                      _t(g[f"{name}_gd2"], dev), _t(g[f"{name}_idx2"], dev))
        assert np.array_equal(_bits(g1.cpu().numpy()), _bits(g[f"{name}_gxyz1"])), name
        assert np.array_equal(_bits(g2.cpu().numpy()), _bits(g[f"{name}_gxyz2"])), name


@pytest.mark.parametrize("kernel", ["scan", "auto"])
@pytest.mark.parametrize("b,n,m", [(4, 2048, 2048), (2, 4500, 777), (3, 100, 5000), (1, 8192, 8192), (70, 33, 65)])
def test_nn_distance_vs_oracle(dev, oracle, b, n, m, kernel):
    """(kernel "auto": the 8192 x 8192 case is large enough for the symmetric scan; "scan": always the reference-shaped entry point)"""
    from geometric_adv_amd import ops
    from conftest import cloud
    x1, x2 = cloud(100 + n, b, n), cloud(200 + m, b, m)
    # near-coincident clouds (the adv-vs-source regime): plenty of near ties
    if n == m:
        x2 = (x1 + np.float32(1e-4) * cloud(7, b, n)).astype(np.float32)
    want = oracle.nn_distance(x1, x2)
    got = ops.nn_distance(_t(x1, dev), _t(x2, dev), kernel=kernel)
    for w, gt, what in zip(want, got, ["dist1", "idx1", "dist2", "idx2"]):
        gt = gt.cpu().numpy()
        if w.dtype == np.int32:
            assert np.array_equal(gt, w), what
        else:
            assert np.array_equal(_bits(gt), _bits(w)), what
    rng = np.random.default_rng(1)
    gd1 = rng.standard_normal((b, n)).astype(np.float32)
    gd2 = rng.standard_normal((b, m)).astype(np.float32)
    wg1, wg2 = oracle.nn_distance_grad(x1, x2, gd1, want[1], gd2, want[3])
    g1, g2 = ops.nn_distance_grad(_t(x1, dev), _t(x2, dev), _t(gd1, dev), got[1], _t(gd2, dev), got[3])
    assert np.array_equal(_bits(g1.cpu().numpy()), _bits(wg1))
    assert np.array_equal(_bits(g2.cpu().numpy()), _bits(wg2))


def test_nn_distance_random_shapes_vs_oracle(dev, oracle):
    """Forty random (b, n, m) from one point to a few thousand, with exact duplicates and lattice coordinates (ties) mixed in:
    every register blocking / workgroup shape the launcher picks -- 4 or 16 waves, 1, 2 or 4 queries per lane, ragged last
    tiles and chunks, several LDS stages -- against the pinned oracle, bit for bit."""
    from geometric_adv_amd import ops
    rng = np.random.default_rng(2024)
    for trial in range(40):
        b = int(rng.integers(1, 40)) if trial % 4 else int(rng.integers(1, 4))
        n = int(rng.integers(1, 300)) if trial % 3 == 0 else int(rng.integers(1, 3000))
        m = int(rng.integers(1, 300)) if trial % 5 == 0 else int(rng.integers(1, 5000))
        if b * max(n, m) > 60000:
            b = max(1, 60000 // max(n, m))
        x1 = rng.random((b, n, 3), dtype=np.float32)
        x2 = rng.random((b, m, 3), dtype=np.float32)
        if trial % 2:                                                    # a coarse lattice: many exactly equal distances
            x1, x2 = np.round(x1 * 4) / 4, np.round(x2 * 4) / 4
        if trial % 7 == 0 and m > 1:                                     # duplicated targets: the lowest index must win
            x2[:, m // 2:] = x2[:, : m - m // 2]
        x1, x2 = x1.astype(np.float32), x2.astype(np.float32)
        want = oracle.nn_distance(x1, x2)
        got = ops.nn_distance(_t(x1, dev), _t(x2, dev), kernel="scan")
        for w, gt, what in zip(want, got, ["dist1", "idx1", "dist2", "idx2"]):
            assert np.array_equal(_bits(gt.cpu().numpy()), _bits(w)), (trial, b, n, m, what)


def test_symmetric_scan_random_shapes_vs_oracle(dev, oracle):
    """The attack loop's symmetric scan as an operator (geoadv_nn_distance_sym): sixty random (b, n, m) -- one point to a few
    thousand rows and columns, so every layout of a workgroup's 8 waves over rows and columns (1 / 2 / 4 / 8 row-waves), one and
    several row super-tiles (n > 2048), column slices of 64 / 128 / 256, ragged last rows, rounds and slices -- with exact
    duplicates (rows AND columns: the lowest index must win on both sides) and lattice coordinates mixed in; bit for bit."""
    from geometric_adv_amd import ops
    rng = np.random.default_rng(515)
    for trial in range(60):
        b = int(rng.integers(1, 40)) if trial % 4 else int(rng.integers(1, 4))
        n = int(rng.integers(1, 300)) if trial % 3 == 0 else int(rng.integers(1, 5000))
        m = int(rng.integers(1, 300)) if trial % 5 == 0 else int(rng.integers(1, 5000))
        if b * max(n, m) > 60000:
            b = max(1, 60000 // max(n, m))
        x1 = rng.random((b, n, 3), dtype=np.float32)
        x2 = rng.random((b, m, 3), dtype=np.float32)
        if trial % 2:                                                    # a coarse lattice: many exactly equal distances
            x1, x2 = np.round(x1 * 4) / 4, np.round(x2 * 4) / 4
        if trial % 7 == 0 and m > 1:                                     # duplicated columns
            x2[:, m // 2:] = x2[:, : m - m // 2]
        if trial % 7 == 3 and n > 1:                                     # duplicated rows, far apart (other waves / super-tiles)
            x1[:, n // 2:] = x1[:, : n - n // 2]
        x1, x2 = x1.astype(np.float32), x2.astype(np.float32)
        want = oracle.nn_distance(x1, x2)
        got = ops.nn_distance_sym(_t(x1, dev), _t(x2, dev))
        for w, gt, what in zip(want, got, ["dist1", "idx1", "dist2", "idx2"]):
            assert np.array_equal(_bits(gt.cpu().numpy()), _bits(w)), (trial, b, n, m, what)


@pytest.mark.parametrize("b,n,m", [(32, 2048, 2048), (8, 2048, 2048), (2, 8192, 8192), (1, 2049, 4097), (64, 1024, 1024), (3, 300, 6000)])
def test_symmetric_scan_attack_shapes_vs_oracle(dev, oracle, b, n, m):
    """The shapes the loop and the scorer run it at (B = 32 / 8 x 2048: 256- and 64-column slices; 8192: four row super-tiles;
    one row over a super-tile; four row-waves x two column-waves), near-coincident clouds where n == m (the regime of near ties)."""
    from geometric_adv_amd import ops
    from conftest import cloud
    x1, x2 = cloud(300 + n, b, n), cloud(400 + m, b, m)
    if n == m:
        x2 = (x1 + np.float32(1e-4) * cloud(7, b, n)).astype(np.float32)
    if b > 8:                                                            # (the oracle is a single-threaded scan: check 8 clouds)
        sel = np.linspace(0, b - 1, 8).astype(int)
    else:
        sel = np.arange(b)
    want = oracle.nn_distance(x1[sel], x2[sel])
    got = ops.nn_distance_sym(_t(x1, dev), _t(x2, dev))
    for w, gt, what in zip(want, got, ["dist1", "idx1", "dist2", "idx2"]):
        assert np.array_equal(_bits(gt.cpu().numpy()[sel]), _bits(w)), what
    ref = ops.nn_distance(_t(x1, dev), _t(x2, dev), kernel="scan")       # and every cloud against the public op's own kernel
    for a_, b_ in zip(ref, got):
        assert np.array_equal(_bits(a_.cpu().numpy()), _bits(b_.cpu().numpy()))


def test_nn_distance_full_size_properties(dev):
    """Config-2 size (B=32, N=2048): size-independent properties instead of a CPU re-run:
    a cloud against itself gives idx = identity / dist = 0; against a permuted copy gives the
    inverse permutation; results do not depend on the batch a cloud sits in."""
    import torch
    from geometric_adv_amd import ops
    from conftest import cloud
    x = _t(cloud(3, 32, 2048), dev)
    d1, i1, d2, i2 = ops.nn_distance(x, x)
    ar = torch.arange(2048, device=dev, dtype=torch.int32).expand(32, -1)
    assert torch.equal(i1, ar) and torch.equal(i2, ar)
    assert not d1.any() and not d2.any()
    perm = torch.randperm(2048, device=dev)
    y = x[:, perm].contiguous()
    d1, i1, d2, i2 = ops.nn_distance(x, y)
    inv = torch.empty_like(perm); inv[perm] = torch.arange(2048, device=dev)
    assert torch.equal(i1.long(), inv.expand(32, -1))
    assert torch.equal(i2.long(), perm.expand(32, -1))
    z = _t(cloud(4, 32, 2048), dev)
    full = ops.nn_distance(x, z)
    part = ops.nn_distance(x[5:7].contiguous(), z[5:7].contiguous())
    for a, b_ in zip(full, part):
        assert torch.equal(a[5:7], b_)


def test_nn_distance_edge_cases(dev):
    import torch
    from geometric_adv_amd import ops
    e = torch.empty((2, 0, 3), device=dev)
    x = torch.rand((2, 5, 3), device=dev)
    d1, i1, d2, i2 = ops.nn_distance(x, e)          # no targets: the CPU loop leaves 0 / 0
    assert d1.shape == (2, 5) and not d1.any() and not i1.any() and d2.shape == (2, 0)
    d1, i1, d2, i2 = ops.nn_distance(torch.empty((0, 4, 3), device=dev), torch.empty((0, 4, 3), device=dev))
    assert d1.shape == (0, 4)
    with pytest.raises(ValueError):
        ops.nn_distance(torch.rand((2, 5, 2), device=dev), x)
    with pytest.raises(ValueError):
        ops.nn_distance(torch.rand((3, 5, 3), device=dev), x)
    with pytest.raises(ValueError):
        ops.nn_distance(torch.rand((2, 5, 3)), torch.rand((2, 5, 3)))   # CPU tensors: no CPU path


def test_nn_distance_autograd(dev):
    """The registered gradient (tf_nndistance.py:35-41) through torch autograd."""
    import torch
    from geometric_adv_amd import ops
    a = torch.rand((2, 50, 3), device=dev, requires_grad=True)
    b_ = torch.rand((2, 60, 3), device=dev, requires_grad=True)
    d1, i1, d2, i2 = ops.nn_distance_autograd(a, b_)
    (d1.mean(1) + d2.mean(1)).sum().backward()
    # same thing with torch's own ops on the gathered pairs
    a2 = a.detach().clone().requires_grad_(True); b2 = b_.detach().clone().requires_grad_(True)
    p1 = ((a2 - torch.gather(b2, 1, i1.long()[..., None].expand(-1, -1, 3))) ** 2).sum(-1)
    p2 = ((b2 - torch.gather(a2, 1, i2.long()[..., None].expand(-1, -1, 3))) ** 2).sum(-1)
    (p1.mean(1) + p2.mean(1)).sum().backward()
    torch.testing.assert_close(a.grad, a2.grad, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(b_.grad, b2.grad, rtol=1e-5, atol=1e-7)


def _paired_case(kind, b, n, seed):
    rng = np.random.default_rng(seed)
    x = (rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5))
    if kind == "attack":              # most points barely move, a few fly far (what the attack's pert looks like)
        pert = (1e-3 * rng.standard_normal((b, n, 3))).astype(np.float32)
        far = rng.random((b, n)) < 0.02
        pert[far] += (rng.standard_normal((int(far.sum()), 3)) * 0.8).astype(np.float32)
        adv = x + pert
    elif kind == "zero":              # adv == x exactly: every distance 0, ties between duplicated points
        x[:, n // 2:] = x[:, :n - n // 2]
        adv = x.copy()
    elif kind == "unpaired":          # no relation at all: every query is a "far" query
        adv = (rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5))
    elif kind == "outside":           # clouds well outside the grid's box, duplicated targets
        x = x * np.float32(3.0) + np.float32(0.7)
        adv = x + (0.05 * rng.standard_normal((b, n, 3))).astype(np.float32)
        if n > 1:
            x[:, 1] = x[:, 0]
    else:                             # medium moves: balls spanning several cells
        adv = x + (0.05 * rng.standard_normal((b, n, 3))).astype(np.float32)
    return adv.astype(np.float32), x


@pytest.mark.parametrize("kind", ["attack", "zero", "unpaired", "outside", "medium"])
@pytest.mark.parametrize("b,n", [(3, 2048), (2, 1000), (2, 64), (1, 1), (1, 4096), (2, 8192), (1, 5000)])
def test_paired_grid_search_equals_default(kind, b, n):
    """Exact grid search seeded with the pairing (the attack's nn_distance(adv, x)): dist and idx bit for bit, whatever
    the data -- good pairing, none at all, exact ties, points outside the grid."""
    import torch
    from geometric_adv_amd import ops
    adv, x = _paired_case(kind, b, n, 100 + n)
    adv, x = torch.as_tensor(adv).cuda(), torch.as_tensor(x).cuda()
    ref = ops.nn_distance(adv, x, kernel="scan")
    got = ops.nn_distance_paired(adv, x)
    for a, c in zip(ref, got):
        assert torch.equal(a, c)


@pytest.mark.parametrize("kernel", ["scan", "symmetric", "paired"])
def test_nn_distance_nonfinite_golden(dev, golden_nn_nonfinite, kernel):
    """NaN / +-inf coordinates: the reference's own outputs (oracle/make_golden_nonfinite.py; tf_nndistance.cpp:31-40 --
    candidate 0 always taken, a NaN never wins later) from all three operator kernels: indices exact, distances equal with
    NaN == NaN.  (The paired search needs n == m.)"""
    from geometric_adv_amd import ops
    g = golden_nn_nonfinite
    ran = 0
    for name in g["cases"]:
        x1, x2 = g[f"{name}_xyz1"], g[f"{name}_xyz2"]
        if kernel == "paired":
            if x1.shape[1] != x2.shape[1]:
                continue
            got = ops.nn_distance_paired(_t(x1, dev), _t(x2, dev))
        else:
            got = ops.nn_distance(_t(x1, dev), _t(x2, dev), kernel=kernel)
        d1, i1, d2, i2 = [t.cpu().numpy() for t in got]
        assert np.array_equal(i1, g[f"{name}_idx1"]), name
        assert np.array_equal(i2, g[f"{name}_idx2"]), name
        assert np.array_equal(d1, g[f"{name}_dist1"], equal_nan=True), name
        assert np.array_equal(d2, g[f"{name}_dist2"], equal_nan=True), name
        ran += 1
    assert ran >= 5
