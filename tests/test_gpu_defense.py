"""GPU: the defenses' packing kernels (csrc/defense.hip: geoadv_outlier_filter, geoadv_critical_split) through the C ABI against
the golden vectors of the reference's own function bodies (tests/golden/host_logic.npz) and the pinned numpy restatement
(oracle/host_defense.py) -- integers and copies: bit for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _np(ts):
    return [t.cpu().numpy() for t in ts]


def test_outlier_filter_golden():
    from geometric_adv_amd import ops
    g = np.load(os.path.join(GOLDEN, "host_logic.npz"))
    o_pc, o_idx, o_num, i_pc = _np(ops.outlier_filter(_t(g["out_pc"]), _t(g["out_knn"]), float(g["out_thresh"])))
    assert np.array_equal(o_pc, g["out_outlier_pc"]) and np.array_equal(o_idx, g["out_outlier_idx"])
    assert np.array_equal(o_num, g["out_outlier_num"]) and np.array_equal(i_pc, g["out_inlier_pc"])
    assert o_idx.dtype == np.int16 and o_num.dtype == np.int16


@pytest.mark.parametrize("b,n", [(3, 1), (2, 63), (4, 1024), (3, 1025), (2, 2048), (2, 5000)])
def test_outlier_filter_vs_oracle(b, n):
    """Point counts around the 64-lane wave and the 1024-point chunk; clouds with no / only outliers; NaN scores (in
    neither set, like np.where); thresholds that equal scores exactly (`>` against `<=`)."""
    from geometric_adv_amd import ops
    from oracle.host_defense import outlier_inlier
    rng = np.random.default_rng(n)
    pc = rng.standard_normal((b, n, 3)).astype(np.float32)
    score = (rng.random((b, n)) * 0.08).astype(np.float32)
    score[0] = 0.0
    if b > 1:
        score[1] = 1.0
    if b > 2:
        score[2, ::3] = np.float32(0.04)                # exactly the threshold: inliers
        score[2, 1::7] = np.nan
    got = _np(ops.outlier_filter(_t(pc), _t(score), 0.04))
    want = outlier_inlier(pc, score, np.float32(0.04))
    for a, w in zip(got, want):
        assert a.dtype == w.dtype and np.array_equal(a, w)
    only_in = ops.outlier_filter(_t(pc), _t(score), 0.04, want_outliers=False)
    assert only_in[0] is None and np.array_equal(only_in[3].cpu().numpy(), want[3])


@pytest.mark.parametrize("top_k", [1, 2, 3, 5, 7])
def test_outlier_filter_fused_score_is_numpys_mean(top_k):
    """The kernel's own score (left-to-right float32 sum / count) == np.mean(knn[:, :, :top_k], axis=2) bit for bit: packing
    from (b, n, 8) distances equals packing from numpy's score."""
    from geometric_adv_amd import ops
    from oracle.host_defense import outlier_inlier
    rng = np.random.default_rng(top_k)
    pc = rng.standard_normal((3, 700, 3)).astype(np.float32)
    knn = np.sort(rng.random((3, 700, 8)).astype(np.float32) * 0.1, axis=2)
    score = knn[:, :, :top_k].mean(axis=2)
    thresh = np.float32(np.median(score))
    got = _np(ops.outlier_filter(_t(pc), _t(knn), float(thresh), top_k=top_k))
    want = outlier_inlier(pc, score, thresh)
    for a, w in zip(got, want):
        assert np.array_equal(a, w)
    with pytest.raises(ValueError):
        ops.outlier_filter(_t(pc), _t(knn), 0.04, top_k=8)


def test_critical_split_golden():
    """Against the reference's function bodies: everything that does not depend on numpy's unstable tie order bit for bit,
    the order of the critical points up to permutations inside a group of equal channel counts."""
    from geometric_adv_amd import ops
    from oracle.host_defense import same_critical_sets
    g = np.load(os.path.join(GOLDEN, "host_logic.npz"))
    pre, pcs = g["crit_pre"], g["crit_in_pc"]
    mv, mi = pre.max(1), pre.argmax(1).astype(np.int32)
    cp, ci, cn, crit_pc, non = _np(ops.critical_split(_t(pcs), _t(mv), _t(mi)))
    assert np.array_equal(cn, g["crit_num"]) and np.array_equal(non, g["crit_noncrit_pc"])
    assert ci.dtype == np.int16 and cn.dtype == np.int16
    assert same_critical_sets(ci, cn, g["crit_idx"], g["crit_num"], mv, mi)
    for i in range(len(cn)):
        assert np.array_equal(cp[i, :cn[i]], pcs[i][ci[i, :cn[i]]]) and not cp[i, cn[i]:].any() and not ci[i, cn[i]:].any()
        assert np.array_equal(crit_pc[i, :cn[i]], cp[i, :cn[i]]) and (crit_pc[i, cn[i]:] == cp[i, cn[i] - 1]).all()


@pytest.mark.parametrize("b,n,c", [(4, 40, 12), (3, 2048, 128), (2, 3000, 128), (2, 100, 256), (2, 64, 1024)])
def test_critical_split_vs_oracle(b, n, c):
    """Random arg-max tables with many shared owners, dead channels, a cloud where ONE point owns every channel and one where
    no channel is alive; more channels than points; several 1024-point chunks."""
    from geometric_adv_amd import ops
    from oracle.host_defense import critical_and_rest
    rng = np.random.default_rng(n + c)
    pc = rng.standard_normal((b, n, 3)).astype(np.float32)
    mi = rng.integers(0, n, size=(b, c)).astype(np.int32)
    mi[:, : c // 3] = rng.integers(0, max(2, n // 50), size=(b, c // 3))     # crowded owners: counts up to a dozen
    mv = rng.random((b, c)).astype(np.float32)
    mv[:, ::5] = 0.0
    mi[0] = 7 % n
    if b > 1:
        mv[1] = 0.0
    got = _np(ops.critical_split(_t(pc), _t(mv), _t(mi)))
    want = critical_and_rest(pc, mv, mi)
    for a, w in zip(got, want):
        assert a.dtype == w.dtype and np.array_equal(a, w)
    assert got[2][0] == 1 and (b == 1 or got[2][1] == 0)


def test_defenses_stay_on_the_device():
    """defend_surface_device / defend_critical_device: GPU tensors in, GPU tensors out, equal to the numpy-facing wrappers."""
    import torch
    from geometric_adv_amd import defense, weights as W
    from geometric_adv_amd.autoencoder import PointNetAE
    from conftest import cloud
    n = 512
    ae = PointNetAE(W.randomized_weights(n, seed=3), n)
    adv, src = cloud(1, 5, n), cloud(2, 5, n)
    adv[:, :7] += 3.0 * (np.arange(7, dtype=np.float32)[None, :, None] + 1)      # each far from everything else
    for dev_fn, np_fn in ((defense.defend_surface_device, defense.defend_surface), (defense.defend_critical_device, defense.defend_critical)):
        d = dev_fn(ae, _t(adv), _t(src))
        h = np_fn(ae, adv, src)
        assert all(isinstance(v, torch.Tensor) and v.is_cuda for v in d.values())
        for k in d:
            assert np.array_equal(d[k].cpu().numpy(), h[k]), k
    assert (defense.defend_surface(ae, adv, src, knn_dist_thresh=0.5)["outlier_num"] == 7).all()
