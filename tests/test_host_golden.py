"""CPU: the host-side numpy bookkeeping (attack_data.py; oracle/host_defense.py, the checker of csrc/defense.hip) against golden vectors produced by the
reference's own function bodies (oracle/make_golden_host.py)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "host_logic.npz"))


def test_prepare_data_for_attack(g):
    from geometric_adv_amd.attack_data import prepare_data_for_attack
    classes, sl, idx, nn = g["prep_classes"], g["prep_slice_idx"], g["prep_attack_idx"], g["prep_nn_idx"]
    for name, key in [("pc", "prep_pcs"), ("lat", "prep_lat"), ("loss", "prep_loss")]:
        for cp_name, cp in [("all", None), ("correct", g["prep_correct"])]:
            src, tgt = prepare_data_for_attack(classes, ["table"], list(classes), g[key], sl, idx, 2, nn, cp)
            assert np.array_equal(src, g[f"prep_{name}_{cp_name}_src"]), (name, cp_name)
            assert np.array_equal(tgt, g[f"prep_{name}_{cp_name}_tgt"]), (name, cp_name)
    src, tgt = prepare_data_for_attack(classes, list(classes), ["chair", "car"], g["prep_pcs"], sl, idx, 3, nn, None)
    assert np.array_equal(src, g["prep_multi_src"]) and np.array_equal(tgt, g["prep_multi_tgt"])


def test_get_quantity_at_index(g):
    from geometric_adv_amd.attack_data import get_quantity_at_index
    assert np.array_equal(get_quantity_at_index([g["gq_quantity"]], g["gq_index"]), g["gq_out"])


def test_outlier_inlier_packing(g):
    """oracle/host_defense.py (the checker of the device packing kernels) against the reference's own function body."""
    from oracle.host_defense import outlier_inlier
    o_pc, o_idx, o_num, i_pc = outlier_inlier(g["out_pc"], g["out_knn"], float(g["out_thresh"]))
    assert np.array_equal(o_pc, g["out_outlier_pc"]) and np.array_equal(o_idx, g["out_outlier_idx"])
    assert np.array_equal(o_num, g["out_outlier_num"]) and np.array_equal(i_pc, g["out_inlier_pc"])
    assert o_idx.dtype == np.int16 and o_num.dtype == np.int16


def test_critical_points_bookkeeping(g):
    """The critical-point restatement against the reference's function bodies.  The reference orders points that own equally
    many channels with numpy's default (unstable) argsort: everything that does not depend on that order is compared bit
    for bit, the order itself up to permutations inside a group of equal counts -- and, with the SAME sort as the fixture's
    (kind=None on the same numpy build), bit for bit as well."""
    from oracle.host_defense import critical_and_rest, same_critical_sets
    pre = g["crit_pre"]
    mv, mi = pre.max(1), pre.argmax(1)
    cp, ci, cn, crit_pc, non = critical_and_rest(g["crit_in_pc"], mv, mi)
    assert np.array_equal(cn, g["crit_num"]) and np.array_equal(non, g["crit_noncrit_pc"])
    assert same_critical_sets(ci, cn, g["crit_idx"], g["crit_num"], mv, mi)
    for i in range(len(cn)):
        assert np.array_equal(cp[i, :cn[i]], g["crit_in_pc"][i][ci[i, :cn[i]]]) and not cp[i, cn[i]:].any()
        assert np.array_equal(crit_pc[i, :cn[i]], cp[i, :cn[i]]) and (crit_pc[i, cn[i]:] == cp[i, cn[i] - 1]).all()
    bad = ci.copy()
    bad[0, [0, int(cn[0]) - 1]] = bad[0, [int(cn[0]) - 1, 0]]          # most-owning point moved to the end: not a tie permutation
    assert not same_critical_sets(bad, cn, g["crit_idx"], g["crit_num"], mv, mi)


def test_load_data_by_basename(tmp_path):
    from geometric_adv_amd.attack_data import load_data
    np.save(tmp_path / "point_clouds_test_set_13l.npy", np.arange(6).reshape(2, 3))
    np.save(tmp_path / "ae_loss_test_set_13l.npy", np.ones(2))
    files = sorted(os.listdir(tmp_path))
    a, b = load_data(str(tmp_path), files, ["point_clouds_test_set", "ae_loss_test_set"])
    assert a.shape == (2, 3) and b.shape == (2,)
    assert load_data(str(tmp_path), files, ["ae_loss"]).shape == (2,)
    with pytest.raises(IndexError):
        load_data(str(tmp_path), files, ["missing"])


@pytest.mark.parametrize("case", ["output", "latent", "tied"])
def test_attack_model_matches_torch_golden(case):
    """oracle/attack_model.py (numpy, hand-written backward) against the committed vectors of the torch second opinion."""
    from geometric_adv_amd import weights as W
    from oracle.attack_model import AEModel, AttackModel
    g = np.load(os.path.join(GOLDEN, "torch_second_opinion.npz"))
    n, x, gt = int(g[f"{case}_n"]), g[f"{case}_x"], g[f"{case}_gt"]
    w = W.randomized_weights(n, seed=int(g[f"{case}_wseed"]))
    m = AEModel(W.canonical(w, n), n, np.float64)
    am = AttackModel(m, x, gt, g[f"{case}_tz"], np.full(len(x), float(g[f"{case}_dw"])), str(g[f"{case}_adv_type"]), str(g[f"{case}_dist_type"]))
    am.init_pert(g[f"{case}_pert"])
    f = am.forward(idx_override=tuple(g[f"{case}_idx{k}"] for k in range(4)))
    np.testing.assert_allclose(f["z"], g[f"{case}_z"], atol=1e-12)
    np.testing.assert_allclose(f["recon"], g[f"{case}_recon"], atol=1e-12)
    np.testing.assert_allclose(f["loss_ae"], g[f"{case}_loss_ae"], rtol=1e-12)
    np.testing.assert_allclose(f["loss_adv"], g[f"{case}_loss_adv"], rtol=1e-10)
    want = g[f"{case}_grad"]
    np.testing.assert_allclose(am.gradient(f), want, atol=1e-10 * np.abs(want).max())


def test_sort_dist_mat(g):
    """scorer.sort_dist_mat against the reference's own function body (prepare_indices_for_attack.py:167-183)."""
    from geometric_adv_amd.scorer import sort_dist_mat
    got = sort_dist_mat(g["sdm_dist"], g["sdm_slice_idx"])
    assert got.dtype == np.int16 and np.array_equal(got, g["sdm_nn_idx"])
