"""CPU: pin the oracle (oracle/geoadv_oracle.c) to the golden vectors produced by the reference's
own CPU functions (oracle/make_golden.py).  Bit-exact for every integer output and for every
float output whose arithmetic order the oracle restates (all of them except nothing)."""
import numpy as np
import pytest


def test_nn_distance_matches_reference_bit_exact(oracle, golden_nn):
    g = golden_nn
    for name in g["cases"]:
        d1, i1, d2, i2 = oracle.nn_distance(g[f"{name}_xyz1"], g[f"{name}_xyz2"])
        assert np.array_equal(i1, g[f"{name}_idx1"]), name
        assert np.array_equal(i2, g[f"{name}_idx2"]), name
        assert np.array_equal(d1.view(np.uint32), g[f"{name}_dist1"].view(np.uint32)), name
        assert np.array_equal(d2.view(np.uint32), g[f"{name}_dist2"].view(np.uint32)), name


def test_nn_distance_nonfinite_inputs_match_reference(oracle, golden_nn_nonfinite):
    """NaN / +-inf coordinates (oracle/make_golden_nonfinite.py; tf_nndistance.cpp:31-40: candidate 0 is always taken, a NaN
    distance never wins later): indices exact, distances equal with NaN == NaN (the payload of a NaN is not pinned)."""
    g = golden_nn_nonfinite
    for name in g["cases"]:
        with np.errstate(all="ignore"):
            d1, i1, d2, i2 = oracle.nn_distance(g[f"{name}_xyz1"], g[f"{name}_xyz2"])
        assert np.array_equal(i1, g[f"{name}_idx1"]), name
        assert np.array_equal(i2, g[f"{name}_idx2"]), name
        assert np.array_equal(d1, g[f"{name}_dist1"], equal_nan=True), name
        assert np.array_equal(d2, g[f"{name}_dist2"], equal_nan=True), name
    # the cases are not vacuous: NaN results at index 0, infinite results, and finite results beside them
    d = np.concatenate([g[f"{n}_dist1"].ravel() for n in g["cases"]])
    assert np.isnan(d).any() and np.isinf(d).any() and np.isfinite(d).any()
    assert not g["nan_first_s_idx1"].any() and np.isnan(g["nan_first_s_dist1"]).all()


def test_nn_distance_ties_pick_lowest_index(golden_nn):
    g = golden_nn
    # duplicated targets sit at indices 40.. (copies of 0..23): never selected
    assert g["dup_idx1"].max() < 40
    # identical clouds: every point matches itself at distance exactly 0
    assert np.array_equal(g["same_idx1"][0], np.arange(130))
    assert not g["same_dist1"].any()


def test_nn_distance_grad_matches_reference_bit_exact(oracle, golden_nn):
    g = golden_nn
    for name in g["cases"]:
        gx1, gx2 = oracle.nn_distance_grad(g[f"{name}_xyz1"], g[f"{name}_xyz2"], g[f"{name}_gd1"],
                                           g[f"{name}_idx1"], g[f"{name}_gd2"], g[f"{name}_idx2"])
        assert np.array_equal(gx1.view(np.uint32), g[f"{name}_gxyz1"].view(np.uint32)), name
        assert np.array_equal(gx2.view(np.uint32), g[f"{name}_gxyz2"].view(np.uint32)), name


def test_nn_distance_values_vs_torch_twin(golden_nn):
    """chamfer_python.distChamfer (float64 GEMM form) agrees in value to ~1e-7 abs."""
    g = golden_nn
    for name in ["small", "mid"]:
        if f"{name}_pt_dist1" not in g:
            pytest.skip("torch twin vectors not generated")
        np.testing.assert_allclose(g[f"{name}_dist1"], g[f"{name}_pt_dist1"], atol=2e-7, rtol=0)
        np.testing.assert_allclose(g[f"{name}_dist2"], g[f"{name}_pt_dist2"], atol=2e-7, rtol=0)


def test_approxmatch_matches_reference(oracle, golden_emd):
    g = golden_emd
    for name in g["cases"]:
        x1, x2 = g[f"{name}_xyz1"], g[f"{name}_xyz2"]
        match = oracle.approx_match(x1, x2)
        assert np.array_equal(match.view(np.uint32), g[f"{name}_match_nm"].view(np.uint32)), name
        cost = oracle.match_cost(x1, x2, match)
        assert np.array_equal(cost, g[f"{name}_cost"]), name
        g1, g2 = oracle.match_cost_grad(x1, x2, match)
        assert np.array_equal(g1, g[f"{name}_grad1"]), name
        assert np.array_equal(g2, g[f"{name}_grad2"]), name


def test_approxmatch_is_a_transport_plan(golden_emd):
    """Property the domain offers: row sums <= capacity max(n,m)/n, column sums <= max(n,m)/m,
    total mass == min(n*fl, m*fr) up to the 1e-9 regularisers."""
    g = golden_emd
    for name in g["cases"]:
        match = g[f"{name}_match_nm"].astype(np.float64)
        b, n, m = match.shape
        fl, fr = max(n, m) // n, max(n, m) // m
        assert (match.sum(2) <= fl + 1e-4).all()
        assert (match.sum(1) <= fr + 1e-4).all()
        np.testing.assert_allclose(match.sum((1, 2)), min(n * fl, m * fr), rtol=2e-2)


def test_selection_sort_known_answer_and_ties(oracle, golden_grouping):
    g = golden_grouping
    # the reference file's own case: rows of 10-i => ascending order is the reversed row
    assert np.array_equal(g["kat_idx"][0, 0, :3], [3, 2, 1])
    for name in ["kat", "rnd", "tie", "full", "swap"]:
        k = int(g[f"{name}_k"])
        idx, val = oracle.selection_sort(k, g[f"{name}_dist"])
        assert np.array_equal(idx, g[f"{name}_idx"]), name
        assert np.array_equal(val, g[f"{name}_val"]), name
    # the swap tie rule: {1,1,0,1,7,7} -> [2,1,0,3], not the stable [2,0,1,3]
    assert list(g["swap_idx"][0, 0, :4]) == [2, 1, 0, 3]
    assert list(g["swap_idx"][0, 1, :4]) == [5, 1, 3, 4]


def test_query_ball_and_group_point(oracle, golden_grouping):
    g = golden_grouping
    idx, cnt = oracle.query_ball_point(float(g["qb_radius"]), int(g["qb_nsample"]), g["qb_xyz1"], g["qb_xyz2"])
    assert np.array_equal(idx, g["qb_idx"])
    assert cnt.min() >= 0 and cnt.max() <= int(g["qb_nsample"])
    out = oracle.group_point(g["gp_points"], g["qb_idx"])
    assert np.array_equal(out, g["gp_out"])
    gp = oracle.group_point_grad(g["gp_points"], g["qb_idx"], g["gp_grad_out"])
    assert np.array_equal(gp, g["gp_grad_points"])


def test_knn_dists_vs_numpy_fallback(oracle):
    """The reference's own numpy fallback (defender/get_knn_dists_per_point.py:124-137):
    sort of the euclidean distance matrix, self dropped.  Values agree to float rounding."""
    from conftest import cloud
    pc = cloud(5, 2, 200)
    got = oracle.knn_dists(pc, 8)
    d = np.linalg.norm(pc[:, :, None, :].astype(np.float64) - pc[:, None, :, :], axis=-1)
    want = np.sort(d, axis=2)[:, :, 1:9]
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-7)
