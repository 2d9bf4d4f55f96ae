"""Generate tests/golden/*.npz from the REFERENCE's own CPU functions -- TEST INFRASTRUCTURE.

Runs only in the build container (needs /root/reference and oracle/_ref/libgeoadv_ref.so built
by oracle/build_ref.sh).  The committed .npz files hold inputs and the reference's outputs (data,
never source).  Re-running is deterministic (fixed seeds).

    python oracle/make_golden.py

Vector set:
  G1 nn_distance forward   : nnsearch (tf_nndistance.cpp:21-43), several shapes + tie cases
  G2 nn_distance gradient  : CPU grad loops (tf_nndistance.cpp:126-163)
  G3 approx-EMD            : approxmatch_cpu / matchcost_cpu / matchcostgrad_cpu (tf_approxmatch.cpp:23-140)
  G4 grouping              : selection_sort_cpu (test/selection_sort.cpp:20-63) incl. the in-file
                             known-answer case (:68-78), query_ball_point_cpu / group_point_cpu /
                             group_point_grad_cpu (test/query_ball_point.cpp:19-84)
  G5 chamfer_python        : transfer/atlasnet/auxiliary/ChamferDistancePytorch/chamfer_python.py:18-39
                             (float64 GEMM form; values only, its argmin may differ on near-ties)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.cpu_oracle import Reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def cloud(rng, b, n):
    """Synthetic cloud, uniform in [-0.5, 0.5)^3."""
    return (rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5)).astype(np.float32)


def main():
    ref = Reference()
    os.makedirs(OUT, exist_ok=True)

    # ---- G1 / G2 / G5 ------------------------------------------------------
    g1 = {}
    cases = [("small", 2, 64, 48, 11), ("mid", 2, 1024, 1024, 12), ("n2048", 1, 2048, 2048, 13),
             ("ragged", 3, 37, 129, 14), ("single", 1, 1, 1, 15), ("one_vs_many", 2, 1, 77, 16)]
    for name, b, n, m, seed in cases:
        rng = np.random.default_rng(seed)
        x1, x2 = cloud(rng, b, n), cloud(rng, b, m)
        d1, i1, d2, i2 = ref.nn_distance(x1, x2)
        gd1 = rng.standard_normal((b, n)).astype(np.float32)
        gd2 = rng.standard_normal((b, m)).astype(np.float32)
        gx1, gx2 = ref.nn_distance_grad(x1, x2, gd1, i1, gd2, i2)
        g1.update({f"{name}_xyz1": x1, f"{name}_xyz2": x2, f"{name}_dist1": d1, f"{name}_idx1": i1,
                   f"{name}_dist2": d2, f"{name}_idx2": i2, f"{name}_gd1": gd1, f"{name}_gd2": gd2,
                   f"{name}_gxyz1": gx1, f"{name}_gxyz2": gx2})
    # ties: duplicated target points (the duplicate at the higher index must never win), a coarse
    # lattice (many exactly equal distances) and identical clouds (distance exactly 0)
    rng = np.random.default_rng(17)
    x1 = cloud(rng, 2, 96)
    x2 = cloud(rng, 2, 64)
    x2[:, 40:] = x2[:, :24]
    lat = (rng.integers(-2, 3, size=(2, 128, 3)).astype(np.float32) * np.float32(0.25))
    lat2 = (rng.integers(-2, 3, size=(2, 80, 3)).astype(np.float32) * np.float32(0.25))
    same = cloud(rng, 1, 130)
    for name, a, c in [("dup", x1, x2), ("lattice", lat, lat2), ("same", same, same.copy())]:
        d1, i1, d2, i2 = ref.nn_distance(a, c)
        b, n, m = a.shape[0], a.shape[1], c.shape[1]
        gd1 = rng.standard_normal((b, n)).astype(np.float32)
        gd2 = rng.standard_normal((b, m)).astype(np.float32)
        gx1, gx2 = ref.nn_distance_grad(a, c, gd1, i1, gd2, i2)
        g1.update({f"{name}_xyz1": a, f"{name}_xyz2": c, f"{name}_dist1": d1, f"{name}_idx1": i1,
                   f"{name}_dist2": d2, f"{name}_idx2": i2, f"{name}_gd1": gd1, f"{name}_gd2": gd2,
                   f"{name}_gxyz1": gx1, f"{name}_gxyz2": gx2})
    g1["cases"] = np.array([c[0] for c in cases] + ["dup", "lattice", "same"])

    # G5: the importable torch twin (values only)
    try:
        import torch
        sys.path.insert(0, "/root/reference/transfer/atlasnet/auxiliary/ChamferDistancePytorch")
        import chamfer_python  # noqa
        for name in ["small", "mid"]:
            a = torch.from_numpy(g1[f"{name}_xyz1"]).double()
            c = torch.from_numpy(g1[f"{name}_xyz2"]).double()
            out = chamfer_python.distChamfer(a, c)
            g1[f"{name}_pt_dist1"] = out[0].numpy()
            g1[f"{name}_pt_dist2"] = out[1].numpy()
    except Exception as e:  # pragma: no cover
        print("chamfer_python unavailable:", e)
    np.savez_compressed(os.path.join(OUT, "nn_distance.npz"), **g1)

    # ---- G3 ------------------------------------------------------------------
    g3 = {}
    cases = [("a", 2, 64, 48, 21), ("b", 1, 256, 256, 22), ("c", 1, 64, 256, 23), ("d", 2, 96, 32, 24)]
    for name, b, n, m, seed in cases:
        rng = np.random.default_rng(seed)
        x1, x2 = cloud(rng, b, n), cloud(rng, b, m)
        if name == "b":                     # the EMD(adv, x) regime: near-coincident pairs
            x2 = (x1 + rng.standard_normal(x1.shape).astype(np.float32) * np.float32(1e-3)).astype(np.float32)
        match = ref.approx_match(x1, x2)
        cost = ref.match_cost(x1, x2, match)
        gx1, gx2 = ref.match_cost_grad(x1, x2, match)
        g3.update({f"{name}_xyz1": x1, f"{name}_xyz2": x2, f"{name}_match_nm": match, f"{name}_cost": cost,
                   f"{name}_grad1": gx1, f"{name}_grad2": gx2})
    g3["cases"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(OUT, "approxmatch.npz"), **g3)

    # ---- G4 ------------------------------------------------------------------
    g4 = {}
    b, n, m, k = 2, 4, 2, 3                 # the file's own known-answer case
    dist = (10 - np.arange(b * m * n, dtype=np.float32)).reshape(b, m, n)
    idx, val = ref.selection_sort(k, dist)
    g4.update(kat_dist=dist, kat_k=np.int32(k), kat_idx=idx, kat_val=val)
    rng = np.random.default_rng(31)
    dist = rng.random((1, 16, 64), dtype=np.float32)
    idx, val = ref.selection_sort(9, dist)
    g4.update(rnd_dist=dist, rnd_k=np.int32(9), rnd_idx=idx, rnd_val=val)
    dist = rng.integers(0, 4, size=(2, 24, 40)).astype(np.float32)       # many exact ties
    idx, val = ref.selection_sort(9, dist)
    g4.update(tie_dist=dist, tie_k=np.int32(9), tie_idx=idx, tie_val=val)
    dist = np.array([[[1, 1, 0, 1, 7, 7], [5, 1, 5, 1, 1, 0]]], np.float32)
    idx, val = ref.selection_sort(4, dist)
    g4.update(swap_dist=dist, swap_k=np.int32(4), swap_idx=idx, swap_val=val)
    dist = rng.random((1, 3, 5), dtype=np.float32)                        # k == n
    idx, val = ref.selection_sort(5, dist)
    g4.update(full_dist=dist, full_k=np.int32(5), full_idx=idx, full_val=val)

    b, n, m, ns, c = 2, 64, 16, 8, 4
    x1 = rng.random((b, n, 3), dtype=np.float32)
    x2 = rng.random((b, m, 3), dtype=np.float32)
    pts = rng.random((b, n, c), dtype=np.float32)
    qidx = ref.query_ball_point(0.3, ns, x1, x2)
    grp = ref.group_point(pts, qidx)
    gout = rng.standard_normal((b, m, ns, c)).astype(np.float32)
    gpts = ref.group_point_grad(pts, qidx, gout)
    g4.update(qb_xyz1=x1, qb_xyz2=x2, qb_radius=np.float32(0.3), qb_nsample=np.int32(ns), qb_idx=qidx,
              gp_points=pts, gp_out=grp, gp_grad_out=gout, gp_grad_points=gpts)
    np.savez_compressed(os.path.join(OUT, "grouping.npz"), **g4)

    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
