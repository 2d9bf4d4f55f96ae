"""Generate tests/golden/nn_distance_nonfinite.npz from the REFERENCE's own nnsearch -- TEST INFRASTRUCTURE.

Runs only in the build container (needs oracle/_ref/libgeoadv_ref.so = tf_nndistance.cpp:21-43 compiled by
oracle/build_ref.sh).  Inputs with NaN / +-inf coordinates: the reference loop always takes candidate 0
(`k==0 || d<best`, :33), so a NaN distance to candidate 0 stays (NaN, 0) and a NaN distance to any later
candidate never wins; an infinite distance loses to every finite one and to an earlier infinite one.
The file holds inputs and the reference's outputs only (data, never source); deterministic (fixed seeds).

    python oracle/make_golden_nonfinite.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.cpu_oracle import Reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def cloud(rng, b, n):
    return (rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5)).astype(np.float32)


def poison(rng, x1, x2, kind):
    nan, inf = np.float32(np.nan), np.float32(np.inf)
    b, n, m = x1.shape[0], x1.shape[1], x2.shape[1]
    if kind == "nan_query":            # a few query points with a NaN coordinate, none at index 0
        for c in range(b):
            x1[c, rng.integers(1, n, 3), rng.integers(0, 3, 3)] = nan
            x2[c, rng.integers(1, m, 3), rng.integers(0, 3, 3)] = nan
    elif kind == "nan_first":          # candidate 0 of both clouds is NaN: every distance to it is NaN and stays
        x1[:, 0, 1] = nan
        x2[:, 0, 2] = nan
    elif kind == "inf_coord":          # +-inf coordinates: infinite distances, inf - inf = NaN where both clouds have them
        for c in range(b):
            x1[c, rng.integers(0, n, 2), 0] = inf
            x2[c, rng.integers(0, m, 2), 0] = inf
            x1[c, rng.integers(0, n, 1), 1] = -inf
            x2[c, rng.integers(1, m, 1), 2] = -inf
    elif kind == "all_inf_row":        # a query at infinity: every distance inf (or NaN), index 0 unless candidate 0 is NaN
        x1[:, n // 2, :] = inf
        x2[:, m // 3, 0] = inf
    elif kind == "mixed":
        x1[:, 0, 0] = inf
        x2[:, 0, 0] = inf              # d(0, 0) = NaN: row 0 and column 0 stay NaN
        x1[:, n - 1, 2] = nan
        x2[:, m - 1, 1] = -inf
    else:
        raise ValueError(kind)


def main():
    ref = Reference()
    out, names = {}, []
    shapes = [("s", 2, 300, 200, 51), ("n2048", 1, 2048, 2048, 52), ("wide", 1, 1500, 4100, 53)]
    for kind in ("nan_query", "nan_first", "inf_coord", "all_inf_row", "mixed"):
        for tag, b, n, m, seed in shapes:
            rng = np.random.default_rng(seed + len(names))
            x1, x2 = cloud(rng, b, n), cloud(rng, b, m)
            poison(rng, x1, x2, kind)
            with np.errstate(all="ignore"):
                d1, i1, d2, i2 = ref.nn_distance(x1, x2)
            name = "%s_%s" % (kind, tag)
            names.append(name)
            out.update({name + "_xyz1": x1, name + "_xyz2": x2, name + "_dist1": d1, name + "_idx1": i1,
                        name + "_dist2": d2, name + "_idx2": i2})
    out["cases"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "nn_distance_nonfinite.npz"), **out)
    print("wrote", len(names), "cases")


if __name__ == "__main__":
    main()
