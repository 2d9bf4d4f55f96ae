"""The matrix-pipe-screened symmetric scan (csrc/chamfer_mx.h) against the two-scan kernel, bit for bit, on many shapes and cloud
kinds, and its time against the unscreened scan (ops.chamfer_screen).   python tools/debug/mx_check.py [time]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from geometric_adv_amd import ops


def timing():
    """us per call (kernel + the operators' closing launch + torch's allocations), screened and unscreened ALTERNATING on the same
    tensors: minimum of five windows each."""
    out = {}
    for b, n, m in ((32, 2048, 2048), (64, 2048, 2048), (128, 2048, 2048), (8, 2048, 2048), (32, 8192, 8192), (16, 2048, 2048), (50, 2048, 2048)):
        x = torch.rand((b, n, 3), device="cuda") - 0.5
        y = torch.rand((b, m, 3), device="cuda") - 0.5
        reps = 40 if n <= 2048 else 10
        best = {True: 1e9, False: 1e9}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for w in range(6):
            for on in (True, False):
                ops.chamfer_screen(on)
                ops.nn_distance_sym(x, y)
                e0.record()
                for _ in range(reps):
                    ops.nn_distance_sym(x, y)
                e1.record(); torch.cuda.synchronize()
                if w:
                    best[on] = min(best[on], e0.elapsed_time(e1) / reps)
        out["%dx%dx%d" % (b, n, m)] = {"screened": round(best[True] * 1e3, 1), "unscreened": round(best[False] * 1e3, 1)}
    ops.chamfer_screen(True)
    print(json.dumps({"us_per_call_incl_torch_alloc": out}))


def check():
    from test_gpu_chamfer_shapes import make_clouds
    bad = 0
    cases = []
    for kind in ("uniform", "sphere", "clusters", "duplicates", "lattice"):
        for b, n, m in ((2, 2048, 2048), (3, 1500, 777), (1, 8192, 8192), (2, 1025, 4100), (5, 2049, 300), (33, 2048, 2048), (2, 4097, 31), (1, 3000, 1), (16, 8192, 8192), (24, 2048, 16384)):
            cases.append((kind, b, n, m, 0.0))
    cases += [("uniform", 4, 2048, 2048, 100.0), ("sphere", 2, 2048, 2048, -1e4), ("uniform", 2, 2048, 2048, 1e-6)]
    for kind, b, n, m, shift in cases:
        a, c = make_clouds(kind, 11, b, n), make_clouds(kind, 12, b, m)
        if shift == 1e-6:
            a, c = (a * np.float32(1e-6)).astype(np.float32), (c * np.float32(1e-6)).astype(np.float32)
        elif shift:
            a, c = (a + np.float32(shift)).astype(np.float32), (c + np.float32(shift)).astype(np.float32)
        if kind in ("duplicates", "lattice") and n == m:
            c = c.copy(); c[:, : n // 2] = a[:, : n // 2]
        ta, tc = torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda()
        got = ops.nn_distance(ta, tc, kernel="symmetric")
        want = ops.nn_distance(ta, tc, kernel="scan")
        ok = all(torch.equal(g, w) for g, w in zip(got, want))
        if not ok:
            bad += 1
            mism = [int((g != w).sum().item()) for g, w in zip(got, want)]
            print("MISMATCH", kind, b, n, m, shift, mism)
    # the attack's paired regime: adv = x + small perturbation, and adv == x
    for scale in (0.0, 1e-4, 0.05):
        x = make_clouds("sphere", 5, 4, 2048)
        p = (x + np.float32(scale) * np.random.default_rng(1).standard_normal(x.shape).astype(np.float32)).astype(np.float32)
        tx, tp = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
        got = ops.nn_distance(tp, tx, kernel="symmetric")
        want = ops.nn_distance(tp, tx, kernel="scan")
        if not all(torch.equal(g, w) for g, w in zip(got, want)):
            bad += 1
            print("MISMATCH paired", scale, [int((g != w).sum().item()) for g, w in zip(got, want)])
    print("cases", len(cases) + 3, "bad", bad)
    return bad


if __name__ == "__main__":
    bad = check() if "notest" not in sys.argv else 0
    timing()
    sys.exit(1 if bad else 0)
