"""Worst deviation of approx_match from the pinned CPU oracle in both weight modes (numbers quoted in DESIGN.md / emd.hip).
Needs oracle/ built (python -c 'import __graft_entry__ as g; g.build()')."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import cloud
from geometric_adv_amd import ops
from oracle.cpu_oracle import Oracle
o = Oracle()
rng = np.random.default_rng(101)
cases = {"512x384": (cloud(1, 2, 512), cloud(2, 2, 384)), "1100x300": (cloud(7, 1, 1100), cloud(8, 1, 300)),
         "4096x1024": (rng.random((2, 4096, 3)).astype(np.float32), rng.random((2, 1024, 3)).astype(np.float32)),
         "2048x2048": (cloud(3, 1, 2048), cloud(4, 1, 2048))}
for name, (x1, x2) in cases.items():
    want = o.approx_match(x1, x2)
    row = {"case": name, "entries": int(want.size)}
    for mode in (False, True):
        got = ops.approx_match(torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda(), mode).cpu().numpy().transpose(0, 2, 1)
        err = np.abs(got - want)
        big = want > 1e-6
        row["reference_weights" if mode else "fast"] = {"max_abs": float(err.max()), "max_rel_where_gt_1e-6": float((err[big] / want[big]).max()),
                                                        "entries_outside_2e-5_2e-6": int((err > 2e-6 + 2e-5 * np.abs(want)).sum()),
                                                        "bit_equal_fraction": float((got == want).mean())}
    print(json.dumps(row), flush=True)
