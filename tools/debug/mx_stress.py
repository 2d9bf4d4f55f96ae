import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from geometric_adv_amd import ops
from test_gpu_chamfer_shapes import make_clouds
bad = 0
for (b, n, m, kind) in [(24, 2048, 16384, "sphere"), (16, 8192, 8192, "uniform"), (36, 2048, 2048, "sphere"), (200, 2048, 2048, "uniform")]:
    a, c = make_clouds(kind, 61, b, n), make_clouds(kind, 62, b, m)
    ta, tc = torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda()
    want = ops.nn_distance(ta, tc, kernel="scan")
    for rep in range(25):
        got = ops.nn_distance(ta, tc, kernel="symmetric")
        mism = [int((g != w).sum().item()) for g, w in zip(got, want)]
        if any(mism):
            bad += 1
            idx = (got[1] != want[1]).nonzero()[:5].tolist(), (got[3] != want[3]).nonzero()[:5].tolist()
            print("MISMATCH", b, n, m, kind, rep, mism, idx)
print("bad", bad)
