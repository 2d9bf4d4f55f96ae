import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from geometric_adv_amd import ops
from test_gpu_chamfer import _paired_case
for kind in ("attack", "zero", "medium", "unpaired"):
    adv, x = _paired_case(kind, 32, 2048, 5)
    adv, x = torch.as_tensor(adv).cuda(), torch.as_tensor(x).cuda()
    for name, f in (("default", ops.nn_distance), ("paired", ops.nn_distance_paired)):
        for _ in range(3): f(adv, x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f(adv, x)
        e1.record(); torch.cuda.synchronize()
        print("%-9s %-8s %.1f us per call (B=32, N=2048, both directions)" % (kind, name, e0.elapsed_time(e1) / 20 * 1e3))
