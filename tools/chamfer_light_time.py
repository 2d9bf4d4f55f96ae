import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometric_adv_amd import ops
rng = np.random.default_rng(0)
x = torch.as_tensor(rng.random((32, 2048, 3), dtype=np.float32) - 0.5).cuda()
y = torch.as_tensor(rng.random((32, 2048, 3), dtype=np.float32) - 0.5).cuda()
for name, f in (("default", ops.nn_distance), ("light", ops.nn_distance_light)):
    for _ in range(3): f(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f(x, y)
    e1.record(); torch.cuda.synchronize()
    print(name, "%.1f us per call (B=32, N=2048, both directions)" % (e0.elapsed_time(e1) / 20 * 1e3))
