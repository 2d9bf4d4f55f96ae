"""nn_distance as an operator: the public op's two-scan kernel against the symmetric scan (ops.nn_distance_sym), ms per call."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geometric_adv_amd import ops
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for b, n, m in [(1, 2048, 2048), (4, 2048, 2048), (32, 2048, 2048), (256, 2048, 2048), (32, 8192, 8192), (32, 1024, 1024), (64, 512, 512), (70, 33, 65), (2, 300, 1000), (8, 4096, 1024)]:
    x = torch.rand((b, n, 3), device="cuda") - 0.5
    y = torch.rand((b, m, 3), device="cuda") - 0.5
    print(json.dumps({"b": b, "n": n, "m": m, "scan_ms": round(t(lambda: ops.nn_distance(x, y)), 4), "sym_ms": round(t(lambda: ops.nn_distance_sym(x, y)), 4)}))
