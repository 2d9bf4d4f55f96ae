"""group_point_grad at a PointNet++-sized shape (b = 32 clouds, n = 2048 points, c = 64 channels, m = 2048 x nsample = 32)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geometric_adv_amd import ops
b, n, c, m, ns = 32, 2048, 64, 2048, 32
g = torch.Generator(device="cuda").manual_seed(0)
points = torch.randn((b, n, c), device="cuda", generator=g)
idx = torch.randint(0, n, (b, m, ns), device="cuda", generator=g, dtype=torch.int32)
grad_out = torch.randn((b, m, ns, c), device="cuda", generator=g)
for _ in range(3): out = ops.group_point_grad(points, idx, grad_out)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): out = ops.group_point_grad(points, idx, grad_out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print(json.dumps({"shape": [b, n, c, m, ns], "ms": dt * 1e3, "grad_out_GB": grad_out.numel() * 4 / 1e9, "GB_per_s": grad_out.numel() * 4 / dt / 1e9}))
