"""What the approx-EMD binning kernel decided for a batch of cloud pairs: per level the grid, `use`, and the heavy-point counts.
Reads the sparse scratch behind the doubles of an emd_cost_grad1 call (layout: csrc/emd.hip, sp_view)."""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from geometric_adv_amd import _lib
kind = sys.argv[1] if len(sys.argv) > 1 else "blob"
b, n, m = 4, 2048, 2048
rng = np.random.default_rng(1)
x2 = rng.random((b, m, 3), dtype=np.float32) - np.float32(0.5)
x1 = (rng.standard_normal((b, n, 3)) * 0.022).astype(np.float32) if kind == "blob" else rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5)
X1, X2 = torch.as_tensor(x1).cuda(), torch.as_tensor(x2).cuda()
lib = _lib.lib()
nf = lib.geoadv_emd_cost_grad1_temp_floats(b, n, m)
temp = torch.zeros(int(nf), dtype=torch.float32, device="cuda")
cost = torch.empty(b, device="cuda"); g1 = torch.empty_like(X1)
_lib.check(lib.geoadv_emd_cost_grad1(b, n, m, _lib.ptr(X1), _lib.ptr(X2), _lib.ptr(cost), _lib.ptr(g1), _lib.ptr(temp), _lib.stream_handle()), "emd")
torch.cuda.synchronize()
raw = temp.cpu().numpy().view(np.uint8)
up = lambda v: (v + 15) & ~15
LEVELS, CELLS = 3, 4096
doubles = (n + m) * 12 * b + b * ((n + 63) // 64)            # levels' doubles + cost partials
base_ptr = temp.data_ptr()
t_al = (base_ptr + 7) & ~7
sp = ((t_al + doubles * 8 + 15) & ~15) - base_ptr
grid_b = up(40 * LEVELS)
cloud_b = lambda nx: up(16 * nx) + up(4 * (CELLS + 4)) + up(4 * nx) + 2 * up(8 * nx) + up(4 * (nx + 4))
pair_b = grid_b + LEVELS * (cloud_b(n) + cloud_b(m))
for c in range(b):
    base = sp + c * pair_b
    out = []
    for lv in range(LEVELS):
        g = raw[base + 40 * lv: base + 40 * lv + 40]
        gi = g.view(np.int32); gf = g.view(np.float32)
        p = base + grid_b + lv * (cloud_b(n) + cloud_b(m))
        h = []
        for nx in (n, m):
            q = p + up(16 * nx) + up(4 * (CELLS + 4)) + up(4 * nx) + 2 * up(8 * nx)
            h.append(int(raw[q:q + 4].view(np.int32)[0]))
            p += cloud_b(nx)
        out.append({"g": gi[6:9].tolist(), "use": int(gi[9]), "heavy_cloud1": h[0], "heavy_cloud2": h[1]})
    print(json.dumps({"pair": c, "levels": out}))
