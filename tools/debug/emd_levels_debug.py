"""Debug: per-level factors of the GPU approx_match against an fp64 numpy run of the same factorised algorithm (golden case a)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geometric_adv_amd import _lib, ops
g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "approxmatch.npz"))
name = sys.argv[1] if len(sys.argv) > 1 else "a"
x1, x2, want = g[name + "_xyz1"], g[name + "_xyz2"], g[name + "_match_nm"]
b, n, _ = x1.shape; m = x2.shape[1]
d1, d2_ = torch.as_tensor(x1).cuda(), torch.as_tensor(x2).cuda()
nf = _lib.lib().geoadv_approx_match_temp_floats(b, n, m)
temp = torch.zeros(int(nf) + 4, dtype=torch.float32, device="cuda")
match = torch.empty((b, m, n), dtype=torch.float32, device="cuda")
_lib.check(_lib.lib().geoadv_approx_match(b, n, m, _lib.ptr(d1), _lib.ptr(d2_), _lib.ptr(match), _lib.ptr(temp), _lib.stream_handle()), "am")
torch.cuda.synchronize()
off = (-temp.data_ptr()) % 8 // 4
T = temp[off:off + 2 * b * (n + m) * 12].cpu().numpy().view(np.float64).reshape(b, 12, n + m)
got = match.cpu().numpy().transpose(0, 2, 1)
err = np.abs(got - want); viol = err > 2e-6 + 2e-5 * np.abs(want)
print("violations", viol.sum(), np.argwhere(viol)[:20].tolist())
for c in range(b):
    X1, X2 = x1[c].astype(np.float64), x2[c].astype(np.float64)
    D2 = ((X1[:, None] - X2[None]) ** 2).sum(-1)
    remL = np.full(n, float(max(n, m) // n)); remR = np.full(m, float(max(n, m) // m))
    for li, j in enumerate(range(8, -3, -1)):
        level = 0.0 if j == -2 else -float(np.float32(4.0) ** np.float32(j))
        w = np.exp((level * D2).astype(np.float32)).astype(np.float64)
        fL = remL / (1e-9 + (w * remR[None]).sum(1))
        Tl = (w * fL[:, None]).sum(0)
        fR = remR * np.minimum(remR / (1e-9 + remR * Tl), 1.0)
        remR = np.maximum(remR - fR * Tl, 0)
        remL = np.maximum(remL - fL * (w * fR[None]).sum(1), 0)
        gL, gR = T[c, 1 + li, :n], T[c, 1 + li, n:]
        eL = np.abs(gL - fL) / np.maximum(np.abs(fL), 1e-300); eR = np.abs(gR - fR) / np.maximum(np.abs(fR), 1e-300)
        print("cloud %d level j=%2d  fL relerr max %.2e (at k=%d, fL=%.3e gpu %.3e)  fR relerr max %.2e (at l=%d, fR=%.3e gpu %.3e)" %
              (c, j, eL.max(), eL.argmax(), fL[eL.argmax()], gL[eL.argmax()], eR.max(), eR.argmax(), fR[eR.argmax()], gR[eR.argmax()]))
