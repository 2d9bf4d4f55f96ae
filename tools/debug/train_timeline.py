"""In-kernel stamps of one AE training step (diagnostic build, see iter_timeline.py): per stamped kernel, when its first / last
workgroup started and ended and the median time between phase stamps; plus how many workgroups were alive over time.
    python tools/debug/train_timeline.py"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from geometric_adv_amd import _lib
from geometric_adv_amd.trainer import PointNetAETrainer, initial_weights
B, N = 50, 2048
tr = PointNetAETrainer(initial_weights(N, seed=1), N, batch_size=B)
x = torch.as_tensor(np.random.default_rng(0).random((B, N, 3), dtype=np.float32) - np.float32(0.5)).cuda()
for _ in range(5):
    tr.partial_fit(x, want_recon=False, sync=False)
torch.cuda.synchronize()
NB = 1024
names = {0: "train_fwd<128,256>", 1: "train_fwd<256,128>", 2: "train_fwd<128,128>", 3: "train_fwd<64,128>",
         6: "train_bwd_fused<128,256> W waves, second tile: 0 start 1 next tile staged + the one after requested 2 products done 3 (barrier B, one-buffer staging) 7 barrier A",
         7: "train_bwd_fused<128,256> X waves, second tile: 0 start 1 chain done 2 epilogue done 7 barrier A"}
buf = (C.c_ulonglong * (8 * NB * 8))()
fn = getattr(_lib.lib(), "geoadv_debug_stamps_train", None)
if fn is None:
    raise SystemExit("this libgeoadv.so was not built with -DGA_STAMPS")
assert fn(buf) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(8, NB, 8).astype(np.int64)
for k, nm in names.items():
    t = s[k]
    t = t[(t[:, 0] > 0) & (t[:, 7] > 0)]
    if not len(t):
        continue
    u = (t - t[:, 0].min()) / 100.0
    ph, prev = {}, 0
    for i in range(1, 8):
        if (t[:, i] > 0).all():
            ph["%d->%d" % (prev, i)] = round(float(np.median(u[:, i] - u[:, prev])), 2)
            prev = i
    # workgroups alive at a few instants
    grid = np.linspace(0, u[:, 7].max(), 9)[1:-1]
    alive = [int(((u[:, 0] <= g) & (u[:, 7] > g)).sum()) for g in grid]
    print(json.dumps({"kernel": nm, "stamped_workgroups": int(len(t)), "last_start": round(float(u[:, 0].max()), 2),
                      "first_end": round(float(u[:, 7].min()), 2), "last_end": round(float(u[:, 7].max()), 2),
                      "workgroup_us_median": round(float(np.median(u[:, 7] - u[:, 0])), 2), "phases_us_median": ph, "alive_at_eighths": alive}))
