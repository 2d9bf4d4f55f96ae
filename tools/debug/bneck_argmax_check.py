"""Where the critical points (arg-max rows of the pool) of an arithmetic differ from the float64 model's: the two candidates' values."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from conftest import cloud
from geometric_adv_amd import weights as W
from geometric_adv_amd.autoencoder import PointNetAE
from oracle.attack_model import AEModel
n, b, bneck = 512, 5, 64
w = W.randomized_weights(n, seed=11, bneck=bneck)
model = AEModel(W.canonical(w, n), n, np.float64)
x = cloud(5, b, n)
p0 = (1e-3 * np.random.default_rng(1).standard_normal((b, n, 3))).astype(np.float32)
adv = (x + p0).astype(np.float32)
z64, hs = model.encode(adv.astype(np.float64), keep=True)
h5 = hs[-1]
arg64 = h5.argmax(axis=1)
for arith in ("f16x2", "bf16x3", "f32"):
    ae = PointNetAE(w, n, encoder_arith=arith)
    mv, mi = ae.max_and_argmax(adv)
    mi = mi.cpu().numpy()
    diff = np.argwhere((mi != arg64) & (z64 > 0))
    print(arith, "channels whose critical point differs from the float64 model's:", len(diff))
    for (bb, c) in diff:
        print("   cloud %d channel %d: gpu row %d (f64 value %.9g) model row %d (%.9g) rel gap %.2e" % (bb, c, mi[bb, c], h5[bb, mi[bb, c], c], arg64[bb, c], h5[bb, arg64[bb, c], c],
              (h5[bb, arg64[bb, c], c] - h5[bb, mi[bb, c], c]) / h5[bb, arg64[bb, c], c]))
