"""f16x2 when a layer's activations are SMALL: layer 0's batch-norm gamma / beta multiplied by 2^-k (layer 1's weights by 2^k, so that
everything behind stays what it was): latent error against the float64 model, in units of its largest component, per arithmetic."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from conftest import cloud
from geometric_adv_amd import weights as W
from geometric_adv_amd.autoencoder import PointNetAE
from oracle.attack_model import AEModel
n, b = 2048, 8
pc = cloud(71, b, n)
for k in (-8, 0, 4, 8, 12, 16, 20, 30):
    f = 2.0 ** -k
    w = dict(W.randomized_weights(n, seed=9))
    for key in ("autoencoder/encoder_conv_layer_0_bnorm/gamma", "autoencoder/encoder_conv_layer_0_bnorm/beta"):
        w[key] = (np.asarray(w[key], dtype=np.float64) * f).astype(np.float32)
    key = "autoencoder/encoder_conv_layer_1/W"
    w[key] = (np.asarray(w[key], dtype=np.float64) / f).astype(np.float32)
    m = AEModel(W.canonical(w, n), n, np.float64)
    z64, hs = m.encode(pc.astype(np.float64), keep=True)
    sc = np.abs(z64).max()
    row = {"k": k, "largest layer-0 activation": float(hs[0].max())}
    for arith in ("f16x2", "bf16x3", "f32"):
        z = PointNetAE(w, n, encoder_arith=arith).transform(pc)
        row[arith] = float(np.abs(z - z64).max() / sc)
    print(row, flush=True)
