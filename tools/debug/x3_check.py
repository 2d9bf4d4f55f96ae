import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from geometric_adv_amd import weights as W
from geometric_adv_amd.autoencoder import PointNetAE
torch.manual_seed(0)
for n, B in ((2048, 4), (2048, 32), (1000, 3), (100, 2)):
    w = W.randomized_weights(n, seed=3)
    a32 = PointNetAE(w, n, encoder_arith="f32")
    ax3 = PointNetAE(w, n, encoder_arith="bf16x3")
    pc = torch.rand(B, n, 3, device="cuda") - 0.5
    z32, r32 = a32.forward(pc)
    zx3, rx3 = ax3.forward(pc)
    z32 = torch.as_tensor(z32); zx3 = torch.as_tensor(zx3)
    print(n, B, "latent max|diff|", float((z32 - zx3).abs().max()), "rel", float(((z32 - zx3).abs() / (z32.abs() + 1e-6)).max()), "max|z|", float(z32.abs().max()))
    m32 = a32.max_and_argmax(pc); mx3 = ax3.max_and_argmax(pc)
    print("   argmax equal frac", float((torch.as_tensor(m32[1]) == torch.as_tensor(mx3[1])).float().mean()))
