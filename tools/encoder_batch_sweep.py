import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from geometric_adv_amd import weights as W
from geometric_adv_amd.autoencoder import PointNetAE
N=2048
ae=PointNetAE(W.synthetic_weights(N), N)
for B in (32, 64, 256, 1024):
    x=torch.rand((B,N,3),device="cuda:0")-0.5
    for _ in range(5): ae.forward(x, want_recon=False)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ae.forward(x, want_recon=False)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/50
    print("B=%d encode %.3f ms -> %.1f TFLOP/s" % (B, ms, 2*90304*B*N/ms/1e9))
