import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from tests.test_gpu_train import _setup, _clouds, _rel
from oracle.train_model import PARAM_GROUPS
for (n,b,seed) in [(256,3,3),(256,3,4),(128,4,3),(512,8,3),(2048,10,3)]:
    w,tr,tm=_setup(n,b)
    x=_clouds(seed,b,n)
    recon,loss=tr.forward_backward(x)
    g=tr.gradients()
    lr,G,c=tm.loss_and_grads(x)
    print(n,b,seed,"loss rel",abs(loss.item()-lr)/lr, "recon max", np.abs(recon.cpu().numpy()-c["recon"]).max())
    for k in PARAM_GROUPS:
        if k=="enc_b": 
            print("  enc_b max", [float(np.abs(a).max()) for a in g[k]]); continue
        print("  ",k,["%.1e"%_rel(g[k][j],G[k][j]) for j in range(len(G[k]))])
