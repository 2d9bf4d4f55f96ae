"""Two attacks side by side on two streams / host threads vs each alone, under both encoder arithmetics (debug)."""
import os, sys, threading
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from geometric_adv_amd import weights as W, _lib
from geometric_adv_amd.adv_ae import AdvAE, Configuration
from conftest import cloud
dev = torch.device("cuda:0")
n, b = 1024, 4
w = W.synthetic_weights(n, seed=3)
xs, gs = cloud(75, b, n), cloud(76, b, n)
emd = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0

def attack(dense, out, stream=None, arith=None):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        from geometric_adv_amd.autoencoder import PointNetAE
        at = AdvAE("adversary", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=6, num_iterations_thresh=1,
                                              emd_weight=emd, emd_dense_levels=dense), ae=PointNetAE(w, n, encoder_arith=arith))
        at.set_inputs(xs, gs, None, 1.0)
        at.init_pert(None, reset_optimizer=True)
        hist = torch.empty((6, 6, b), device=dev)
        at.run(0, 6, 1, hist)
        torch.cuda.current_stream().synchronize()
        out[dense] = hist.cpu().numpy()

for arith in os.environ.get("ARITH", "f32,bf16x3").split(","):
    for rep in range(int(os.environ.get("REPS", "3"))):
        alone, together = {}, {}
        attack(False, alone, arith=arith)
        attack(True, alone, arith=arith)
        ts = [threading.Thread(target=attack, args=(d, together, torch.cuda.Stream(dev), arith)) for d in (False, True)]
        for t in ts: t.start()
        for t in ts: t.join()
        print(arith, rep, [bool(np.array_equal(alone[d], together[d])) for d in (False, True)], "alone row0", alone[True][0, 0], "together row0", together[True][0, 0], together[False][0, 0], flush=True)
