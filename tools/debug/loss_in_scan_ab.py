"""The loss + gradient workgroups as riders of the symmetric scan's launch (Configuration.loss_in_scan) against a launch of their own:
the trajectories must agree bit for bit (same bodies, same order of every sum), then us per iteration, alternating.
    python tools/debug/loss_in_scan_ab.py [B ...]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from geometric_adv_amd import weights as W
from geometric_adv_amd.adv_ae import AdvAE, Configuration
from geometric_adv_amd.autoencoder import PointNetAE
N = 2048
ITERS = int(os.environ.get("ITERS", "120"))          # iterations compared bit for bit (a soak: ITERS=3000)
w = W.synthetic_weights(N, seed=7); ae = PointNetAE(w, N)
for B in [int(a) for a in sys.argv[1:]] or [32, 16, 8, 64]:
    rng = np.random.default_rng(B)
    x = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5); gt = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5)
    for prune in (True, False):
        hs, ats = {}, {}
        for on in (True, False):
            at = AdvAE("a", Configuration(batch_size=B, n_points=N, weights=w, num_iterations=ITERS + 3000, num_iterations_thresh=50, chamfer_prune=prune,
                                          loss_in_scan="always" if on else False), ae=ae)
            at.set_inputs(x, gt, ae.transform(gt), 1.0); at.init_pert(None, reset_optimizer=True)
            h = torch.empty((ITERS, 6, B), device=ae.device)
            at.run(0, ITERS, 50, h); at.status()
            hs[on] = h.cpu().numpy(); ats[on] = at
        same = bool(np.array_equal(hs[True], hs[False])) and all(torch.equal(ats[True].peek()[k], ats[False].peek()[k]) for k in ("pert", "idx_r1", "idx_a1", "grad"))
        best = {True: 1e9, False: 1e9}
        for rep in range(5):
            for on in (True, False):
                at = ats[on]
                at.run(ITERS, 20, 10 ** 6); torch.cuda.synchronize()
                t0 = time.perf_counter(); at.run(ITERS + 20, 400, 10 ** 6); torch.cuda.synchronize()
                best[on] = min(best[on], (time.perf_counter() - t0) / 400 * 1e6)
        print(json.dumps({"B": B, "prune": prune, "bit_identical_iterations": ITERS if same else False, "riding_us_per_it": round(best[True], 2), "own_launch_us_per_it": round(best[False], 2)}))
        del ats
