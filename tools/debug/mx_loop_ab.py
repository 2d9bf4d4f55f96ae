"""The loop at batch B (argv, default 32) with the screened scan allowed / not allowed, alternating in one process: all-pairs (the
scan's launch has no rider: the Jacobian rides in the loss launch and the screened kernel runs from two tiles per CU on) and with the
paired search.   python tools/debug/mx_loop_ab.py [B ...]"""
import sys, json, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from geometric_adv_amd import ops, weights as W
from geometric_adv_amd.adv_ae import AdvAE, Configuration
from geometric_adv_amd.autoencoder import PointNetAE
N = 2048
for B in [int(a) for a in sys.argv[1:]] or [32]:
  rng = np.random.default_rng(0)
  x = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5); gt = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5)
  w = W.synthetic_weights(N, seed=7); ae = PointNetAE(w, N)
  res = {}
  for prune in (False, True):
      at = AdvAE("a", Configuration(batch_size=B, n_points=N, weights=w, num_iterations=2000, num_iterations_thresh=10**6, chamfer_prune=prune), ae=ae)
      at.set_inputs(x, gt, ae.transform(gt), 1.0); at.init_pert(None, reset_optimizer=True)
      at.run(0, 100, 10**6); torch.cuda.synchronize()
      best = {True: 1e9, False: 1e9}
      for rep in range(5):
          for on in (True, False):
              ops.chamfer_screen(on)
              at.run(100, 20, 10**6); torch.cuda.synchronize()
              t0 = time.perf_counter(); at.run(120, 300, 10**6); torch.cuda.synchronize()
              best[on] = min(best[on], (time.perf_counter() - t0) / 300 * 1e6)
      res["prune=%s" % prune] = {"screened_us_per_it": round(best[True], 1), "unscreened_us_per_it": round(best[False], 1)}
      del at
  print(json.dumps({"B": B, **res}), flush=True)
