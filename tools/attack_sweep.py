"""Attack throughput across batch sizes / point counts (one GPU): iterations/s and cloud-iterations/s.
    python tools/attack_sweep.py > sweep.json
B = 4 is what each GPU of an 8-GPU strong-scaled config 2 would run; N = 8192 is BASELINE config 5."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from geometric_adv_amd import weights as W
from geometric_adv_amd.adv_ae import AdvAE, Configuration
from geometric_adv_amd.autoencoder import PointNetAE


def run(B, N, iters, **cfg):
    rng = np.random.default_rng(B * 7 + N)
    x = (rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5))
    gt = (rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5))
    w = W.synthetic_weights(N, seed=7)
    ae = PointNetAE(w, N)
    conf = Configuration(batch_size=B, n_points=N, weights=w, num_iterations=iters + 20, num_iterations_thresh=iters, **cfg)
    at = AdvAE("adversary", conf, ae=ae)
    at.set_inputs(x, gt, ae.transform(gt), 1.0)
    at.init_pert(None, reset_optimizer=True)
    at.run(0, 20, iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    at.run(20, iters, iters)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return {"batch": B, "n_points": N, "iterations_per_s": 1.0 / dt, "ms_per_iteration": dt * 1e3,
            "cloud_iterations_per_s": B / dt}


if __name__ == "__main__":
    out = []
    for B, N, it in [(1, 2048, 300), (4, 2048, 300), (8, 2048, 300), (16, 2048, 300), (32, 2048, 300), (64, 2048, 200),
                     (128, 2048, 100), (256, 2048, 60), (1024, 2048, 20), (32, 1024, 300), (32, 4096, 100), (32, 8192, 40), (256, 8192, 8)]:
        out.append(run(B, N, it))
        sys.stderr.write(json.dumps(out[-1]) + "\n")
    print(json.dumps(out, indent=1))
