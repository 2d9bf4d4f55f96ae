"""Throughput of the AE training step at default_train_params: batch 50 x 2048 points.
    python tools/train_bench.py [--steps 50] [--batch 50] [--points 2048]
Prints one JSON line: steps/s, clouds/s, and the MFMA roofline fraction of the whole step
(algorithmic flop = 3 x 2*B*N*90304 encoder [forward, data gradient, weight gradient] + decoder)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=50)
    ap.add_argument("--points", type=int, default=2048)
    a = ap.parse_args()
    from geometric_adv_amd.trainer import PointNetAETrainer, initial_weights
    B, N = a.batch, a.points
    tr = PointNetAETrainer(initial_weights(N, seed=1), N, batch_size=B)
    rng = np.random.default_rng(0)
    x = torch.as_tensor(rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5)).cuda()
    for _ in range(a.warmup):
        tr.partial_fit(x, want_recon=False, sync=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.partial_fit(x, want_recon=False, sync=False)
    t_enq = (time.perf_counter() - t0) / a.steps               # host time to enqueue one step
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    flop = 3 * 2.0 * B * N * 90304 + 3 * 2.0 * B * (98304 + 768 * N)
    print(json.dumps({"metric": "AE training steps/sec", "batch": B, "n_points": N, "steps_per_s": 1.0 / dt,
                      "ms_per_step": dt * 1e3, "host_enqueue_ms_per_step": t_enq * 1e3, "clouds_per_s": B / dt, "algorithmic_gflop_per_step": flop / 1e9,
                      "achieved_tflops": flop / dt / 1e12, "frac_of_fp32_mfma_peak": flop / dt / 157.3e12}))


if __name__ == "__main__":
    main()
