"""Attack throughput against a TRAINED victim (the benchmark's victim has random weights): trains the AE on synthetic
shapes with this package's trainer, then times the attack loop with and without the paired grid search and prints how
far points move and how close reconstructions get to their targets.
    python tools/trained_victim_attack.py"""
import os, subprocess, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def shapes(rng, count, n):
    u = rng.standard_normal((count, n, 3)).astype(np.float32)
    u /= np.linalg.norm(u, axis=2, keepdims=True)
    scale = rng.uniform(0.15, 0.45, size=(count, 1, 3)).astype(np.float32)
    box = rng.random((count, 1, 1)) < 0.5
    return (np.where(box, np.clip(u * 3.0, -1.0, 1.0), u) * scale).astype(np.float32)


def main():
    from geometric_adv_amd.trainer import PointNetAETrainer, initial_weights
    from geometric_adv_amd.autoencoder import PointNetAE
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd import ops
    N, B = 2048, 32
    rng = np.random.default_rng(0)
    if len(sys.argv) > 1:                                     # child: time the loop with the grid search on (argv[2] = 1) or off
        w = dict(np.load(sys.argv[1]))
        w = {k.replace("__", "/"): v for k, v in w.items()}
        ae = PointNetAE(w, N)
        src, tgt = shapes(rng, B, N), shapes(rng, B, N)
        at = AdvAE("a", Configuration(batch_size=B, n_points=N, weights=w, num_iterations=520, num_iterations_thresh=10**6,
                                      chamfer_prune=sys.argv[2] == "1"), ae=ae)
        at.set_inputs(src, tgt, ae.transform(tgt), 1.0); at.init_pert(None, reset_optimizer=True)
        at.run(0, 20, 10**6); torch.cuda.synchronize()
        t0 = time.perf_counter(); at.run(20, 500, 10**6); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 500
        p = at.peek()
        pn = p["pert"].norm(dim=2).flatten()
        d1, _, d2, _ = ops.nn_distance(p["recon"], torch.as_tensor(tgt).cuda())
        print(json.dumps({"prune": sys.argv[2], "it_per_s": 1 / dt, "ms": dt * 1e3,
                          "pert_median": pn.median().item(), "pert_p99": torch.quantile(pn[:1000000], 0.99).item(), "pert_max": pn.max().item(),
                          "recon_to_target_nn_dist_median": d1.sqrt().median().item(), "target_to_recon_nn_dist_median": d2.sqrt().median().item()}))
        return
    tr = PointNetAETrainer(initial_weights(N, seed=2), N, batch_size=50, learning_rate=0.001)
    data = shapes(rng, 400, N)
    for ep in range(40):
        loss, _ = tr._single_epoch_train(data)
    print("trained victim: reconstruction loss %.5f after 320 steps" % loss)
    from geometric_adv_amd import weights as W
    path = "/tmp/trained_victim.npz"
    W.save_npz(path, tr.export_weights())
    for flag in ("1", "0"):
        subprocess.run([sys.executable, os.path.abspath(__file__), path, flag], check=True)


if __name__ == "__main__":
    main()
