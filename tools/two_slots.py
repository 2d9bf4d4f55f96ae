"""Two (or more) independent B = 32 batches attacked concurrently on one GPU, each through its own attack handle, stream and
host thread (ctypes releases the GIL inside geoadv_attack_run): how much of the per-launch latency of the loop overlaps?
    python tools/two_slots.py [slots ...]"""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geometric_adv_amd import weights as W
from geometric_adv_amd.adv_ae import AdvAE, Configuration
from geometric_adv_amd.autoencoder import PointNetAE
B, N, ITERS = int(os.environ.get("GEOADV_TOOL_B", "32")), 2048, 400
w = W.synthetic_weights(N, seed=7)
for slots in [int(a) for a in sys.argv[1:]] or [1, 2, 3]:
    ats, streams = [], []
    for s in range(slots):
        rng = np.random.default_rng(100 + s)
        x = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5); gt = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            ae = PointNetAE(w, N)
            at = AdvAE("a", Configuration(batch_size=B, n_points=N, weights=w, num_iterations=ITERS + 20, num_iterations_thresh=10**6), ae=ae)
            at.set_inputs(x, gt, ae.transform(gt), 1.0); at.init_pert(None, reset_optimizer=True)
            at.run(0, 20, 10**6)
        ats.append(at); streams.append(st)
    torch.cuda.synchronize()
    def work(at, st):
        with torch.cuda.stream(st):
            at.run(20, ITERS, 10**6)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(a, s)) for a, s in zip(ats, streams)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"slots": slots, "batch": B, "iterations_per_s_all_slots": slots * ITERS / dt, "cloud_iterations_per_s": slots * B * ITERS / dt, "ms_per_iteration_per_slot": dt / ITERS * 1e3}))
    del ats
