"""Attack iteration time with the combined Chamfer + approx-EMD adversarial loss (BASELINE config 4's loss)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geometric_adv_amd import weights as W, ops
from geometric_adv_amd.adv_ae import AdvAE, Configuration
from geometric_adv_amd.autoencoder import PointNetAE
N = 2048
CASES = {8: 40, 32: 20, 128: 8}
for B, iters in ([(int(a), CASES.get(int(a), 10)) for a in sys.argv[1:]] or list(CASES.items())):
    rng = np.random.default_rng(B)
    x = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5); gt = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5)
    w = W.synthetic_weights(N, seed=7); ae = PointNetAE(w, N)
    at = AdvAE("a", Configuration(batch_size=B, n_points=N, weights=w, num_iterations=iters + 5, num_iterations_thresh=10**6, emd_weight=1.0), ae=ae)
    at.set_inputs(x, gt, ae.transform(gt), 1.0); at.init_pert(None, reset_optimizer=True)
    at.run(0, 3, 10**6); torch.cuda.synchronize()
    t0 = time.perf_counter(); at.run(3, iters, 10**6); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    xs, ys = torch.as_tensor(x).cuda(), torch.as_tensor(gt).cuda()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(3): m = ops.approx_match(xs, ys)
    torch.cuda.synchronize(); tm = (time.perf_counter() - t1) / 3
    def timed(f, reps=5):
        f(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
    tc = timed(lambda: ops.match_cost(xs, ys, m)); tg = timed(lambda: ops.match_cost_grad(xs, ys, m))
    tf = timed(lambda: ops.emd_cost_grad1(xs, ys))
    ops.emd_sparse_levels(False)
    tm_dense = timed(lambda: ops.approx_match(xs, ys), 3); tf_dense = timed(lambda: ops.emd_cost_grad1(xs, ys))
    ops.emd_sparse_levels(True)
    tm = timed(lambda: ops.approx_match(xs, ys), 3) * 1e-3
    tmr = timed(lambda: ops.approx_match(xs, ys, reference_weights=True), 3)
    tfr = timed(lambda: ops.emd_cost_grad1(xs, ys, reference_weights=True))
    pw = 21.4 * B * N * N                    # pair-weights per approx_match: 10 levels x (B + C + A) sweeps, minus the missing first C / last A
    print(json.dumps({"batch": B, "ms_per_iteration_chamfer_plus_emd": dt * 1e3, "approx_match_ms": tm * 1e3,
                      "match_cost_ms": tc, "match_cost_grad_ms": tg, "fused_levels_cost_grad1_ms": tf,
                      "every_sweep_dense": {"approx_match_ms": tm_dense, "fused_levels_cost_grad1_ms": tf_dense},
                      "reference_weights": {"approx_match_ms": tmr, "fused_levels_cost_grad1_ms": tfr},
                      "sweep_Tpair_weights_per_s_incl_plan_write": pw / tm / 1e12,
                      "match_bytes_GB": B * N * N * 4 / 1e9}))
