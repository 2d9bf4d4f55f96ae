"""Bulk Chamfer scorer (attacker/prepare_indices_for_attack.py:104-164): all-pairs Chamfer distance matrix between two
sets of clouds -- distance evaluations per second (each serves both directions) against the symmetric scan's rate in the loop."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geometric_adv_amd import ops
N = 2048
for na, nb in ((16, 16), (64, 64), (128, 256), (100, 4379)):
    rng = np.random.default_rng(na)
    a = torch.as_tensor(rng.random((na, N, 3), dtype=np.float32) - np.float32(0.5)).cuda()
    b = torch.as_tensor(rng.random((nb, N, 3), dtype=np.float32) - np.float32(0.5)).cuda()
    ops.chamfer_dist_matrix(a, b); torch.cuda.synchronize()
    reps = 3 if na * nb < 100000 else 1
    t0 = time.perf_counter()
    for _ in range(reps): out = ops.chamfer_dist_matrix(a, b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ev = na * nb * N * N
    print(json.dumps({"clouds_a": na, "clouds_b": nb, "ms": dt * 1e3, "cloud_pairs_per_s": na * nb / dt, "T_distance_evals_per_s": ev / dt / 1e12,
                      "T_directional_pair_evals_per_s": 2 * ev / dt / 1e12,
                      "full_4379x4379_matrix_s_on_one_gpu": 4379 * 4379 / (na * nb / dt)}))
from geometric_adv_amd.scorer import get_chamfer_dist_mat_full
rng = np.random.default_rng(9)
pcs = rng.random((1024, N, 3), dtype=np.float32) - np.float32(0.5)
get_chamfer_dist_mat_full(pcs[:64]); torch.cuda.synchronize()
t0 = time.perf_counter(); full = get_chamfer_dist_mat_full(pcs); dt = time.perf_counter() - t0
print(json.dumps({"full_symmetric_matrix_clouds": 1024, "s": dt, "cloud_pairs_per_s_effective": 1024 * 1024 / dt,
                  "projected_4379x4379_s_on_one_gpu": dt * (4379 / 1024) ** 2}))
