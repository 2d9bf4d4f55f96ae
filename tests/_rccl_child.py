"""Child of test_gpu_dist_rccl.py: a ONE-rank process group on the 'nccl' backend (= RCCL on ROCm; RCCL refuses two ranks on
one device, one it accepts) and every collective helper of geometric_adv_amd/dist.py on device tensors.  Prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as tdist
from geometric_adv_amd import dist as gdist

for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
    os.environ.pop(k, None)
rank, world, local = gdist.init("nccl", single_rank_group=True, timeout_s=120)
assert (rank, world) == (0, 1) and tdist.is_initialized() and tdist.get_backend() == "nccl"
dev = torch.device("cuda", local)
gdist.barrier()
m = torch.arange(2 * 3 * 5, dtype=torch.float32, device=dev).reshape(2, 3, 5)       # [W, n_local, 5] like the metrics gather
g = gdist.all_gather_examples(m)
t = torch.full((1000,), 2.5, device=dev)
gdist.all_reduce_sum_(t)
mx = gdist.max_over_ranks(7.25, device=dev)
# the loop's own use: an attack sharded over this (1-rank) group, gathered through RCCL
import numpy as np
from geometric_adv_amd import weights as W
from geometric_adv_amd.adv_ae import AdvAE, Configuration
from geometric_adv_amd.autoencoder import PointNetAE
n, b = 256, 2
w = W.randomized_weights(n)
ae = PointNetAE(w, n, device=dev)
rng = np.random.default_rng(5)
x = rng.random((4, n, 3), dtype=np.float32) - np.float32(0.5)
gt = rng.random((4, n, 3), dtype=np.float32) - np.float32(0.5)
conf = Configuration(batch_size=b, n_points=n, weights=w, num_iterations=6, num_iterations_thresh=3)
ref = ae.get_loss_per_pc(gt)
metrics, adv, rec, sl = gdist.attack_sharded(AdvAE("a", conf, device=dev, ae=ae), x, ae.transform(gt), gt, ref, gather_clouds=True)
plain = AdvAE("a", conf, device=dev, ae=ae).attack(x, ae.transform(gt), gt, ref)
info = gdist.backend_info()
gdist.barrier()
tdist.destroy_process_group()
print("RCCL_CHILD " + json.dumps({"gather_equal": bool(torch.equal(g, m)), "sum_ok": bool((t == 2.5).all().item()), "max": mx,
                            "attack_equal": bool(np.array_equal(metrics, plain[0]) and np.array_equal(adv, plain[1])),
                            "slice": [sl.start, sl.stop], "info": info}))
