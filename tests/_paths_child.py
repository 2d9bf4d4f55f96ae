"""Child of test_gpu_attack.py::test_odd_sizes_all_code_paths_agree: runs a short attack for a list of cloud sizes with the
Configuration switches given as a JSON object in argv[1] (a fresh process per path) and prints one hash per size (perturbation, nearest-neighbour indices, keep-best metrics)."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geometric_adv_amd import weights as W
from geometric_adv_amd.adv_ae import AdvAE, Configuration
from geometric_adv_amd.autoencoder import PointNetAE

out = {}
paths = json.loads(sys.argv[1])
for n in [int(a) for a in sys.argv[2:]]:
    rng = np.random.default_rng(n)
    B = 3
    x = rng.random((B, n, 3), dtype=np.float32) - np.float32(0.5); gt = rng.random((B, n, 3), dtype=np.float32) - np.float32(0.5)
    w = W.randomized_weights(n); ae = PointNetAE(w, n)
    at = AdvAE("a", Configuration(batch_size=B, n_points=n, weights=w, num_iterations=10, num_iterations_thresh=3, learning_rate=0.05, **paths), ae=ae)
    at.set_inputs(x, gt, ae.transform(gt), 1.0); at.init_pert(None, reset_optimizer=True)
    at.run(0, 10, 3); torch.cuda.synchronize()
    p = at.peek()
    assert torch.isfinite(p["pert"]).all() and p["pert"].abs().max() > 0
    h = hashlib.sha1()
    for k in ("pert", "idx_a1", "idx_a2"):
        h.update(p[k].cpu().numpy().tobytes())
    m, _, _ = at.get_best(torch.as_tensor(ae.get_loss_per_pc(gt)).cuda())
    h.update(m.cpu().numpy().tobytes())
    out[str(n)] = h.hexdigest()
print("HASHES " + json.dumps(out))
