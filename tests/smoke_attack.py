"""One small invocation of the whole hot path on the GPU, checked against the oracle
(imported only from __graft_entry__.smoke(); the oracle is the checker, never the path).  Lives under tests/: nothing
inside the product package geometric_adv_amd/ may reference the oracle."""
import numpy as np


def run(dev):
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from oracle.attack_model import AEModel, AttackModel
    from oracle.cpu_oracle import Oracle
    n, b = 256, 2
    w = W.synthetic_weights(n)
    rng = np.random.default_rng(1)
    x = (rng.random((b, n, 3), dtype=np.float32) - 0.5).astype(np.float32)
    gt = (rng.random((b, n, 3), dtype=np.float32) - 0.5).astype(np.float32)
    conf = Configuration(batch_size=b, n_points=n, weights=w, num_iterations=4, num_iterations_thresh=2)
    at = AdvAE("adversary", conf, device=dev)
    at.set_inputs(x, gt, None, 1.0)
    at.init_pert(None, reset_optimizer=True)
    at.run(0, 3, 2)
    s = {k: v.cpu().numpy() for k, v in at.peek().items()}
    model = AEModel(W.canonical(w, n), n)
    am = AttackModel(model, x, gt, None, np.ones(b))
    am.init_pert(s["pert"])
    f = am.forward()
    assert np.allclose(s["recon"], f["recon"], atol=2e-6), "reconstruction differs from the model"
    o = Oracle()
    _, i1, _, i2 = o.nn_distance(s["recon"], gt)
    assert np.array_equal(s["idx_r1"], i1) and np.array_equal(s["idx_r2"], i2), "NN indices differ from the oracle"
    at.run(3, 1, 2)
    g = am.gradient(am.forward(idx_override=(s["idx_r1"], s["idx_r2"], s["idx_a1"], s["idx_a2"])))
    got = at.peek()["grad"].cpu().numpy()
    sc = np.abs(g).max()
    assert np.allclose(got / sc, g / sc, atol=1e-4), "gradient differs from the model"
