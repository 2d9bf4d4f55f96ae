"""Golden trajectory of the whole attack from the numpy model (fp64) -- TEST INFRASTRUCTURE.

BASELINE config 0 shape: a single source/target pair, N = 1024, 10 attack iterations (plus a
B = 2, N = 256 latent-space case).  The network part of the model is unpinned against the
reference (no TensorFlow here, see oracle/attack_model.py); these vectors pin the GPU path to the
model and guard against regressions.

    python oracle/make_golden_attack.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from geometric_adv_amd import weights as W            # noqa: E402
from geometric_adv_amd.adversary import init_pert_value   # noqa: E402
from oracle.attack_model import AEModel, AttackModel   # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def cloud(seed, b, n):
    rng = np.random.default_rng(seed)
    return (rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5)).astype(np.float32)


def run(name, b, n, iters, adv_type, dist_type, dw, wseed):
    w = W.randomized_weights(n, seed=wseed)
    model = AEModel(W.canonical(w, n), n, np.float64)
    x, gt = cloud(1000 + n, b, n), cloud(2000 + n, b, n)
    tz = model.encode(gt)
    am = AttackModel(model, x, gt, tz, np.full(b, dw), adv_type, dist_type, lr=0.01)
    p0 = init_pert_value(b, n)
    am.init_pert(p0)
    hist = np.zeros((iters, 6, b))
    for it in range(iters):
        am.step()
        f = am.forward()
        fourth = f["loss_max"] if dist_type == "pert" else f["max_dist"]
        hist[it] = [f["loss_adv"], f["loss_dist"], f["loss_pert"], fourth, f["input_dist"], f["loss_ae"]]
    f = am.forward()
    return {f"{name}_n": np.int32(n), f"{name}_wseed": np.int32(wseed), f"{name}_x": x, f"{name}_gt": gt,
            f"{name}_tz": tz.astype(np.float32), f"{name}_dw": np.float32(dw), f"{name}_hist": hist,
            f"{name}_pert": am.pert, f"{name}_recon": f["recon"], f"{name}_adv_type": np.array(adv_type),
            f"{name}_dist_type": np.array(dist_type)}


def main():
    g = {}
    g.update(run("config0", 1, 1024, 10, "chamfer", "chamfer", 1.0, 3))
    g.update(run("latent", 2, 256, 10, "latent", "chamfer", 150.0, 4))
    g["cases"] = np.array(["config0", "latent"])
    np.savez_compressed(os.path.join(OUT, "attack_trajectory.npz"), **g)
    print(os.path.getsize(os.path.join(OUT, "attack_trajectory.npz")) // 1024, "KiB")
    print(g["config0_hist"][:, 5, 0])


if __name__ == "__main__":
    main()
