"""The trained-victim leg of bench.py with its box shapes as they are (sphere directions x 3 clipped to the cube: ~22 % of a box cloud's
points are exact duplicates at the corners) and with a duplicate-free box surface (directions projected onto the cube)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench


def cube_projected(rng, count, n):
    u = rng.standard_normal((count, n, 3)).astype(np.float32)
    u /= np.linalg.norm(u, axis=2, keepdims=True)
    scale = rng.uniform(0.15, 0.45, size=(count, 1, 3)).astype(np.float32)
    box = rng.random((count, 1, 1)) < 0.5
    return (np.where(box, u / np.abs(u).max(axis=2, keepdims=True), u) * scale).astype(np.float32)


dev = torch.device("cuda:0")
for name, fn in (("clipped (bench.py)", bench.victim_shapes), ("cube-projected", cube_projected)):
    bench.victim_shapes = fn
    s = fn(np.random.default_rng(0), 32, 2048)
    dup = [2048 - len(np.unique(c, axis=0)) for c in s]
    out = bench.trained_victim_leg(dev, 200, 20)
    print(json.dumps({"shapes": name, "duplicate_points_per_cloud_max": int(max(dup)), "clouds_with_duplicates": int(sum(d > 0 for d in dup)),
                      "its": {k: round(out[k]["attack_iterations_per_sec"]) for k in ("grid_search", "all_pairs", "adaptive_default") if "attack_iterations_per_sec" in out.get(k, {})},
                      "handed_back": out["grid_search"].get("clouds_handed_back_of_32_along_the_attack"), "pert_norm_median": out.get("pert_norm_median"),
                      "victim": out["victim"][-30:]}), flush=True)
