"""Golden vectors for the HOST-side numpy bookkeeping of the path -- TEST INFRASTRUCTURE.

The reference modules cannot be imported here (src/general_utils.py imports seaborn, src/adv_ae.py TensorFlow),
but the functions we need are plain numpy.  Their source LINES are read from /root/reference at run time and
exec'd in a namespace that holds only numpy -- nothing is copied into the repo, no stand-in module is written.
Only inputs and the reference's outputs are stored (tests/golden/host_logic.npz).

  src/adversary_utils.py:26-85    prepare_data_for_attack        (+ :88-98 get_idx_for_correct_pred)
  src/adversary_utils.py:101-112  get_quantity_at_index
  src/adversary_utils.py:149-178  get_outlier_pc_inlier_pc
  src/general_utils.py:64-91      get_complementary_points / get_complementary_idx
  src/ae_utils.py:12-80           get_critical_points / get_critical_pc_non_critical_pc
  attacker/prepare_indices_for_attack.py:167-180  sort_dist_mat
"""
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def lines(path, a, b):
    with open(os.path.join(REF, path)) as f:
        src = f.read().splitlines()
    return "\n".join(src[a - 1:b]) + "\n"


def main():
    ns = {"np": np, "osp": os.path}
    exec(lines("src/adversary_utils.py", 26, 112), ns)
    exec(lines("src/adversary_utils.py", 149, 178), ns)
    exec(lines("src/general_utils.py", 64, 91), ns)
    exec(lines("src/ae_utils.py", 12, 80), ns)
    g = {}
    rng = np.random.default_rng(5)

    # ---- prepare_data_for_attack: 3 classes of 6 / 5 / 7 clouds of 16 points ----
    sizes = [6, 5, 7]
    slice_idx = np.concatenate([[0], np.cumsum(sizes)])
    n_all = slice_idx[-1]
    pc_classes = np.array(["chair", "table", "car"])
    pcs = rng.random((n_all, 16, 3)).astype(np.float32)
    lat = rng.random((n_all, 8)).astype(np.float32)
    loss = rng.random(n_all).astype(np.float32) + 0.1
    attack_pc_idx = np.stack([rng.permutation(min(sizes))[:3] for _ in sizes])            # 3 clouds per class
    # nn_idx_mat[s, slice(t)] = ordering of class-t clouds for source s (indices local to class t)
    nn_idx = np.zeros((n_all, n_all), np.int16)
    for s in range(n_all):
        for t in range(3):
            nn_idx[s, slice_idx[t]:slice_idx[t + 1]] = rng.permutation(sizes[t])
    correct = rng.random(n_all) > 0.3
    correct[[0, 6, 11]] = True
    for name, data in [("pc", pcs), ("lat", lat), ("loss", loss)]:
        for cp_name, cp in [("all", None), ("correct", correct)]:
            src, tgt = ns["prepare_data_for_attack"](pc_classes, ["table"], list(pc_classes), data, slice_idx, attack_pc_idx, 2,
                                                     nn_idx, cp)
            g[f"prep_{name}_{cp_name}_src"], g[f"prep_{name}_{cp_name}_tgt"] = src, tgt
    src, tgt = ns["prepare_data_for_attack"](pc_classes, list(pc_classes), ["chair", "car"], pcs, slice_idx, attack_pc_idx, 3, nn_idx, None)
    g["prep_multi_src"], g["prep_multi_tgt"] = src, tgt
    g.update(prep_classes=pc_classes, prep_slice_idx=slice_idx, prep_pcs=pcs, prep_lat=lat, prep_loss=loss,
             prep_attack_idx=attack_pc_idx, prep_nn_idx=nn_idx, prep_correct=correct)

    q = rng.random((4, 6, 5)).astype(np.float32)
    idx = rng.integers(0, 4, size=6)
    g.update(gq_quantity=q, gq_index=idx, gq_out=ns["get_quantity_at_index"]([q], idx))

    # ---- get_outlier_pc_inlier_pc ----
    pc = rng.random((4, 20, 3)).astype(np.float32)
    kd = rng.random((4, 20)).astype(np.float32) * 0.08
    kd[2] = 0.0                                    # no outliers at all
    kd[3] = 1.0                                    # everything is an outlier
    o_pc, o_idx, o_num, i_pc = ns["get_outlier_pc_inlier_pc"](pc, kd, 0.04)
    g.update(out_pc=pc, out_knn=kd, out_thresh=np.float32(0.04), out_outlier_pc=o_pc, out_outlier_idx=o_idx,
             out_outlier_num=o_num, out_inlier_pc=i_pc)

    # ---- critical points ----
    pc = rng.random((3, 40, 3)).astype(np.float32)
    pre = np.maximum(rng.standard_normal((3, 40, 12)), 0).astype(np.float32)
    pre[:, :, 5] = 0                               # a channel that is 0 for the whole cloud
    pre[1, 7, :4] = 9.0                            # one point critical for several channels
    cp, ci, cn, crit_pc, noncrit_pc = ns["get_critical_pc_non_critical_pc"](pc, pre)
    g.update(crit_in_pc=pc, crit_pre=pre, crit_points=cp, crit_idx=ci, crit_num=cn, crit_pc=crit_pc, crit_noncrit_pc=noncrit_pc)

    # ---- sort_dist_mat (attacker/prepare_indices_for_attack.py:167-183; reads the module globals range_num_classes, slice_idx) ----
    sizes = [4, 3, 5]
    sl = np.concatenate([[0], np.cumsum(sizes)])
    pts = rng.random((sl[-1], 6))
    dm = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1)).astype(np.float32)      # symmetric, zero diagonal, no ties off it
    ns2 = {"np": np, "range_num_classes": range(len(sizes)), "slice_idx": sl}
    exec(lines("attacker/prepare_indices_for_attack.py", 167, 180), ns2)
    g.update(sdm_dist=dm, sdm_slice_idx=sl, sdm_nn_idx=ns2["sort_dist_mat"](dm))

    np.savez_compressed(os.path.join(OUT, "host_logic.npz"), **g)
    print("host_logic.npz", os.path.getsize(os.path.join(OUT, "host_logic.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
