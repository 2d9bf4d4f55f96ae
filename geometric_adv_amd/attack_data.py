"""Host-side data plumbing of attacker/run_attack.py: file lookup by base name, construction of
the (source, target) pairs from the nearest-neighbour index matrix, selection helpers.  Pure numpy, like the
reference (src/adversary_utils.py); pinned to it by tests/golden/host_logic.npz.
"""
import os
import os.path as osp

import numpy as np


def load_data(data_path, file_list, base_name_list):
    """adversary_utils.py:13-23: for every base name the FIRST file of file_list containing it; one array is
    returned bare, several as a list."""
    out = []
    for base in base_name_list:
        matches = [f for f in file_list if base in f]
        if not matches:
            raise IndexError("no file containing %r in %s" % (base, data_path))
        out.append(np.load(osp.join(data_path, matches[0])))
    return out[0] if len(out) == 1 else out


def _restrict_to_correct(nn_local, correct_pred, lo, hi):
    """adversary_utils.py:88-98: keep, in order, only targets whose prediction was correct; pad each row with its
    first kept entry."""
    ok = set(np.where(correct_pred[lo:hi])[0].tolist())
    res = nn_local.copy()
    for r in range(len(res)):
        kept = np.array([i for i in res[r] if i in ok], dtype=res.dtype)
        res[r, :len(kept)] = kept
        res[r, len(kept):] = kept[0]
    return res


def prepare_data_for_attack(pc_classes, source_classes_for_attack, target_classes_for_attack, classes_data, slice_idx,
                            attack_pc_idx, num_pc_for_target, nn_idx_mat, correct_pred):
    """adversary_utils.py:26-85.  For every source class in source_classes_for_attack, every attacked cloud s of it
    (attack_pc_idx[class], indices local to the class) and every OTHER class t in target_classes_for_attack, the
    first num_pc_for_target entries of nn_idx_mat[s, class t] (local to t) pick the targets.  Returns
    (source_data, target_data), index-aligned, source rows repeated once per target; per source cloud the targets
    run class-major (all of class t1, then t2, ...)."""
    src_out, tgt_out = [], []
    for i, s_name in enumerate(pc_classes):
        if s_name not in source_classes_for_attack:
            continue
        s_lo, s_hi = slice_idx[i], slice_idx[i + 1]
        picked = attack_pc_idx[i]
        s_data = classes_data[s_lo:s_hi][picked]                       # [S, ...]
        per_class = []
        for j, t_name in enumerate(pc_classes):
            if t_name not in target_classes_for_attack or t_name == s_name:
                continue
            t_lo, t_hi = slice_idx[j], slice_idx[j + 1]
            order = nn_idx_mat[s_lo:s_hi, t_lo:t_hi][picked].copy()     # [S, |t|] local indices into class t
            if correct_pred is not None:
                order = _restrict_to_correct(order, correct_pred, t_lo, t_hi)
            per_class.append(classes_data[t_lo:t_hi][order[:, :num_pc_for_target]])   # [S, T, ...]
        tgt = np.concatenate(per_class, axis=1)                         # [S, T * classes, ...]
        reps = tgt.shape[1]
        tgt_out.append(tgt.reshape((tgt.shape[0] * reps,) + tgt.shape[2:]))
        # the reference builds np.vstack([[row] * reps for row in s_data]): for per-cloud SCALARS (ae_loss) that is a
        # (S, reps) matrix, for arrays the rows repeated along axis 0; targets of scalars come out (1, S * reps) after
        # the final vstack -- run_attack.py:131 flattens them.  Shapes are kept identical.
        src_out.append(np.repeat(s_data, reps, axis=0) if s_data.ndim > 1 else np.repeat(s_data[:, None], reps, axis=1))
    return np.vstack(src_out), np.vstack(tgt_out)


def get_quantity_at_index(quantity_list, index):
    """adversary_utils.py:101-112: out[j] = quantity[index[j], j]."""
    cols = np.arange(len(index))
    outs = [q[np.asarray(index), cols].copy() for q in quantity_list]
    return outs[0] if len(outs) == 1 else outs


def create_dir(path):
    os.makedirs(path, exist_ok=True)
    return path
