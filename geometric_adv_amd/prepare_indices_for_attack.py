"""attacker/prepare_indices_for_attack.py --get_chamfer_nn_idx on MI355X (first half): the all-pairs Chamfer
distance matrix of the test set and the per-class-pair neighbour order the attack picks its targets from.

Same flags and files as the reference (:32-36,104-164): every call fills columns [pc_start_idx, pc_start_idx +
pc_batch_size) of chamfer_dist_mat_complete_<set>.npy (created with -1 on the first call) and, once no -1 is left, writes
chamfer_nn_idx_complete_<set>.npy = sort_dist_mat(...).  The reference needs 44 processes of 100 columns each
(runner_indices_for_attack.sh:11-15); here --pc_batch_size may simply be the whole set, or the slices may be dealt over
ranks with scorer.get_chamfer_dist_mat_sharded.

    python -m geometric_adv_amd.prepare_indices_for_attack --ae_folder log/autoencoder_victim --get_chamfer_nn_idx 1
"""
import argparse
import os
import os.path as osp
import time

import numpy as np


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--ae_folder', type=str, default='log/autoencoder_victim')
    p.add_argument('--get_chamfer_nn_idx', type=int, default=0)
    p.add_argument('--pc_start_idx', type=int, default=0)
    p.add_argument('--pc_batch_size', type=int, default=100)
    p.add_argument('--top_dir', type=str, default='.')
    p.add_argument('--device', type=str, default='cuda:0')
    return p


def get_chamfer_nn(flags):
    from .attack_data import load_data
    from .scorer import get_chamfer_dist_mat_slice, sort_dist_mat
    start_time = time.time()
    data_path = osp.join(flags.top_dir, flags.ae_folder, 'eval')
    files = [f for f in os.listdir(data_path) if osp.isfile(osp.join(data_path, f))]
    point_clouds, slice_idx = load_data(data_path, files, ['point_clouds_test_set', 'slice_idx_test_set'])
    parts = [f for f in files if 'slice_idx_test_set' in f][0].split('_')[-3:]              # e.g. test_set_13l.npy (:58-59)
    n_all = len(point_clouds)
    if flags.pc_start_idx == 0 and flags.pc_batch_size >= n_all:        # the whole matrix at once: half the work (symmetry)
        from .scorer import get_chamfer_dist_mat_full
        cur = get_chamfer_dist_mat_full(point_clouds, flags.device)
    else:
        cur = get_chamfer_dist_mat_slice(point_clouds, flags.pc_start_idx, flags.pc_batch_size, flags.device)
    assert cur.min() >= 0, 'the chamfer_dist_mat_curr matrix was not filled correctly'
    mat_path = osp.join(data_path, '_'.join(['chamfer_dist_mat_complete'] + parts))
    mat = np.load(mat_path) if osp.exists(mat_path) else -1 * np.ones([n_all, n_all], dtype=np.float32)
    mat[:, flags.pc_start_idx:flags.pc_start_idx + flags.pc_batch_size] = cur
    np.save(mat_path, mat)
    print('start index %d end index %d, out of size %d, duration (minutes): %.2f' %
          (flags.pc_start_idx, min(flags.pc_start_idx + flags.pc_batch_size, n_all), n_all, (time.time() - start_time) / 60.0))
    if mat.min() >= 0:
        np.save(osp.join(data_path, '_'.join(['chamfer_nn_idx_complete'] + parts)), sort_dist_mat(mat, slice_idx))


def main(argv=None):
    flags = build_parser().parse_args(argv)
    if flags.get_chamfer_nn_idx:
        get_chamfer_nn(flags)


if __name__ == '__main__':
    main()
