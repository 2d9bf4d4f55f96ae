"""attacker/get_dists_per_point.py on MI355X (second half): for every attacked class, the distance of every
adversarial point to its nearest neighbour in the clean source cloud,

    adversarial_pc_input_dists.npy [W, n_examples, N] = sqrt(dists_first_to_second) of nn_distance(adversarial_pc_input, source_pc)

(get_dists_per_point.py:70-126; the #outlier metric of the paper thresholds it).  Same flags, same files.  With
--do_sanity_checks 1 the Chamfer distance recomputed by the op from the SAVED adversarial clouds must equal the
source_chamfer_dist the attack recorded (adversarial_metrics[:, :, 2]) under np.array_equal (:114-115) -- which holds here
because the loop's metric and ops.chamfer_per_pc share one kernel arithmetic and one summation order.

The attack's settings come from <output_folder>/attack_configuration.json, written by geometric_adv_amd.run_attack (the
reference unpickles a Configuration, which needs tflearn).

    python -m geometric_adv_amd.get_dists_per_point --ae_folder log/autoencoder_victim --do_sanity_checks 1
"""
import argparse
import json
import os
import os.path as osp

import numpy as np


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--ae_folder', type=str, default='log/autoencoder_victim')
    p.add_argument('--attack_pc_idx', type=str, default='log/autoencoder_victim/eval/sel_idx_rand_100_test_set_13l.npy')
    p.add_argument('--do_sanity_checks', type=int, default=0)
    p.add_argument('--output_folder_name', type=str, default='attack_res')
    p.add_argument('--top_dir', type=str, default='.', help='root that --ae_folder / --attack_pc_idx are relative to')
    p.add_argument('--device', type=str, default='cuda:0')
    return p


def dists_per_point(adversarial_pc_input, source_pc, device='cuda:0', chamfer_batch_size=10):
    """-> (squared dists_first_to_second [W, n, N], chamfer_dist [W, n]) of nn_distance(adversarial_pc_input[w], source_pc),
    walked in batches of chamfer_batch_size like get_dists_per_point.py:103-117 (the batch size does not change a value)."""
    import torch
    from . import ops
    W, n_ex = adversarial_pc_input.shape[:2]
    d2 = -1 * np.ones(adversarial_pc_input.shape[:3], dtype=np.float32)
    ch = np.zeros((W, n_ex), np.float32)
    src = torch.as_tensor(np.ascontiguousarray(source_pc, dtype=np.float32)).to(device)
    for j in range(W):
        adv = torch.as_tensor(np.ascontiguousarray(adversarial_pc_input[j], dtype=np.float32)).to(device)
        for k in range(0, n_ex, chamfer_batch_size):
            first, _, second, _ = ops.nn_distance(adv[k:k + chamfer_batch_size], src[k:k + chamfer_batch_size])
            d2[j, k:k + chamfer_batch_size] = first.cpu().numpy()
            ch[j, k:k + chamfer_batch_size] = ops.chamfer_per_pc(first, second).cpu().numpy()
    return d2, ch


def main(argv=None):
    flags = build_parser().parse_args(argv)
    from .attack_data import load_data, prepare_data_for_attack
    data_path = osp.join(flags.top_dir, flags.ae_folder, 'eval')
    files = [f for f in os.listdir(data_path) if osp.isfile(osp.join(data_path, f))]
    output_path = osp.join(data_path, flags.output_folder_name)
    with open(osp.join(output_path, 'attack_configuration.json')) as f:
        conf = json.load(f)
    point_clouds, pc_classes, slice_idx = load_data(data_path, files, ['point_clouds_test_set', 'pc_classes', 'slice_idx_test_set'])
    nn_idx_dict = {'latent_nn': 'latent_nn_idx_test_set', 'chamfer_nn_complete': 'chamfer_nn_idx_complete_test_set'}
    nn_idx = load_data(data_path, files, [nn_idx_dict[conf['target_pc_idx_type']]])
    correct_pred = None
    if conf['correct_pred_only']:
        pc_labels, pc_pred_labels = load_data(data_path, files, ['pc_label_test_set', 'pc_pred_labels_test_set'])
        correct_pred = (pc_labels == pc_pred_labels)
    attack_pc_idx = np.load(osp.join(flags.top_dir, flags.attack_pc_idx))[:, :conf['num_pc_for_attack']]
    classes = conf['class_names']
    for i in range(len(pc_classes)):
        name = str(pc_classes[i])
        if name not in classes:
            continue
        source_pc, _ = prepare_data_for_attack(pc_classes, [pc_classes[i]], classes, point_clouds, slice_idx, attack_pc_idx,
                                               conf['num_pc_for_target'], nn_idx, correct_pred)
        load_dir = osp.join(output_path, name)
        adversarial_metrics = np.load(osp.join(load_dir, 'adversarial_metrics.npy'))
        adversarial_pc_input = np.load(osp.join(load_dir, 'adversarial_pc_input.npy'))
        source_chamfer_dist = adversarial_metrics[:, :, 2]
        d2, ch = dists_per_point(adversarial_pc_input, source_pc, flags.device)
        if flags.do_sanity_checks:
            assert np.array_equal(ch, source_chamfer_dist), 'mismatch for chamfer dist'             # get_dists_per_point.py:114-115
        assert np.all(d2 >= 0), 'The adversarial_pc_input_dists was not filled correctly'
        # the distances from nn_distance() are squared: take a square root of them before saving (:122-123)
        np.save(osp.join(load_dir, 'adversarial_pc_input_dists'), np.sqrt(d2))


if __name__ == '__main__':
    main()
