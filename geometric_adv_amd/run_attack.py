"""attacker/run_attack.py on MI355X: same flags, same input files (looked up by base name in
<ae_folder>/eval), same outputs per shape class (adversarial_metrics.npy [W,n,5], adversarial_pc_input.npy,
adversarial_pc_recon.npy, dist_weight.npy, attack_stats.txt).

Differences forced by the environment: the victim's weights are read from <ae_folder>/models.ckpt-<restore_epoch>
by a TF-free reader of the V2 checkpoint format (tf_checkpoint.py; <ae_folder>/weights.npz with the same variable
names is the fallback), and the class list of the pickled Configuration (conf.class_names,
which needs tflearn to unpickle) is passed with --class_names (default: all of pc_classes).  Multi-GPU: launch with
torchrun; every rank attacks a contiguous run of batches and the metrics are all-gathered (dist.attack_sharded).  Adam's slots
are never reset between batches (the reference's behaviour, adv_ae.py:74) and live per rank, so an N-rank run equals N
one-process runs on the N shards (or --batch_slots N), not a one-process run over all batches, bit for bit.

    python -m geometric_adv_amd.run_attack --ae_folder log/autoencoder_victim --batch_size 10 ...
"""
import argparse
import os
import os.path as osp

import numpy as np


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--learning_rate', type=float, default=0.01)
    p.add_argument('--loss_dist_type', type=str, default='chamfer')
    p.add_argument('--loss_adv_type', type=str, default='chamfer')
    p.add_argument('--dist_weight_list', nargs='+', default=[1.0])
    p.add_argument('--max_point_pert_weight', type=float, default=0.0)
    p.add_argument('--max_point_dist_weight', type=float, default=0.0)
    p.add_argument('--num_iterations', type=int, default=500)
    p.add_argument('--num_iterations_thresh', type=int, default=400)
    p.add_argument('--batch_size', type=int, default=10)
    p.add_argument('--ae_folder', type=str, default='log/autoencoder_victim')
    p.add_argument('--restore_epoch', type=int, default=500, help='Restore epoch of a trained autoencoder [default: 500]')
    p.add_argument('--attack_pc_idx', type=str, default='log/autoencoder_victim/eval/sel_idx_rand_100_test_set_13l.npy')
    p.add_argument('--target_pc_idx_type', type=str, default='chamfer_nn_complete')
    p.add_argument('--num_pc_for_attack', type=int, default=25)
    p.add_argument('--num_pc_for_target', type=int, default=5)
    p.add_argument('--correct_pred_only', type=int, default=0)
    p.add_argument('--output_folder_name', type=str, default='attack_res')
    p.add_argument('--top_dir', type=str, default='.', help='root that --ae_folder / --attack_pc_idx are relative to')
    p.add_argument('--class_names', nargs='+', default=None, help='classes to attack / target [default: all]')
    p.add_argument('--batch_slots', type=int, default=1,
                   help='batches attacked concurrently on each GPU (AdvAE.attack; not a reference flag) [default: 1]')
    return p


def victim_weights_path(ae_dir, restore_epoch):
    """<ae_dir>/models.ckpt-<epoch> (the TF V2 checkpoint the reference restores, adversary_autoencoder.py:48 --
    read without TensorFlow by tf_checkpoint.py) when it exists, else <ae_dir>/weights.npz."""
    prefix = osp.join(ae_dir, 'models.ckpt-%d' % int(restore_epoch))
    return prefix if osp.exists(prefix + '.index') else osp.join(ae_dir, 'weights.npz')


def main(argv=None):
    flags = build_parser().parse_args(argv)
    assert flags.loss_dist_type in ['pert', 'chamfer'], 'wrong loss_dist_type: %s' % flags.loss_dist_type
    assert flags.loss_adv_type in ['latent', 'chamfer'], 'wrong loss_adv_type: %s' % flags.loss_adv_type
    assert flags.num_iterations_thresh <= flags.num_iterations, \
        'num_iterations_thresh (%d) should be smaller or equal to num_iterations (%d)' % (flags.num_iterations_thresh, flags.num_iterations)
    assert flags.target_pc_idx_type in ['latent_nn', 'chamfer_nn_complete'], 'wrong target_pc_idx_type: %s' % flags.target_pc_idx_type

    from . import dist as gdist
    from .adv_ae import AdvAE, Configuration
    from .attack_data import create_dir, load_data, prepare_data_for_attack

    rank, world, local = gdist.init()
    data_path = osp.join(flags.top_dir, flags.ae_folder, 'eval')
    files = [f for f in os.listdir(data_path) if osp.isfile(osp.join(data_path, f))]
    output_path = create_dir(osp.join(data_path, flags.output_folder_name))

    point_clouds, latent_vectors, pc_classes, slice_idx, ae_loss = load_data(
        data_path, files, ['point_clouds_test_set', 'latent_vectors_test_set', 'pc_classes', 'slice_idx_test_set', 'ae_loss_test_set'])
    assert np.all(ae_loss > 0), 'Note: not all autoencoder loss values are larger than 0 as they should!'
    nn_idx_dict = {'latent_nn': 'latent_nn_idx_test_set', 'chamfer_nn_complete': 'chamfer_nn_idx_complete_test_set'}
    nn_idx = load_data(data_path, files, [nn_idx_dict[flags.target_pc_idx_type]])
    correct_pred = None
    if flags.correct_pred_only:
        pc_labels, pc_pred_labels = load_data(data_path, files, ['pc_label_test_set', 'pc_pred_labels_test_set'])
        correct_pred = (pc_labels == pc_pred_labels)
    attack_pc_idx = np.load(osp.join(flags.top_dir, flags.attack_pc_idx))[:, :flags.num_pc_for_attack]

    classes = list(flags.class_names) if flags.class_names else [str(c) for c in pc_classes]
    conf = Configuration(batch_size=flags.batch_size, n_points=point_clouds.shape[1],
                         weights=victim_weights_path(osp.join(flags.top_dir, flags.ae_folder), flags.restore_epoch),
                         loss_adv_type=flags.loss_adv_type, loss_dist_type=flags.loss_dist_type,
                         dist_weight_list=[float(w) for w in flags.dist_weight_list],
                         max_point_pert_weight=flags.max_point_pert_weight, max_point_dist_weight=flags.max_point_dist_weight,
                         num_iterations=flags.num_iterations, num_iterations_thresh=flags.num_iterations_thresh,
                         learning_rate=flags.learning_rate, batch_slots=flags.batch_slots)
    if rank == 0:      # the reference pickles its Configuration here (run_attack.py:109; unpickling needs tflearn): same fields as JSON
        import json
        with open(osp.join(output_path, 'attack_configuration.json'), 'w') as f:
            json.dump({'class_names': classes, 'target_pc_idx_type': flags.target_pc_idx_type,
                       'num_pc_for_attack': flags.num_pc_for_attack, 'num_pc_for_target': flags.num_pc_for_target,
                       'correct_pred_only': flags.correct_pred_only, 'dist_weight_list': conf.dist_weight_list,
                       'batch_size': flags.batch_size, 'learning_rate': flags.learning_rate, 'loss_adv_type': flags.loss_adv_type,
                       'loss_dist_type': flags.loss_dist_type, 'num_iterations': flags.num_iterations,
                       'num_iterations_thresh': flags.num_iterations_thresh, 'restore_epoch': flags.restore_epoch,
                       'max_point_pert_weight': flags.max_point_pert_weight, 'max_point_dist_weight': flags.max_point_dist_weight}, f)
    import torch
    dev = torch.device("cuda", local)
    ae = None
    for i in range(len(pc_classes)):
        name = str(pc_classes[i])
        if name not in classes:
            continue
        adv = AdvAE('adversary', conf, device=dev, ae=ae)              # one graph per class in the reference (Adam state restarts)
        ae = adv.ae
        save_dir = create_dir(osp.join(output_path, name))
        prep = lambda data: prepare_data_for_attack(pc_classes, [pc_classes[i]], classes, data, slice_idx, attack_pc_idx,
                                                     flags.num_pc_for_target, nn_idx, correct_pred)
        source_pc, target_pc = prep(point_clouds)
        _, target_latent = prep(latent_vectors)
        _, target_ae_loss_ref = prep(ae_loss)
        target_ae_loss_ref = target_ae_loss_ref.reshape(-1)
        # rank 0 writes attack_stats.txt like the reference; the other ranks log THEIR batches to attack_stats.rank<r>.txt
        fout = open(osp.join(save_dir, 'attack_stats.txt' if rank == 0 else 'attack_stats.rank%d.txt' % rank), 'a', 1)
        fout.write('Train flags: %s\n' % flags)
        if world > 1:
            metrics, pc_in, pc_rec, _ = gdist.attack_sharded(adv, source_pc, target_latent, target_pc, target_ae_loss_ref,
                                                             gather_clouds=True, log_file=fout)
        else:
            metrics, pc_in, pc_rec = adv.attack(source_pc, target_latent, target_pc, target_ae_loss_ref, conf, log_file=fout)
        if fout:
            fout.close()
        if rank == 0:
            np.save(osp.join(save_dir, 'adversarial_metrics'), metrics)
            np.save(osp.join(save_dir, 'adversarial_pc_input'), pc_in)
            np.save(osp.join(save_dir, 'adversarial_pc_recon'), pc_rec)
            np.save(osp.join(save_dir, 'dist_weight'), np.array(conf.dist_weight_list))


if __name__ == '__main__':
    main()
