"""autoencoder/train_ae.py on MI355X: trains the victim auto-encoder and writes `models.ckpt-<epoch>`
(TF V2 checkpoint format, written without TensorFlow by tf_checkpoint.py; saver_step 50 plus the first and last epoch,
autoencoder.py:213-215) and `train_stats.txt` (epoch, loss, minutes: autoencoder.py:206-209) into --train_folder.

Limitation of the checkpoint: the bundle holds the 36 `autoencoder/*` model variables (weights, biases, BN parameters and
moving averages) -- exactly what the attack path restores (restore_ae_model filters the var_list by the 'autoencoder' prefix,
adversary_autoencoder.py:42-51).  It does NOT hold `autoencoder/epoch`, the Adam slots or the beta powers, so the
reference's own AutoEncoder.restore_model (neural_net.py:33-36, a Saver over ALL globals) would fail with NotFound on it:
it is a victim for the attack, not a resumable training state.  The reader / writer pair is pinned to the published format
and to its own round trip only; no TF-written bundle exists in this environment to test against.

Differences forced by the environment: the ShapeNet folder reader (src/in_out.load_dataset, PLY files) is out of
scope, so the training clouds come from one `.npy` of shape (n, 2048, 3) (--train_data; axes already sorted if
wanted), shuffled once per epoch with numpy instead of PointCloudDataSet.next_batch's permutation.  Multi-GPU:
launch with torchrun; every rank takes its shard of each batch and the flat gradient buffer is all-reduced (RCCL).

    python -m geometric_adv_amd.train_ae --train_data clouds.npy --train_folder log/autoencoder_victim --training_epochs 500
"""
import argparse
import os
import os.path as osp

import numpy as np


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--training_epochs', type=int, default=500, help='Number of training epochs [default: 500]')
    p.add_argument('--train_folder', type=str, default='log/autoencoder_victim')
    p.add_argument('--train_data', type=str, required=True, help='.npy of shape (n, n_points, 3)')
    p.add_argument('--batch_size', type=int, default=50)            # default_train_params, ae_templates.py:43-51
    p.add_argument('--learning_rate', type=float, default=0.0005)
    p.add_argument('--saver_step', type=int, default=50)
    p.add_argument('--seed', type=int, default=0)
    return p


def main(argv=None):
    flags = build_parser().parse_args(argv)
    import torch
    from . import dist as gdist, tf_checkpoint
    from .trainer import PointNetAETrainer, initial_weights
    rank, world, local = gdist.init()
    data = np.load(flags.train_data).astype(np.float32)
    assert data.ndim == 3 and data.shape[2] == 3, 'train_data must be (n, n_points, 3)'
    n_points = data.shape[1]
    assert flags.batch_size % world == 0, 'batch_size must divide over the ranks'
    local_bs = flags.batch_size // world
    tr = PointNetAETrainer(initial_weights(n_points, seed=flags.seed), n_points, batch_size=local_bs,
                           learning_rate=flags.learning_rate, device=torch.device('cuda', local))
    os.makedirs(flags.train_folder, exist_ok=True)
    fout = open(osp.join(flags.train_folder, 'train_stats.txt'), 'a', 1) if rank == 0 else None
    rng = np.random.default_rng(flags.seed)
    n_batches = len(data) // flags.batch_size
    stats = []
    for epoch in range(1, flags.training_epochs + 1):
        perm = rng.permutation(len(data))[:n_batches * flags.batch_size]       # same permutation on every rank
        shard = data[perm].reshape(n_batches, world, local_bs, n_points, 3)[:, rank].reshape(-1, n_points, 3)
        loss, duration = tr._single_epoch_train(shard)
        stats.append((epoch, loss, duration))
        if rank == 0:
            print("Epoch:", '%04d' % epoch, 'training time (minutes)=', "{:.4f}".format(duration / 60.0), "loss=", "{:.9f}".format(loss))
            fout.write('%04d\t%.9f\t%.4f\n' % (epoch, loss, duration / 60.0))
            if epoch % flags.saver_step == 0 or epoch == 1 or epoch == flags.training_epochs:
                tf_checkpoint.write_checkpoint(osp.join(flags.train_folder, 'models.ckpt-%d' % epoch), tr.export_weights())
    if fout:
        fout.close()
    return stats


if __name__ == '__main__':
    main()
