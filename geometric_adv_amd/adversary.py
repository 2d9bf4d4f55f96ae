"""Adversary (src/adversary.py:9-57): the additive perturbation and its norms, host side."""
import numpy as np


def truncated_normal(shape, stddev, seed):
    """Values of N(0, stddev) re-drawn until within two standard deviations, like
    tf.truncated_normal (adversary.py:28).  TF's Philox stream is not reproducible without TF, so
    the stream is numpy's PCG64 seeded with the same seed; the distribution is the same."""
    rng = np.random.default_rng(seed)
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * stddev).astype(np.float32)


def init_pert_value(batch_size, num_points, stddev=0.0000001, seed=55):
    """Adversary.init_pert (adversary.py:27-28)."""
    return truncated_normal((batch_size, num_points, 3), stddev, seed)


def get_pert_loss_np(pert, sqrt=True):
    """The numpy restatement the reference carries in-file (adversary.py:77-90)."""
    per_point = np.sum(np.square(pert), axis=2)
    norm_sq = np.sum(per_point, axis=1)
    max_sq = np.max(per_point, axis=1)
    return (np.sqrt(norm_sq), np.sqrt(max_sq)) if sqrt else (norm_sq, max_sq)
