"""Bulk Chamfer scorer (attacker/prepare_indices_for_attack.py:104-164): the all-pairs Chamfer
distance matrix over a set of clouds, from which the attack's target candidates are ranked.

The reference fills `chamfer_dist_mat[:, start:start+size]` with one process per 100-column slice (44 launches,
runner_indices_for_attack.sh:11-15), 10 x 10 tiled pairs per sess.run.  Here one call computes a slice with the
pair-indexed symmetric kernel (no tiled copies); slices shard over GPUs exactly like the reference's processes.
"""
import numpy as np
import torch

from . import dist as gdist
from . import ops


def get_chamfer_dist_mat_slice(point_clouds, pc_start_idx, pc_batch_size, device="cuda:0", row_chunk=512):
    """chamfer_dist_mat_curr of prepare_indices_for_attack.py:116-139: shape (num_all, num_curr), entry [i, j] =
    Chamfer(source = point_clouds[start + j], target = point_clouds[i])."""
    pcs = np.ascontiguousarray(point_clouds, dtype=np.float32)
    cur = torch.as_tensor(pcs[pc_start_idx:pc_start_idx + pc_batch_size]).to(device)
    out = np.empty((len(pcs), cur.shape[0]), np.float32)
    for r0 in range(0, len(pcs), row_chunk):
        rows = torch.as_tensor(pcs[r0:r0 + row_chunk]).to(device)
        # nn_distance(source, target): source = current cloud j, target = cloud i  ->  pair (j, i)
        out[r0:r0 + row_chunk] = ops.chamfer_dist_matrix(cur, rows).T.cpu().numpy()
    return out


def sort_dist_mat_rows(dist_mat):
    """Ascending neighbour order per row (the argsort at the heart of sort_dist_mat)."""
    return np.argsort(dist_mat, axis=1).astype(np.int16 if dist_mat.shape[1] < 32768 else np.int32)


def sort_dist_mat(dist_mat, slice_idx):
    """prepare_indices_for_attack.py:167-183: for every (source class i, target class j) block of the distance matrix,
    the ascending order of the targets of class j for each source of class i, as int16 indices LOCAL to class j (they
    start from 0 in every block; for i == j the first entry is the instance itself at distance 0, which
    prepare_data_for_attack skips).  slice_idx: class boundaries, len = classes + 1."""
    nn_idx = -1 * np.ones(dist_mat.shape, dtype=np.int16)
    k = len(slice_idx) - 1
    for i in range(k):
        for j in range(k):
            block = dist_mat[slice_idx[i]:slice_idx[i + 1], slice_idx[j]:slice_idx[j + 1]]
            nn_idx[slice_idx[i]:slice_idx[i + 1], slice_idx[j]:slice_idx[j + 1]] = np.argsort(block, axis=1).astype(np.int16)
    assert nn_idx.min() >= 0, 'the nn_idx matrix was not filled correctly'
    return nn_idx


def get_chamfer_dist_mat_full(point_clouds, device="cuda:0", block=256, rank=0, world=1):
    """The complete (num_all, num_all) matrix of prepare_indices_for_attack.py:104-153 in one call, at half the work: the
    Chamfer distance of a pair is the same whichever cloud is called the source (mean of both directions, equal point counts),
    and this build's kernel returns the same BITS for (i, j) and (j, i) (tested), so only the block pairs I <= J are computed
    and mirrored.  All clouds stay resident on the GPU (4379 x 2048 x 3 floats = 108 MB).  With world > 1 the upper-triangular
    block pairs are dealt round-robin over the ranks and the caller sums the partial matrices (entries not owned are 0)."""
    pcs = torch.as_tensor(np.ascontiguousarray(point_clouds, dtype=np.float32)).to(device)
    n_all = pcs.shape[0]
    out = torch.zeros((n_all, n_all), dtype=torch.float32, device=device)
    starts = list(range(0, n_all, block))
    k = 0
    for a, i0 in enumerate(starts):
        for j0 in starts[a:]:
            if k % world == rank:
                sub = ops.chamfer_dist_matrix(pcs[i0:i0 + block], pcs[j0:j0 + block])
                out[i0:i0 + block, j0:j0 + block] = sub
                if j0 != i0:
                    out[j0:j0 + block, i0:i0 + block] = sub.T
            k += 1
    return out.cpu().numpy()


def get_chamfer_dist_mat_sharded(point_clouds, device=None, col_chunk=100):
    """Full (num_all, num_all) matrix with the column slices dealt out over the ranks (one process per GPU) and
    all-gathered -- the only collective, once, at the end."""
    rank, world = (torch.distributed.get_rank(), torch.distributed.get_world_size()) if torch.distributed.is_initialized() else (0, 1)
    n_all = len(point_clouds)
    slices = list(range(0, n_all, col_chunk))
    mine = gdist.shard_batches(len(slices), rank, world)
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    cols = [get_chamfer_dist_mat_slice(point_clouds, slices[k], col_chunk, dev) for k in mine]
    local = np.concatenate(cols, axis=1) if cols else np.zeros((n_all, 0), np.float32)
    full = gdist.all_gather_examples(torch.as_tensor(local).to(dev), axis=1)
    return full.cpu().numpy()
