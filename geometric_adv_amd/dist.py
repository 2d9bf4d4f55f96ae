"""Multi-GPU: one process per GPU, batches sharded across ranks, RCCL only for the result gather.

The attack has no cross-example coupling (the loss is a sum over the batch, adv_ae.py:105; pert
and the Adam slots are per example; weights are frozen), and AdvAE.attack already walks the
examples in independent batches of batch_size (adv_ae.py:166-177).  So ranks take disjoint
contiguous runs of BATCHES, run them with no communication at all, and all-gather only the final
per-cloud metric scalars (5 floats per cloud and dist weight) -- KB-sized, latency-bound, the xGMI
link bandwidth is irrelevant.  Clouds are gathered to rank 0 only on request.

Backend: 'nccl' (= RCCL on ROCm) on GPUs, 'gloo' for the CPU tests of this module.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def env_rank():
    """(rank, world_size, local_rank) from the torchrun environment (1-process defaults)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


_EXERCISE_ONE_RANK = False      # init(single_rank_group=True): the caller WANTS a one-rank group's collectives to run through the backend


def _active():
    """True where a collective has something to do: several ranks, or a one-rank group brought up to exercise the backend.  A
    one-rank group somebody else initialised (world == 1) keeps the no-op fast path."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or _EXERCISE_ONE_RANK)


def init(backend=None, single_rank_group=False, timeout_s=None):
    """Initialise torch.distributed from the environment if WORLD_SIZE > 1.  Returns (rank, world, local_rank).

    single_rank_group: also create a group when WORLD_SIZE is 1 (file-store rendezvous in a fresh temporary directory) -- the
    collectives below then really run through the backend (RCCL accepts ONE rank per device, so this is how a 1-GPU box
    exercises librccl: tests/test_gpu_dist_rccl.py, bench.py at --gpus 1).  timeout_s: collective / rendezvous timeout."""
    global _EXERCISE_ONE_RANK
    rank, world, local = env_rank()
    if single_rank_group and world == 1:
        _EXERCISE_ONE_RANK = True
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        kw = {}
        if timeout_s is not None:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
        if world == 1 and "MASTER_PORT" not in os.environ:
            # a one-rank group needs no network rendezvous: a file store in a fresh temporary directory (no port to pick, so no
            # race with another process taking it between the pick and the bind)
            import tempfile
            import atexit
            import shutil
            store_dir = tempfile.mkdtemp(prefix="geoadv_pg_")
            kw["init_method"] = "file://" + os.path.join(store_dir, "store")
            # removed when the process ends, not right after init: RCCL creates its communicator lazily and passes the unique
            # id through this store at the first collective
            atexit.register(shutil.rmtree, store_dir, ignore_errors=True)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def backend_info():
    """What the collectives of this process run on: {"backend", "world", "rccl_version"} (rccl_version only for 'nccl' = RCCL
    on ROCm; the world size is the one the process group reports, i.e. after the communicator came up)."""
    if not dist.is_initialized():
        return {"backend": "none", "world": 1, "rccl_version": None}
    b = dist.get_backend()
    ver = None
    if b == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            ver = "unknown"
    return {"backend": "rccl" if b == "nccl" else b, "world": dist.get_world_size(), "rccl_version": ver}


def shard_batches(n_batches, rank, world):
    """Contiguous split of range(n_batches); the first n_batches % world ranks get one extra."""
    q, r = divmod(n_batches, world)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def barrier():
    if _active():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    if not _active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_reduce_sum_(t):
    """In-place sum over the ranks of a (GPU) tensor: RCCL directly; a gloo group (CPU tests, or several ranks sharing
    one GPU) stages through host memory."""
    if not _active():
        return t
    if dist.get_backend() == "gloo" and t.is_cuda:
        h = t.detach().cpu()
        dist.all_reduce(h)
        t.copy_(h)
    else:
        dist.all_reduce(t)
    return t


def all_gather_examples(local, counts=None, axis=1):
    """all-gather a tensor whose `axis` is the example axis, concatenated in rank order.
    Ranks may hold different numbers of examples (pads to the maximum, trims after)."""
    if not _active():
        return local
    world = dist.get_world_size()
    n_local = torch.tensor([local.shape[axis]], dtype=torch.int64, device=local.device)
    ns = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(ns, n_local)
    ns = [int(t.item()) for t in ns]
    cap = max(ns)
    loc = local.movedim(axis, 0).contiguous()
    if loc.shape[0] < cap:
        pad = torch.zeros((cap - loc.shape[0],) + tuple(loc.shape[1:]), dtype=loc.dtype, device=loc.device)
        loc = torch.cat([loc, pad], 0)
    parts = [torch.empty_like(loc) for _ in range(world)]
    dist.all_gather(parts, loc)
    out = torch.cat([p[:k] for p, k in zip(parts, ns)], 0)
    return out.movedim(0, axis).contiguous()


def attack_sharded(adv_ae, source_pc, target_latent, target_pc, target_ae_loss_ref, gather_clouds=False, device=None,
                   log_file=None):
    """AdvAE.attack over all examples with the batches sharded across ranks.

    Every rank passes the FULL arrays (n_examples must divide by batch_size, adv_ae.py:162).
    Returns a 4-tuple (metrics [W, n_examples, 5] on every rank, adv, recon, local_slice): adv / recon cover all examples
    when gather_clouds, else only this rank's, whose position in the example order is `local_slice`.
    log_file: this rank's log (the per-iteration and per-batch lines of adv_ae.py:223-232,181-184 for ITS batches).

    Adam state and sharding: the reference initialises Adam's slots and beta powers once per graph and never resets them
    (adv_ae.py:74, 152-153), so in a one-process run batch k starts from the slots batch k-1 left behind.  A rank here is one
    such graph for ITS run of batches: an N-rank result equals, bit for bit, N single-process attacks on the N shards (and
    `Configuration.batch_slots = N` on one GPU) -- not a one-process attack over all batches, whose later batches inherit
    other slots.  Every batch's optimum is the same problem either way; only the trajectory differs.
    "Bit for bit" holds at equal Configuration: every kernel sums a cloud's numbers in an order fixed by the cloud alone
    (tests/test_gpu_attack.py::test_trajectory_of_a_cloud_does_not_depend_on_its_batch), but encoder_backward="auto"
    picks the masked backward while batch * n_points < 4096 (GEOADV_SYM_MIN_POINTS, include/geoadv.h: one cloud of 2048
    points) and the pool Jacobian from there on, and those two forms
    agree to rounding only: pin encoder_backward when shards of different batch size must reproduce each other exactly."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    c = adv_ae.configuration
    n_examples = len(source_pc)
    assert n_examples % c.batch_size == 0, \
        'The number of examples (%d) should be divided by the batch size (%d)' % (n_examples, c.batch_size)
    mine = shard_batches(n_examples // c.batch_size, rank, world)
    lo, hi = mine.start * c.batch_size, mine.stop * c.batch_size
    W = len(c.dist_weight_list)
    if hi > lo:
        args = (source_pc[lo:hi], None if target_latent is None else target_latent[lo:hi], target_pc[lo:hi],
                target_ae_loss_ref[lo:hi], c)
        m_, a_, r_ = adv_ae.attack(*args, log_file=log_file) if log_file is not None else adv_ae.attack(*args)
    else:
        n = c.n_input[0]
        m_, a_, r_ = np.zeros((W, 0, 5), np.float32), np.zeros((W, 0, n, 3), np.float32), np.zeros((W, 0, n, 3), np.float32)
    dev = device if device is not None else (adv_ae.device if getattr(adv_ae, "device", None) is not None else "cpu")
    metrics = all_gather_examples(torch.as_tensor(m_).to(dev)).cpu().numpy()
    if gather_clouds:
        a_ = all_gather_examples(torch.as_tensor(a_).to(dev)).cpu().numpy()
        r_ = all_gather_examples(torch.as_tensor(r_).to(dev)).cpu().numpy()
    return metrics, a_, r_, slice(lo, hi)
