"""CPU, world_size 2 (gloo): the N > 1 path -- batch sharding and the result gather -- with a
stand-in for the per-rank GPU attack (the test double marks every example with its global index,
so the gathered order can be checked)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


class _FakeConf:
    def __init__(self, batch_size, n, weights_list):
        self.batch_size = batch_size
        self.n_input = [n, 3]
        self.dist_weight_list = weights_list


class _FakeAdvAE:
    """Test double: 'attacks' by copying; metrics[:, i, 0] = first coordinate of example i."""
    device = None

    def __init__(self, conf):
        self.configuration = conf
        self.calls = 0

    def attack(self, source_pc, target_latent, target_pc, ref, conf):
        self.calls += 1
        W, k, n = len(conf.dist_weight_list), len(source_pc), conf.n_input[0]
        assert k % conf.batch_size == 0
        m = np.zeros((W, k, 5), np.float32)
        m[:, :, 0] = source_pc[:, 0, 0][None]
        m[:, :, 4] = ref[None]
        adv = np.broadcast_to(source_pc[None], (W, k, n, 3)).copy()
        return m, adv, adv.copy()


def _worker(rank, world, port, n_examples, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from geometric_adv_amd import dist as gdist
    r, w, _ = gdist.init("gloo")
    assert (r, w) == (rank, world)
    n, bsz = 8, 2
    src = np.zeros((n_examples, n, 3), np.float32)
    src[:, 0, 0] = np.arange(n_examples)
    ref = np.arange(n_examples, dtype=np.float32) * 10
    fake = _FakeAdvAE(_FakeConf(bsz, n, [0.5, 2.0]))
    metrics, adv, rec, sl = gdist.attack_sharded(fake, src, None, src, ref, gather_clouds=True, device="cpu")
    t = gdist.max_over_ranks(float(rank + 1))
    gdist.barrier()
    q.put((rank, metrics, adv, (sl.start, sl.stop), t, fake.calls))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_examples", [8, 6, 2])
def test_sharded_attack_world2(n_examples):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_examples, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    slices = [o[3] for o in outs]
    # contiguous, disjoint, covering; whole batches only
    assert slices[0][0] == 0 and slices[0][1] == slices[1][0] and slices[1][1] == n_examples
    assert all((b - a) % 2 == 0 for a, b in slices)
    for rank, metrics, adv, sl, t, calls in outs:
        assert metrics.shape == (2, n_examples, 5)
        assert np.array_equal(metrics[0, :, 0], np.arange(n_examples))            # rank order == example order
        assert np.array_equal(metrics[1, :, 4], np.arange(n_examples) * 10)
        assert adv.shape == (2, n_examples, 8, 3) and np.array_equal(adv[0, :, 0, 0], np.arange(n_examples))
        assert t == 2.0                                                            # MAX over ranks
        assert calls == (1 if sl[1] > sl[0] else 0)


def test_shard_batches_partition():
    from geometric_adv_amd.dist import shard_batches
    for n in range(0, 20):
        for world in (1, 2, 3, 8):
            parts = [shard_batches(n, r, world) for r in range(world)]
            flat = [i for p in parts for i in p]
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
