"""one training step of the victim auto-encoder (csrc/train.hip through the C ABI) against the numpy
fp64 model oracle/train_model.py (PointNetAutoEncoder._create_loss/_setup_optimizer + partial_fit, tflearn BN in
training mode).  Tolerances: loss 1e-5 relative (the north star's Chamfer tolerance), gradients 5e-5 of each
variable's norm (fp32 sums over B*N rows against fp64; measured ~1e-6), weights after one Adam step where the gradient is not
rounding noise."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])

torch = pytest.importorskip("torch")


def _clouds(seed, b, n):
    rng = np.random.default_rng(seed)
    return (rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5)).astype(np.float32)


def _setup(n, b, seed=11, lr=0.0005, loss="chamfer", max_workgroups=0):
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.trainer import PointNetAETrainer
    from oracle.train_model import TrainModel
    w = W.randomized_weights(n, seed=seed)
    tr = PointNetAETrainer(w, n, batch_size=b, learning_rate=lr, loss=loss, max_workgroups=max_workgroups)
    tm = TrainModel(W.canonical(w, n), n, lr=lr, loss=loss)
    return w, tr, tm


def _well_conditioned(tm, x, margin=5e-6):
    """A ReLU input within rounding of 0 (or a near-tie in the max-pool) makes the mask a coin flip between fp32 and
    fp64 -- one flipped row shifts a gradient by ~1/rows.  With ~1e6 activations per small batch that happens for
    about one seed in three, so the parity batches are drawn until the fp64 model has no such element."""
    c = tm.forward(x)
    for i in range(5):
        if np.abs(c["xhat"][i] * tm.p["gamma"][i] + tm.p["beta"][i]).min() < margin:
            return False
    for d in (c["d1"], c["d2"]):
        if np.abs(d[d > 0]).min() < margin:
            return False
    top = np.sort(c["h5"], axis=1)[:, -2:, :]
    gap = (top[:, 1] - top[:, 0])[c["z"] > 0]
    return gap.size == 0 or gap.min() > margin


def _batch(tm, b, n, seed, duplicate=False):
    for s in range(seed, seed + 50):
        x = _clouds(s, b, n // 2 if duplicate else n)
        if duplicate:
            x = np.concatenate([x, x], axis=1)
            c = tm.forward(x)                                   # ties are the point here; only the ReLUs must be clear
            if all(np.abs(c["xhat"][i] * tm.p["gamma"][i] + tm.p["beta"][i]).min() >= 5e-6 for i in range(5)):
                return x
        elif _well_conditioned(tm, x):
            return x
    raise AssertionError("no well-conditioned batch found")


def _rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30))


# max_workgroups: the layer kernels are persistent (tiles dealt round-robin over the workgroups; two LDS tile buffers, loads
# requested one or two tiles ahead).  At these shapes the device has more workgroups than there are tiles, so the default
# runs one tile per workgroup; 1 / 2 / 3 workgroups walk 4-24 tiles each through every stage of those pipelines (odd and
# even tile counts per workgroup, workgroups with one tile fewer than their neighbours).
@pytest.mark.parametrize("n,b,max_workgroups", [(128, 4, 0), (256, 3, 0), (64, 1, 0), (256, 3, 1), (256, 3, 2), (128, 4, 3), (256, 3, 5)])
def test_loss_recon_and_gradients_match_the_oracle(n, b, max_workgroups):
    from oracle.train_model import PARAM_GROUPS
    w, tr, tm = _setup(n, b, max_workgroups=max_workgroups)
    x = _batch(tm, b, n, 3)
    recon, loss = tr.forward_backward(x)
    g_gpu = tr.gradients()
    loss_ref, G, c = tm.loss_and_grads(x)
    assert abs(float(loss.item()) - loss_ref) <= 1e-5 * abs(loss_ref)
    assert np.abs(recon.cpu().numpy() - c["recon"]).max() <= 1e-5
    for k in PARAM_GROUPS:
        for j in range(len(G[k])):
            if k == "enc_b":          # exactly zero behind a batch norm: both sides hold rounding noise only
                scale = np.abs(G["enc_w"][j]).max()
                assert np.abs(g_gpu[k][j]).max() <= 1e-4 * scale and np.abs(G[k][j]).max() <= 1e-10 * max(scale, 1e-30)
                continue
            assert g_gpu[k][j].shape == G[k][j].shape
            assert _rel(g_gpu[k][j], G[k][j]) <= 5e-5, (k, j, _rel(g_gpu[k][j], G[k][j]))


@pytest.mark.parametrize("n,b", [(128, 4), (256, 3), (512, 2)])
def test_emd_loss_step_matches_the_oracle(n, b):
    """conf.loss == 'emd' (src/pointnet_ae.py:77-79): loss = reduce_mean(match_cost(recon, gt, approx_match(recon, gt))) and every
    variable's gradient against the fp64 model, whose EMD pieces are the pinned C restatements of the reference's CPU ops
    (approxmatch_cpu / matchcost_cpu / matchcostgrad_cpu).  The kernel's plan uses fp32 pair weights: loss 2e-5
    relative, gradients 2e-4 of each variable's norm."""
    from oracle.train_model import PARAM_GROUPS
    w, tr, tm = _setup(n, b, loss="emd")
    x = _batch(tm, b, n, 3)
    gt = _clouds(77, b, n)                                   # a denoising-style target: recon is matched against other clouds
    recon, loss = tr.forward_backward(x, gt)
    g_gpu = tr.gradients()
    loss_ref, G, c = tm.loss_and_grads(x, gt)
    assert abs(float(loss.item()) - loss_ref) <= 2e-5 * abs(loss_ref)
    for k in PARAM_GROUPS:
        for j in range(len(G[k])):
            if k == "enc_b":
                continue
            assert _rel(g_gpu[k][j], G[k][j]) <= 2e-4, (k, j, _rel(g_gpu[k][j], G[k][j]))
    # and a step moves the weights: the loss of the next forward on the same batch is lower
    first = float(tr.partial_fit(x, gt)[1])
    for _ in range(5):
        last = float(tr.partial_fit(x, gt)[1])
    assert last < first


def test_one_adam_step_and_moving_averages():
    n, b = 128, 4
    w, tr, tm = _setup(n, b)
    x = _batch(tm, b, n, 5)
    _, G, _ = tm.loss_and_grads(x)
    recon, loss = tr.partial_fit(x)
    loss_ref, recon_ref = tm.step(x)
    assert abs(loss - loss_ref) <= 1e-5 * abs(loss_ref)
    from geometric_adv_amd import weights as W
    new = W.canonical(tr.export_weights(), n)
    lr = 0.0005
    for k in ("enc_w", "gamma", "beta", "dec_w", "dec_b"):
        for j in range(len(G[k])):
            solid = np.abs(G[k][j]) > 1e-4 * np.abs(G[k][j]).max()          # Adam's first step is lr * g / (|g| + 3e-7)
            diff = np.abs(new[k][j].astype(np.float64) - tm.p[k][j])[solid]
            assert diff.size and diff.max() <= 2e-2 * lr, (k, j, diff.max())
            moved = np.abs(new[k][j].astype(np.float64) - np.asarray(W.canonical(w, n)[k][j], np.float64))[solid]
            assert moved.min() > 0.5 * lr                                   # every such variable really moved by ~lr
    for i in range(5):
        assert np.allclose(new["mean"][i], tm.p["mean"][i], rtol=1e-5, atol=1e-6)
        assert np.allclose(new["var"][i], tm.p["var"][i], rtol=1e-4, atol=1e-7)


def test_loss_trajectory_tracks_the_oracle():
    n, b = 128, 4
    w, tr, tm = _setup(n, b, lr=0.0005)
    x = _clouds(9, b, n)
    gpu, ref = [], []
    for _ in range(12):
        gpu.append(tr.partial_fit(x, want_recon=False)[1])
        ref.append(tm.step(x)[0])
    gpu, ref = np.array(gpu), np.array(ref)
    assert ref[-1] < 0.7 * ref[0]                                           # it trains
    assert np.abs(gpu[:4] - ref[:4]).max() <= 1e-4 * ref[0]
    assert np.abs(gpu - ref).max() <= 2e-2 * ref[0]                         # sign-like Adam updates amplify rounding slowly


def test_results_do_not_depend_on_the_number_of_workgroups():
    """Everything but the weight gradients (whose per-workgroup partial sums regroup) is bit-identical for any cap."""
    n, b = 256, 3
    outs = []
    for cap in (0, 1, 4):
        w, tr, tm = _setup(n, b, max_workgroups=cap)
        x = _batch(tm, b, n, 3)
        recon, loss = tr.forward_backward(x)
        g = tr.gradients()
        outs.append((recon.cpu().numpy().copy(), float(loss.item()), g))
    for r, l, g in outs[1:]:
        assert np.array_equal(r, outs[0][0]) and l == outs[0][1]
        for j in range(len(g["enc_w"])):
            assert _rel(g["enc_w"][j], np.asarray(outs[0][2]["enc_w"][j], np.float64)) <= 2e-6


def test_step_is_deterministic():
    n, b = 256, 6
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.trainer import PointNetAETrainer
    w = W.randomized_weights(n, seed=2)
    x = _clouds(1, b, n)
    outs = []
    for _ in range(2):
        tr = PointNetAETrainer(w, n, batch_size=b)
        for _ in range(3):
            tr.partial_fit(x, want_recon=False)
        tr.forward_backward(x, want_recon=False)
        torch.cuda.synchronize()
        outs.append(tr.gradient_buffer().cpu().numpy().copy())
    assert np.array_equal(outs[0], outs[1])


def test_duplicated_points_split_the_pool_gradient():
    """Exact ties in the max-pool (TF _MinOrMaxGrad equal split): every point appears twice."""
    n, b = 128, 2
    w, tr, tm = _setup(n, b)
    x = _batch(tm, b, n, 4, duplicate=True)
    tr.forward_backward(x)
    g_gpu = tr.gradients()
    _, G, _ = tm.loss_and_grads(x)
    for k in ("enc_w", "gamma", "beta"):
        for j in range(5):
            assert _rel(g_gpu[k][j], G[k][j]) <= 5e-5, (k, j)


def test_full_size_step_trains_and_exports():
    """default_train_params shape (batch 50, 2048 points, lr 0.0005): the loss of a fixed batch falls, and the exported
    variables drive the inference/attack handle (BN moving averages included)."""
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.trainer import PointNetAETrainer, initial_weights
    from geometric_adv_amd.autoencoder import PointNetAE
    n, b = 2048, 50
    tr = PointNetAETrainer(initial_weights(n, seed=1), n, batch_size=b)
    x = torch.as_tensor(_clouds(7, b, n)).cuda()
    losses = [tr.partial_fit(x, want_recon=False)[1] for _ in range(30)]
    assert np.all(np.isfinite(losses)) and losses[-1] < 0.5 * losses[0]
    exported = tr.export_weights()
    assert set(exported) == set(W.variable_names())
    ae = PointNetAE(exported, n)
    recon, latent = ae.forward(x[:4])
    assert torch.isfinite(recon).all() and torch.isfinite(latent).all()


def test_bad_shapes_are_rejected():
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.trainer import PointNetAETrainer
    with pytest.raises(ValueError):
        PointNetAETrainer(W.synthetic_weights(100), 100, batch_size=2)      # n_points must be a multiple of 64
    tr = PointNetAETrainer(W.synthetic_weights(64), 64, batch_size=2)
    with pytest.raises(ValueError):
        tr.partial_fit(np.zeros((3, 64, 3), np.float32))


_DP_WORKER = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import torch.distributed as dist
from geometric_adv_amd import weights as W
from geometric_adv_amd.trainer import PointNetAETrainer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)       # both ranks share the one GPU of the test box
n, b = 128, 4
x = np.load(sys.argv[1])
tr = PointNetAETrainer(W.randomized_weights(n, seed=11), n, batch_size=b // world, sync_bn=False)
losses = [tr.partial_fit(x[rank * (b // world):(rank + 1) * (b // world)], want_recon=False)[1] for _ in range(3)]
torch.cuda.synchronize()
if rank == 0:
    np.savez(sys.argv[2], losses=np.array(losses), params=tr.parameter_buffer().cpu().numpy())
dist.destroy_process_group()
"""


def test_data_parallel_step_equals_summed_shard_gradients(tmp_path):
    """Two ranks (gloo group, one GPU), each with half of the batch: all-reduced gradients / world size must give
    exactly the update of two replicas whose gradient buffers are added by hand."""
    import os, subprocess, sys
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.trainer import PointNetAETrainer
    n, b = 128, 4
    x = _clouds(21, b, n)
    np.save(tmp_path / "x.npy", x)
    (tmp_path / "worker.py").write_text(_DP_WORKER)
    _PORT1 = _free_port()                              # (a fixed port collides with a concurrent run of this suite)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_PORT1, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _PORT1, str(tmp_path / "worker.py"), str(tmp_path / "x.npy"), str(tmp_path / "out.npz")]
    subprocess.run(cmd, check=True, env=env, timeout=300, cwd=os.getcwd())
    got = np.load(tmp_path / "out.npz")
    # the same thing in one process: two replicas, gradients added by hand
    w = W.randomized_weights(n, seed=11)
    reps = [PointNetAETrainer(w, n, batch_size=b // 2) for _ in range(2)]
    losses = []
    for _ in range(3):
        ls = [float(r.forward_backward(x[i * 2:(i + 1) * 2], want_recon=False)[1].item()) for i, r in enumerate(reps)]
        total = reps[0].gradient_buffer() + reps[1].gradient_buffer()
        for r in reps:
            r.gradient_buffer().copy_(total)
            r.apply(0.5)
        losses.append(0.5 * (ls[0] + ls[1]))
    torch.cuda.synchronize()
    assert np.allclose(got["losses"], losses, rtol=1e-6)
    assert np.array_equal(got["params"], reps[0].parameter_buffer().cpu().numpy())
    assert np.array_equal(reps[0].parameter_buffer().cpu().numpy(), reps[1].parameter_buffer().cpu().numpy())


_SYNC_WORKER = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import torch.distributed as dist
from geometric_adv_amd import weights as W
from geometric_adv_amd.trainer import PointNetAETrainer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n, b = 128, 4
x = np.load(sys.argv[1])
tr = PointNetAETrainer(W.randomized_weights(n, seed=11), n, batch_size=b // world)      # sync_bn=True
shard = x[rank * (b // world):(rank + 1) * (b // world)]
tr.apply = lambda scale=1.0: None                     # keep the all-reduced gradients of the first step for inspection
_, loss1 = tr.partial_fit(shard, want_recon=False)
grads = tr.gradient_buffer().cpu().numpy().copy()
del tr.apply
tr2 = PointNetAETrainer(W.randomized_weights(n, seed=11), n, batch_size=b // world)
losses = [tr2.partial_fit(shard, want_recon=False)[1] for _ in range(4)]
w = tr2.export_weights()
torch.cuda.synchronize()
if rank == 0:
    np.savez(sys.argv[2], loss1=loss1, grads=grads, losses=np.array(losses),
             mean0=w["autoencoder/encoder_conv_layer_0_bnorm/moving_mean"], var3=w["autoencoder/encoder_conv_layer_3_bnorm/moving_variance"])
dist.destroy_process_group()
"""


def test_synchronised_bn_equals_one_replica_on_the_global_batch(tmp_path):
    """Two ranks x 2 clouds with the batch-norm sums all-reduced between the phases == one replica on all 4 clouds
    (the reference's step): loss, every gradient, the loss trajectory and the moving averages."""
    import subprocess, sys
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.trainer import PointNetAETrainer
    n, b = 128, 4
    w0 = W.randomized_weights(n, seed=11)
    from oracle.train_model import TrainModel
    x = _batch(TrainModel(W.canonical(w0, n), n), b, n, 31)
    np.save(tmp_path / "x.npy", x)
    (tmp_path / "worker.py").write_text(_SYNC_WORKER)
    _PORT2 = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_PORT2, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _PORT2, str(tmp_path / "worker.py"), str(tmp_path / "x.npy"), str(tmp_path / "out.npz")]
    subprocess.run(cmd, check=True, env=env, timeout=300, cwd=os.getcwd())
    got = np.load(tmp_path / "out.npz")
    one = PointNetAETrainer(w0, n, batch_size=b)
    _, loss = one.forward_backward(x, want_recon=False)
    assert abs(float(got["loss1"]) - float(loss.item())) <= 1e-6 * float(loss.item())
    ref = one._unflatten(one.gradient_buffer().cpu().numpy())
    dp = one._unflatten(got["grads"])
    for k in ref:
        for j in range(len(ref[k])):
            if k == "enc_b":
                continue                                   # rounding noise on both sides (zero behind a batch norm)
            assert _rel(dp[k][j], ref[k][j].astype(np.float64)) <= 2e-5, (k, j, _rel(dp[k][j], ref[k][j].astype(np.float64)))
    one2 = PointNetAETrainer(w0, n, batch_size=b)
    losses = [one2.partial_fit(x, want_recon=False)[1] for _ in range(4)]
    assert np.allclose(got["losses"], losses, rtol=2e-4)
    w = one2.export_weights()
    # the conv bias in front of a batch norm has a zero gradient: Adam turns its rounding noise into +-lr steps, so the
    # bias -- and with it the moving MEAN -- random-walks by up to 4 * lr on either side; the variance is unaffected
    assert np.allclose(got["mean0"], w["autoencoder/encoder_conv_layer_0_bnorm/moving_mean"], rtol=0, atol=4 * 0.0005 + 1e-5)
    assert np.allclose(got["var3"], w["autoencoder/encoder_conv_layer_3_bnorm/moving_variance"], rtol=1e-3, atol=1e-7)


def test_train_ae_cli_writes_a_restorable_checkpoint(tmp_path):
    from geometric_adv_amd import train_ae, tf_checkpoint, weights as W
    from geometric_adv_amd.autoencoder import PointNetAE
    data = _clouds(33, 12, 128)
    np.save(tmp_path / "clouds.npy", data)
    stats = train_ae.main(["--train_data", str(tmp_path / "clouds.npy"), "--train_folder", str(tmp_path / "ae"),
                           "--training_epochs", "3", "--batch_size", "4", "--saver_step", "2"])
    assert len(stats) == 3 and stats[-1][1] < stats[0][1]
    for epoch in (1, 2, 3):
        assert os.path.exists(tmp_path / "ae" / ("models.ckpt-%d.index" % epoch))
    lines = open(tmp_path / "ae" / "train_stats.txt").read().strip().splitlines()
    assert len(lines) == 3 and lines[0].startswith("0001\t")
    w = tf_checkpoint.restore_ae_weights(str(tmp_path / "ae"), 3)          # restore_ae_model's path, TF-free
    recon, latent = PointNetAE(w, 128).forward(data[:2])
    assert torch.isfinite(recon).all()
