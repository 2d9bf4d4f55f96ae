"""Second opinion for the unpinned half of the oracle (CPU only): oracle/attack_model.py and oracle/train_model.py
(numpy, hand-derived backward) against oracle/torch_model.py (torch library layers + autograd), fp64.

Both restate tflearn 0.3.2 / TF 1.13 semantics that cannot be executed here; agreement to 1e-10 shows the checker the HIP
kernels are measured against is not one author's single reading -- it does not pin either to the reference."""
import numpy as np
import pytest
import torch

from geometric_adv_amd import weights as W
from geometric_adv_amd.adversary import init_pert_value
from oracle.attack_model import AEModel, AttackModel
from oracle.torch_model import TorchAE, TorchAttack

N = 128


def _clouds(seed, b, n=N):
    rng = np.random.default_rng(seed)
    return (rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5)).astype(np.float32)


@pytest.fixture(scope="module")
def models():
    w = W.randomized_weights(N, seed=11)
    return w, AEModel(W.canonical(w, N), N, np.float64), TorchAE(w, N, torch.float64)


def test_forward_agrees(models):
    w, m, t = models
    x = _clouds(1, 3)
    z = m.encode(x)
    recon = m.decode(z)
    zt = t.encode(torch.as_tensor(x))
    rt = t.decode(zt)
    np.testing.assert_allclose(zt.numpy(), z, rtol=0, atol=1e-12)
    np.testing.assert_allclose(rt.numpy(), recon, rtol=0, atol=1e-12)


@pytest.mark.parametrize("adv_type,dist_type,mppw,mpdw", [("chamfer", "chamfer", 0.0, 0.0), ("latent", "chamfer", 0.0, 0.5),
                                                          ("latent", "pert", 0.3, 0.0), ("chamfer", "pert", 0.0, 0.0)])
def test_gradient_agrees_with_autograd(models, adv_type, dist_type, mppw, mpdw):
    w, m, t = models
    b = 2
    x, gt = _clouds(2, b), _clouds(3, b)
    tz = m.encode(gt)
    wts = np.array([1.0, 150.0])
    p0 = (1e-3 * np.random.default_rng(5).standard_normal((b, N, 3))).astype(np.float32)
    am = AttackModel(m, x, gt, tz, wts, adv_type, dist_type, max_point_pert_weight=mppw, max_point_dist_weight=mpdw)
    am.init_pert(p0)
    f = am.forward()
    g = am.gradient(f)
    ta = TorchAttack(t, x, gt, tz, wts, adv_type, dist_type, max_point_pert_weight=mppw, max_point_dist_weight=mpdw)
    ta.init_pert(p0)
    gt_, ft = ta.gradient(idx=f["idx"])
    for k in ("loss_adv", "loss_dist", "loss_ae", "input_dist", "loss_pert", "loss_max"):
        np.testing.assert_allclose(ft[k].detach().numpy(), f[k], rtol=1e-12, atol=1e-14, err_msg=k)
    np.testing.assert_allclose(gt_.numpy(), g, rtol=0, atol=1e-10 * max(1.0, np.abs(g).max()))


def test_tied_pool_maximum_splits_equally(models):
    """Duplicated points tie the symmetric max-pool: TF's _MinOrMaxGrad divides the gradient equally among the tied rows;
    the hand-written backward and torch.amax's autograd must both do exactly that."""
    w, m, t = models
    b = 2
    x, gt = _clouds(7, b), _clouds(8, b)
    x[:, N // 2:] = x[:, :N // 2]                      # every point twice: every channel's maximum is tied
    tz = m.encode(gt)
    p0 = np.zeros((b, N, 3), np.float32)                # identical rows stay identical
    am = AttackModel(m, x, gt, tz, np.ones(b), "latent", "pert")
    am.init_pert(p0 + 1e-4)
    f = am.forward()
    g = am.gradient(f)
    ta = TorchAttack(t, x, gt, tz, np.ones(b), "latent", "pert")
    ta.init_pert(p0 + 1e-4)
    g2, _ = ta.gradient(idx=f["idx"])
    np.testing.assert_allclose(g2.numpy(), g, rtol=0, atol=1e-10 * np.abs(g).max())
    np.testing.assert_allclose(g[:, :N // 2], g[:, N // 2:], rtol=0, atol=1e-15)      # the split is equal


def test_adam_trajectory_agrees(models):
    """Five full iterations (matches recomputed by the pinned C Chamfer in both): the hand-coded ApplyAdam of both models."""
    w, m, t = models
    b = 2
    x, gt = _clouds(12, b), _clouds(13, b)
    am = AttackModel(m, x, gt, None, np.ones(b))
    ta = TorchAttack(t, x, gt, None, np.ones(b))
    p0 = init_pert_value(b, N)
    am.init_pert(p0); ta.init_pert(p0)
    for _ in range(5):
        am.step(); ta.step()
    assert np.array_equal(ta.pert.numpy(), am.pert)      # fp32-rounded state: equal unless an update differs by > 1/2 ulp
    np.testing.assert_allclose(ta.m.numpy(), am.m, rtol=1e-6)
    np.testing.assert_allclose(ta.v.numpy(), am.v, rtol=1e-6)


def test_training_step_agrees():
    """oracle/train_model.py (BN with batch statistics, differentiated through) against F.batch_norm(training=True) + autograd."""
    import torch.nn.functional as F
    from oracle.train_model import TrainModel
    n, b = 64, 3
    w = W.randomized_weights(n, seed=21)
    canon = W.canonical(w, n)
    tm = TrainModel(canon, n)
    x = _clouds(31, b, n)
    loss, G, c = tm.loss_and_grads(x)
    i1, i2 = [torch.as_tensor(a, dtype=torch.int64) for a in tm_idx(tm, c, x)]
    P = {k: [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in canon[k]] for k in ("enc_w", "enc_b", "gamma", "beta", "dec_w", "dec_b")}
    h = torch.as_tensor(x, dtype=torch.float64).transpose(1, 2)
    for i in range(5):
        h = F.conv1d(h, P["enc_w"][i].t()[:, :, None], P["enc_b"][i])
        h = F.batch_norm(h, None, None, P["gamma"][i], P["beta"][i], training=True, eps=1e-5)
        h = F.relu(h)
    z = torch.amax(h, dim=2)
    d = F.relu(F.linear(z, P["dec_w"][0].t(), P["dec_b"][0]))
    d = F.relu(F.linear(d, P["dec_w"][1].t(), P["dec_b"][1]))
    recon = F.linear(d, P["dec_w"][2].t(), P["dec_b"][2]).reshape(b, n, 3)
    gt = torch.as_tensor(x, dtype=torch.float64)
    ga = lambda cl, idx: torch.gather(cl, 1, idx[:, :, None].expand(-1, -1, 3))
    lt = ((recon - ga(gt, i1)) ** 2).sum(-1).mean() + ((gt - ga(recon, i2)) ** 2).sum(-1).mean()
    lt.backward()
    assert abs(float(lt.detach()) - loss) < 1e-12
    np.testing.assert_allclose(recon.detach().numpy(), c["recon"], atol=1e-12)
    for k in P:
        for j, p in enumerate(P[k]):
            sc = max(1e-30, np.abs(G[k][j]).max())
            if k == "enc_b":          # exactly zero behind a batch norm in exact arithmetic: both are rounding noise
                assert np.abs(p.grad.numpy()).max() < 1e-12 and np.abs(G[k][j]).max() < 1e-12
                continue
            np.testing.assert_allclose(p.grad.numpy() / sc, G[k][j] / sc, atol=1e-9, err_msg="%s[%d]" % (k, j))


def tm_idx(tm, c, x):
    from oracle.attack_model import _o
    _, i1, _, i2 = _o().nn_distance(c["recon"].astype(np.float32), np.asarray(x, np.float32))
    return i1, i2
