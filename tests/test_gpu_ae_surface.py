"""GPU: the AutoEncoder class surface the defender scripts use beyond reconstruct / transform (src/autoencoder.py:140-148,
178-194, 296-307; src/pointnet_ae.py:140-143), against the fp64 model (oracle/attack_model.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ae(n, seed=21):
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.autoencoder import PointNetAE
    from oracle.attack_model import AEModel
    w = W.randomized_weights(n, seed=seed)
    return PointNetAE(w, n), AEModel(W.canonical(w, n), n, np.float64)


def test_decode_is_the_decoder_half_of_the_forward():
    """decode(transform(X)) == reconstruct(X) bit for bit (same kernels, same summation order); against the model 2e-6; a
    single code (128,) is accepted like the reference's np.expand_dims branch; negative code entries are decoded, not clipped."""
    from conftest import cloud
    n = 1024
    ae, model = _ae(n)
    x = cloud(3, 7, n)
    recon, _ = ae.reconstruct(x, compute_loss=False)
    z = ae.transform(x)
    assert np.array_equal(ae.decode(z), recon)
    assert np.array_equal(ae.decode(z[2]), recon[2:3])
    zz = z.copy()
    zz[:, ::3] = -1.5 - zz[:, ::3]
    np.testing.assert_allclose(ae.decode(zz), model.decode(zz.astype(np.float64)), atol=2e-6)
    with pytest.raises(ValueError):
        ae.decode(np.zeros((2, 64), np.float32))


def test_interpolate_get_reconstructions_get_loss():
    from conftest import cloud
    n = 512
    ae, model = _ae(n, seed=22)
    x = cloud(4, 9, n)
    out = ae.interpolate(x[0], x[1], 3)
    assert out.shape == (5, n, 3)
    z = ae.transform(x[:2])
    assert np.array_equal(out[0], ae.decode(z[0])[0]) and np.array_equal(out[-1], ae.decode(z[1])[0])
    mid = (0.5 * z[1].astype(np.float64) + 0.5 * z[0].astype(np.float64)).astype(np.float32)
    assert np.array_equal(out[2], ae.decode(mid)[0])
    rec = ae.get_reconstructions(x, batch_size=4)                    # 4 + 4 + 1 clouds
    assert rec.shape == x.shape and np.array_equal(rec, ae.reconstruct(x, compute_loss=False)[0])
    per = ae.get_loss_per_pc(x)
    assert abs(ae.get_loss(x) - per.mean()) < 1e-6 * per.mean()
    gt = cloud(5, 9, n)
    assert abs(ae.get_loss(x, gt) - ae.get_loss_per_pc(x, gt).mean()) < 1e-6 * per.mean()
    assert abs(ae.reconstruct(x, gt)[1] - ae.get_loss(x, gt)) < 1e-6 * per.mean()


def test_gradient_of_input_wrt_loss():
    """tf.gradients(self.loss, self.x): the loop's own backward at zero perturbation against the fp64 model's gradient of
    mean_b [mean_j dist1 + mean_k dist2]."""
    from oracle.attack_model import AttackModel
    from conftest import cloud
    n, b = 512, 4
    ae, model = _ae(n, seed=23)
    x, gt = cloud(6, b, n), cloud(7, b, n)
    g, = ae.gradient_of_input_wrt_loss(x, gt)
    am = AttackModel(model, x, gt, None, np.zeros(b))
    am.init_pert(np.zeros((b, n, 3)))
    want = am.gradient(am.forward()) / b
    assert g.shape == want.shape
    np.testing.assert_allclose(g, want, atol=3e-5 * np.abs(want).max())
    g2, = ae.gradient_of_input_wrt_loss(x)                            # gt = x: the plain auto-encoding loss
    assert np.isfinite(g2).all() and np.abs(g2).max() > 0


def test_attack_status_is_ok_and_large_clouds_take_the_two_launch_backward():
    """geoadv_attack_status after ordinary runs; at n = 4096 (batch + 2 n / 32 workgroups no longer fit the chip at once) the
    Jacobian path uses the tail and the dense backward as two launches -- same gradient as the masked backward to rounding, also
    for a cloud with a tied pool maximum (duplicated points)."""
    import torch
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd.autoencoder import PointNetAE
    from conftest import cloud
    n, b = 4096, 8
    w = W.randomized_weights(n, seed=5)
    ae = PointNetAE(w, n)
    x, gt = cloud(11, b, n), cloud(12, b, n)
    x[1, 1000:1400] = x[1, 100:500]                                   # duplicated points: tied maxima in cloud 1
    grads = {}
    for form in ("jacobian", "masked"):
        at = AdvAE("a", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=3, num_iterations_thresh=9,
                                      encoder_backward=form), ae=ae)
        at.set_inputs(x, gt, None, 1.0)
        at.init_pert(None, reset_optimizer=True)
        at.run(0, 2, 9)
        at.status()
        grads[form] = at.peek()["grad"]
    scale = grads["masked"].abs().max()
    assert scale > 0 and (grads["jacobian"] - grads["masked"]).abs().max() <= 5e-6 * scale
