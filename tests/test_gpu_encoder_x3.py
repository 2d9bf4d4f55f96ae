"""GPU: the encoder's three arithmetics (include/geoadv.h GEOADV_ENC_ARITH_*).

f16x2 (the default) forms every fp32 product as three fp16 piece products of power-of-two-scaled operands on the fp16 matrix
pipe, bf16x3 as six bf16 piece products; f32 is the fp32 MFMA.  All are checked against the float64 model of the reference
encoder (oracle/attack_model.py, src/encoders_decoders.py:37-72) at the same tolerance, against each other, and each against
itself across batch sizes / kernel forms (bit for bit); f16x2's range guard is driven to trip."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ARITHS = ("f16x2", "bf16x3", "f32")


def _models(n, seed=3):
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.autoencoder import PointNetAE
    w = W.randomized_weights(n, seed=seed)
    return w, {a: PointNetAE(w, n, encoder_arith=a) for a in ARITHS}


@pytest.mark.parametrize("n,b", [(2048, 3), (2048, 40), (1000, 5), (100, 2), (33, 4), (1, 2)])
def test_latent_of_both_arithmetics_against_the_float64_model(n, b):
    from geometric_adv_amd import weights as W
    from oracle.attack_model import AEModel
    from conftest import cloud
    w, aes = _models(n)
    pc = cloud(11 + n, b, n)
    ref = AEModel(W.canonical(w, n), n, np.float64)
    z64 = ref.encode(pc.astype(np.float64))
    got = {}
    for a, ae in aes.items():
        assert ae.encoder_arith == a
        got[a] = ae.forward(pc, want_recon=False)[1].cpu().numpy()
    scale = np.abs(z64).max()
    for a in ARITHS:
        np.testing.assert_allclose(got[a] / scale, z64 / scale, atol=2e-6, err_msg=a)
    np.testing.assert_allclose(got["bf16x3"], got["f32"], atol=2e-6 * np.abs(got["f32"]).max())
    np.testing.assert_allclose(got["f16x2"], got["f32"], atol=2e-6 * np.abs(got["f32"]).max())


@pytest.mark.parametrize("arith", ARITHS)
def test_a_clouds_latent_and_critical_points_do_not_depend_on_the_batch(arith):
    """The forward picks its workgroup shape by batch size (bf16x3: four waves sharing a 32-point unit for few clouds, a wave per
    32 points otherwise; f32: 32- or 64-row tiles): the same chain per output element in every form, so the same bits."""
    import torch
    from conftest import cloud
    for n in (2048, 777):
        w, aes = _models(n, seed=5)
        ae = aes[arith]
        pc = cloud(21, 48, n)
        z_all, i_all = ae.max_and_argmax(pc)
        for b in (1, 2, 7, 16, 17, 33):
            z, i = ae.max_and_argmax(pc[:b])
            assert torch.equal(torch.as_tensor(z), torch.as_tensor(z_all)[:b]), (n, b)
            assert torch.equal(torch.as_tensor(i), torch.as_tensor(i_all)[:b]), (n, b)


@pytest.mark.parametrize("arith", ARITHS)
def test_attack_trajectory_against_the_float64_model_under_each_arithmetic(arith):
    """Ten iterations of the output-space attack: the per-iteration losses against the float64 model driven with the GPU's own
    perturbation (1e-5, the path's tolerance), under either arithmetic; small batch (split form) and large."""
    import torch
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from oracle.attack_model import AEModel, AttackModel
    from conftest import cloud
    n = 1024
    w = W.synthetic_weights(n, seed=2)
    for b in (3, 40):
        x, gt = cloud(41, b, n), cloud(42, b, n)
        at = AdvAE("adversary", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=10, num_iterations_thresh=5,
                                              encoder_arith=arith))
        assert at.ae.encoder_arith == arith
        at.set_inputs(x, gt, None, 1.0)
        at.init_pert(None, reset_optimizer=True)
        hist = torch.empty((10, 6, b), device=at.device)
        at.run(0, 10, 1, hist)
        s = {k: v.cpu().numpy() for k, v in at.peek().items()}
        sel = [0, b - 1]
        am = AttackModel(AEModel(W.canonical(w, n), n, np.float64), x[sel], gt[sel], None, np.ones(len(sel)))
        am.pert = s["pert"][sel].astype(np.float64)
        f = am.forward()
        np.testing.assert_allclose(s["latent"][sel], f["z"], atol=2e-6)
        np.testing.assert_allclose(s["recon"][sel], f["recon"], atol=2e-6)


@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("backward", ["auto", "masked", "jacobian"])
def test_attack_state_agrees_between_a_two_cloud_and_a_forty_cloud_batch(arith, backward):
    """Three iterations on the same clouds (same initial perturbation) in a batch of 2 (bf16x3: the split form) and of 40: every
    piece of state bit for bit -- the ReLU masks, tie counts and arg-max rows feed the backward, so they are covered through it."""
    import torch
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from conftest import cloud
    n = 2048
    w = W.synthetic_weights(n, seed=4)
    x, gt = cloud(51, 40, n), cloud(52, 40, n)
    p0 = (1e-3 * np.random.default_rng(3).standard_normal((40, n, 3))).astype(np.float32)      # (init_pert(None) draws by shape)
    peeks = {}
    for b in (2, 40):
        at = AdvAE("adversary", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=3, num_iterations_thresh=1,
                                              encoder_arith=arith, encoder_backward=backward))
        at.set_inputs(x[:b], gt[:b], None, 1.0)
        at.init_pert(p0[:b], reset_optimizer=True)
        at.run(0, 3, 1, torch.empty((3, 6, b), device=at.device))
        peeks[b] = {k: v.cpu().numpy() for k, v in at.peek().items()}
    for k in ("latent", "recon", "adv", "pert", "grad", "idx_r1", "idx_r2", "idx_a1", "idx_a2"):
        assert np.array_equal(peeks[2][k], peeks[40][k][:2]), k


def test_arithmetic_selection_api():
    """geoadv_ae_set_encoder_arith / geoadv_ae_encoder_arith / geoadv_set_default_encoder_arith: round trip, refusal of unknown
    values, and a switched model reproducing the other arithmetic's model bit for bit."""
    import torch
    from geometric_adv_amd import _lib
    from conftest import cloud
    n = 512
    w, aes = _models(n, seed=9)
    pc = cloud(61, 4, n)
    z = {a: aes[a].forward(pc, want_recon=False)[1].clone() for a in ARITHS}
    a = aes["f32"]
    a.set_encoder_arith("bf16x3")
    assert a.encoder_arith == "bf16x3" and torch.equal(a.forward(pc, want_recon=False)[1], z["bf16x3"])
    a.set_encoder_arith("f16x2")
    assert a.encoder_arith == "f16x2" and torch.equal(a.forward(pc, want_recon=False)[1], z["f16x2"])
    a.set_encoder_arith("f32")
    assert a.encoder_arith == "f32" and torch.equal(a.forward(pc, want_recon=False)[1], z["f32"])
    lib = _lib.lib()
    from geometric_adv_amd.autoencoder import PointNetAE
    assert PointNetAE(w, n).encoder_arith == "f16x2"                   # the library default (AUTO) on a sane model
    assert lib.geoadv_ae_set_encoder_arith(a.handle, 7) != 0 and a.encoder_arith == "f32"
    assert lib.geoadv_set_default_encoder_arith(7) != 0
    with pytest.raises(KeyError):
        a.set_encoder_arith("fp16")
    # a live attack handle caches a forward in the model's arithmetic: a switch under it is refused, the same value accepted
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    at = AdvAE("a", Configuration(batch_size=2, n_points=n, weights=w, num_iterations=2, num_iterations_thresh=1), ae=a)
    assert lib.geoadv_ae_set_encoder_arith(a.handle, 1) != 0 and a.encoder_arith == "f32"
    assert lib.geoadv_ae_set_encoder_arith(a.handle, 0) == 0
    del at
    import gc
    gc.collect()
    a.set_encoder_arith("bf16x3")
    assert a.encoder_arith == "bf16x3"


def _f16x2_limits(canon):
    """Largest activation each of layers 0..3 may hand on under f16x2: 65504 / s_j with s_j = 2^6 / (the power of two nearest to the
    rms of sqrt(gamma^2 + beta^2) over the layer's channels) -- the rule of geoadv_ae_create (csrc/ae.hip), restated."""
    lim = []
    for g, be in zip(canon["gamma"][:4], canon["beta"][:4]):
        t = np.sqrt(np.mean(np.asarray(g, np.float64) ** 2 + np.asarray(be, np.float64) ** 2))
        lim.append(65504.0 / 2.0 ** (6 - int(np.round(np.log2(t)))))
    return lim


def _input_scale_for(model, limits, pc, target):
    """The factor on the input coordinates at which the largest activation / limit ratio over layers 0..3 equals `target` (bisection on
    the float64 model; far from the unit cube the activations grow with the factor)."""
    def ratio(f):
        hs = model.encode(pc.astype(np.float64) * f, keep=True)[1]
        return max(float(h.max()) / l for h, l in zip(hs[:4], limits))
    lo, hi = 1.0, 1e7
    assert ratio(lo) < target < ratio(hi)
    for _ in range(40):
        mid = np.sqrt(lo * hi)
        lo, hi = (mid, hi) if ratio(mid) < target else (lo, mid)
    return float(lo)


@pytest.mark.parametrize("b", [2, 40])
def test_f16x2_range_guard_trips_loudly_and_only_when_it_must(b):
    """Clouds far outside the unit cube the victim's batch norms were made for: while every layer's largest activation stays under
    its limit (1023.5 x the magnitude the layer's gamma / beta announce) f16x2 still agrees with the float64 model; beyond it the
    forward gives +inf latents for the clouds concerned (never a plausible number), geoadv_ae_status reports GEOADV_ERANGE once
    (the flag is cleared), transform() / the attack's status / the defenses raise, and bf16x3 runs the same clouds."""
    import torch
    from geometric_adv_amd import _lib, weights as W
    from geometric_adv_amd.autoencoder import PointNetAE
    from oracle.attack_model import AEModel
    from conftest import cloud
    n = 2048
    w = W.randomized_weights(n, seed=9)
    canon = W.canonical(w, n)
    model = AEModel(canon, n, np.float64)
    limits = _f16x2_limits(canon)
    one = cloud(71, 1, n)                                               # every cloud of the batch = the same points in another order:
    rng = np.random.default_rng(5)                                      # the same activations, so one bisection on one cloud serves all
    pc = np.concatenate([one[:, rng.permutation(n)] for _ in range(b)])
    lib = _lib.lib()
    ae = PointNetAE(w, n, encoder_arith="f16x2")
    # just inside the range
    pc_in = (pc * np.float32(_input_scale_for(model, limits, one, 0.85))).astype(np.float32)
    z = ae.forward(pc_in, want_recon=False)[1].cpu().numpy()
    assert lib.geoadv_ae_status(ae.handle, _lib.stream_handle()) == 0
    z64 = model.encode(pc_in.astype(np.float64))
    np.testing.assert_allclose(z / np.abs(z64).max(), z64 / np.abs(z64).max(), atol=2e-6)
    # beyond it
    pc_out = (pc * np.float32(_input_scale_for(model, limits, one, 1.1))).astype(np.float32)
    z = ae.forward(pc_out, want_recon=False)[1].cpu().numpy()
    assert np.isinf(z).any() and not np.isnan(z).any()
    assert lib.geoadv_ae_status(ae.handle, _lib.stream_handle()) == 4          # GEOADV_ERANGE
    assert b"f16x2" in lib.geoadv_last_error()
    assert lib.geoadv_ae_status(ae.handle, _lib.stream_handle()) == 0          # reported once
    with pytest.raises(RuntimeError, match="f16x2"):
        ae.transform(pc_out)
    # the attack on such clouds: status() (which attack() calls before it returns anything) raises, defend_* raise
    from geometric_adv_amd._lib import GeoAdvError
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd import defense
    at = AdvAE("a", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=2, num_iterations_thresh=1), ae=ae)
    at.set_inputs(pc_out, pc_out[::-1].copy(), None, 1.0)
    at.init_pert(None, reset_optimizer=True)
    at.run(0, 2, 1, torch.empty((2, 6, b), device=ae.device))
    with pytest.raises(GeoAdvError, match="f16x2"):
        at.status()
    at.status()                                                         # (cleared)
    with pytest.raises(GeoAdvError, match="f16x2"):
        defense.defend_critical(ae, pc_out, pc_out)
    del at
    ae3 = PointNetAE(w, n, encoder_arith="bf16x3")
    z3 = ae3.transform(pc_out)
    z64 = model.encode(pc_out.astype(np.float64))
    np.testing.assert_allclose(z3 / np.abs(z64).max(), z64 / np.abs(z64).max(), atol=2e-6)


@pytest.mark.parametrize("log2_gamma", [-20, -8, 8])
def test_f16x2_activation_scale_follows_the_batch_norm(log2_gamma):
    """Layer 0's gamma and beta multiplied by 2^k and layer 1's weights divided by it: layer 0's activations are 2^k times what
    they were, everything behind them is unchanged -- and so is f16x2's accuracy, because the power of two a layer's activations are
    carried times follows the layer's own batch-norm constants (with the fixed 2^6 of the first build the latent was 6e-4 off at
    k = -20: tools/debug/h2_low_side.py)."""
    from geometric_adv_amd import _lib, weights as W
    from geometric_adv_amd.autoencoder import PointNetAE
    from oracle.attack_model import AEModel
    from conftest import cloud
    n, b = 1024, 6
    f = np.float64(2.0) ** log2_gamma
    w = dict(W.randomized_weights(n, seed=9))
    for k in ("autoencoder/encoder_conv_layer_0_bnorm/gamma", "autoencoder/encoder_conv_layer_0_bnorm/beta"):
        w[k] = (np.asarray(w[k], dtype=np.float64) * f).astype(np.float32)
    k = "autoencoder/encoder_conv_layer_1/W"
    w[k] = (np.asarray(w[k], dtype=np.float64) / f).astype(np.float32)
    pc = cloud(73, b, n)
    ae = PointNetAE(w, n)
    assert ae.encoder_arith == "f16x2"
    z = ae.transform(pc)
    z64 = AEModel(W.canonical(w, n), n, np.float64).encode(pc.astype(np.float64))
    np.testing.assert_allclose(z / np.abs(z64).max(), z64 / np.abs(z64).max(), atol=2e-6)
    assert _lib.lib().geoadv_ae_status(ae.handle, _lib.stream_handle()) == 0


def test_f16x2_is_refused_for_a_model_that_does_not_scale():
    """A non-finite encoder weight: the library default falls back to bf16x3 and an explicit f16x2 is refused with a message."""
    from geometric_adv_amd import _lib, weights as W
    from geometric_adv_amd.autoencoder import PointNetAE
    n = 256
    w = dict(W.randomized_weights(n, seed=2))
    k = "autoencoder/encoder_conv_layer_2/W"
    a = np.array(w[k], dtype=np.float32); a[0, 0, 3, 5] = np.inf
    w[k] = a
    ae = PointNetAE(w, n)
    assert ae.encoder_arith == "bf16x3"
    assert _lib.lib().geoadv_ae_set_encoder_arith(ae.handle, 2) != 0
    assert b"f16x2" in _lib.lib().geoadv_last_error()


def test_arithmetics_on_a_trained_victim_and_its_adversarial_clouds():
    """A victim trained with the repo's trainer (batch statistics folded into moving averages, activations no longer those of the
    random initialisation): latents of clean and of attacked clouds under every arithmetic against the float64 model at the
    path's 2e-6, the range guard quiet, and the largest activation the float64 model sees two orders of magnitude inside f16x2's
    range."""
    import torch
    from geometric_adv_amd import _lib, weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd.autoencoder import PointNetAE
    from geometric_adv_amd.trainer import PointNetAETrainer, initial_weights
    from oracle.attack_model import AEModel
    n, b = 512, 8
    rng = np.random.default_rng(0)

    def shapes(count):
        u = rng.standard_normal((count, n, 3)).astype(np.float32)
        u /= np.linalg.norm(u, axis=2, keepdims=True)
        scale = rng.uniform(0.15, 0.45, size=(count, 1, 3)).astype(np.float32)
        return (u * scale).astype(np.float32)

    data = shapes(64)
    tr = PointNetAETrainer(initial_weights(n, seed=2), n, batch_size=b, learning_rate=0.002)
    for _ in range(12):
        tr._single_epoch_train(data)
    w = tr.export_weights()
    del tr
    model = AEModel(W.canonical(w, n), n, np.float64)
    src, tgt = shapes(b), shapes(b)
    ae = PointNetAE(w, n)
    assert ae.encoder_arith == "f16x2"
    at = AdvAE("a", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=40, num_iterations_thresh=20, learning_rate=0.01), ae=ae)
    at.set_inputs(src, tgt, ae.transform(tgt), 1.0)
    at.init_pert(None, reset_optimizer=True)
    at.run(0, 40, 20, torch.empty((40, 6, b), device=ae.device))
    at.status()
    adv = at.peek()["adv"].cpu().numpy()
    for pcs in (src, adv):
        z64, hs = model.encode(pcs.astype(np.float64), keep=True)
        assert max(float(h.max()) for h in hs) < 10.0                   # (1023.5 is the guard)
        sc = np.abs(z64).max()
        for arith in ARITHS:
            a = PointNetAE(w, n, encoder_arith=arith)
            np.testing.assert_allclose(a.transform(pcs) / sc, z64 / sc, atol=2e-6, err_msg=arith)
            assert _lib.lib().geoadv_ae_status(a.handle, _lib.stream_handle()) == 0


@pytest.mark.parametrize("log2_factor", [-40, 30])
def test_f16x2_weight_scaling_follows_the_layer(log2_factor):
    """One layer's weights and bias multiplied by 2^k with its batch-norm mean multiplied and its gamma divided likewise (var is
    left alone, so the float64 model is rebuilt from the modified weights): the per-layer power of two that f16x2 scales
    the weights by follows, and the latent stays within the path's 2e-6 of the float64 model."""
    from geometric_adv_amd import _lib, weights as W
    from geometric_adv_amd.autoencoder import PointNetAE
    from oracle.attack_model import AEModel
    from conftest import cloud
    n, b = 1024, 5
    f = np.float64(2.0) ** log2_factor
    w = dict(W.randomized_weights(n, seed=13))
    for layer in (2, 4):
        pre = "autoencoder/encoder_conv_layer_%d" % layer
        var = np.asarray(w[pre + "_bnorm/moving_variance"], dtype=np.float64)
        w[pre + "/W"] = (np.asarray(w[pre + "/W"], dtype=np.float64) * f).astype(np.float32)
        w[pre + "/b"] = (np.asarray(w[pre + "/b"], dtype=np.float64) * f).astype(np.float32)
        w[pre + "_bnorm/moving_mean"] = (np.asarray(w[pre + "_bnorm/moving_mean"], dtype=np.float64) * f).astype(np.float32)
        w[pre + "_bnorm/moving_variance"] = (var * f * f).astype(np.float32)          # inv = gamma / sqrt(var f^2 + 1e-5)
    pc = cloud(81, b, n)
    ae = PointNetAE(w, n)
    assert ae.encoder_arith == "f16x2"
    z = ae.transform(pc)
    z64 = AEModel(W.canonical(w, n), n, np.float64).encode(pc.astype(np.float64))
    sc = np.abs(z64).max()
    assert sc > 0 and np.isfinite(z).all()
    np.testing.assert_allclose(z / sc, z64 / sc, atol=2e-6)
    assert _lib.lib().geoadv_ae_status(ae.handle, _lib.stream_handle()) == 0
