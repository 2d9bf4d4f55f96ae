"""CPU: host-side logic of the package (no GPU, no compute calls through the C ABI)."""
import os

import numpy as np
import pytest


def test_weights_roundtrip_and_validation(tmp_path):
    from geometric_adv_amd import weights as W
    w = W.synthetic_weights(256)
    assert sorted(w) == sorted(W.variable_names())
    assert w["autoencoder/encoder_conv_layer_0/W"].shape == (1, 1, 3, 64)          # tflearn conv_1d filter shape
    assert w["autoencoder/decoder_fc_2/W"].shape == (256, 768)
    p = str(tmp_path / "ae.npz")
    W.save_npz(p, w)
    w2 = W.load_npz(p)
    assert all(np.array_equal(w[k], w2[k]) for k in w)
    c = W.canonical(w2, 256)
    assert c["enc_w"][3].shape == (128, 256) and c["dec_w"][2].shape == (256, 768)
    bad = dict(w); del bad["autoencoder/decoder_fc_1/b"]
    with pytest.raises(KeyError):
        W.canonical(bad, 256)
    with pytest.raises(ValueError):
        W.canonical(w, 128)                                                        # decoder sized for 256 points


def test_init_pert_is_truncated_normal():
    from geometric_adv_amd.adversary import init_pert_value, get_pert_loss_np
    p = init_pert_value(4, 512)
    assert p.shape == (4, 512, 3) and p.dtype == np.float32
    assert np.abs(p).max() <= 2e-7 + 1e-12 and 0.5e-7 < p.std() < 1.1e-7           # sigma 1e-7, cut at 2 sigma
    assert np.array_equal(p, init_pert_value(4, 512))                              # seed 55 => reproducible
    lp, lm = get_pert_loss_np(p)
    np.testing.assert_allclose(lp, np.sqrt((p.astype(np.float64) ** 2).sum((1, 2))), rtol=1e-5)
    np.testing.assert_allclose(lm, np.sqrt((p.astype(np.float64) ** 2).sum(2).max(1)), rtol=1e-5)


def test_ops_reject_cpu_tensors_and_bad_shapes():
    """There is no CPU path: CPU tensors are an error, like a wrong rank (tf_nndistance.cpp:51-58)."""
    import torch
    from geometric_adv_amd import ops
    with pytest.raises(ValueError, match="no CPU path"):
        ops.nn_distance(torch.rand(2, 5, 3), torch.rand(2, 5, 3))
    with pytest.raises(ValueError):
        ops.nn_distance(torch.rand(5, 3), torch.rand(2, 5, 3))
    with pytest.raises(TypeError):
        ops.nn_distance(np.zeros((2, 5, 3), np.float32), np.zeros((2, 5, 3), np.float32))


def test_configuration_validation():
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    conf = Configuration(batch_size=2, n_points=64, weights={}, loss="emd")
    with pytest.raises(ValueError, match="chamfer"):
        AdvAE("adversary", conf)
    conf = Configuration(batch_size=2, n_points=64, weights={}, loss_adv_type="bogus")
    with pytest.raises(ValueError):
        AdvAE("adversary", conf)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from geometric_adv_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.GeoAdvError, match="not built"):
        _lib.lib()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under geometric_adv_amd/ may reference it."""
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "geometric_adv_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("oracle of record", ""), os.path.join(dirpath, f)


def test_attack_model_gradient_matches_finite_differences():
    """The numpy model's hand-written backward (Appendix A) against central differences, fp64."""
    from geometric_adv_amd import weights as W
    from oracle.attack_model import AEModel, AttackModel
    from conftest import cloud
    n, b = 64, 2
    w = W.randomized_weights(n)
    m = AEModel(W.canonical(w, n), n)
    x, gt = cloud(1, b, n), cloud(2, b, n)
    tz = m.encode(gt)
    rng = np.random.default_rng(0)
    for adv_t, dist_t in [("chamfer", "chamfer"), ("latent", "pert")]:
        am = AttackModel(m, x, gt, tz, np.array([1.0, 3.0]), adv_t, dist_t, fp32_state=False)
        am.init_pert(1e-2 * rng.standard_normal((b, n, 3)))
        f = am.forward()
        g = am.gradient(f)
        d = rng.standard_normal(am.pert.shape)
        eps = 1e-6
        p0 = am.pert.copy()
        vals = []
        for sgn in (+1, -1):
            am.pert = p0 + sgn * eps * d
            fs = am.forward(idx_override=f["idx"])
            vals.append((fs["loss_adv"] + am.w * fs["loss_dist"]).sum())
        fd = (vals[0] - vals[1]) / (2 * eps)
        np.testing.assert_allclose((g * d).sum(), fd, rtol=2e-3)


# ---------------------------------------------------------------- TF V2 checkpoint reader
def test_crc32c_known_answers():
    from geometric_adv_amd import tf_checkpoint as T
    assert T.crc32c(b"123456789") == 0xE3069283                       # the CRC-32C check value
    assert T.crc32c(bytes(32)) == 0x8A9136AA                          # RFC 3720 B.4
    assert T.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    rng = np.random.default_rng(5)
    for n in (16384, 70001, (1 << 20) + 77):                          # lockstep/vectorised path == scalar recurrence
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert T.crc32c(b) == T._crc_scalar(b, 0xFFFFFFFF) ^ 0xFFFFFFFF
    assert T.unmask_crc(T.mask_crc(0xDEADBEEF)) == 0xDEADBEEF


def test_tf_checkpoint_round_trip(tmp_path):
    from geometric_adv_amd import tf_checkpoint as T, weights as W
    w = W.randomized_weights(256)
    extra = dict(w)
    extra["autoencoder/encoder_conv_layer_0/W/Adam"] = np.zeros((1, 1, 3, 64), np.float32)    # optimizer slots are skipped
    extra["beta1_power"] = np.array(0.5, np.float32)
    extra["global_step"] = np.array(500, np.int64)
    prefix = str(tmp_path / "models.ckpt-500")
    T.write_checkpoint(prefix, extra, block_size=512)                 # several data blocks -> index block is exercised
    names = dict(T.list_variables(prefix))
    assert names["global_step"] == () and names["autoencoder/decoder_fc_2/W"] == (256, 768)
    got = T.restore_ae_weights(str(tmp_path), 500)
    assert set(got) == set(W.variable_names())
    for k in got:
        assert got[k].dtype == w[k].dtype and np.array_equal(got[k], w[k])
    assert np.array_equal(W.load(prefix)["autoencoder/decoder_fc_0/b"], w["autoencoder/decoder_fc_0/b"])
    assert T.load_checkpoint(prefix)["global_step"] == 500
    # corruption is detected: flip one byte of a tensor, then one of the index
    data = T.shard_path(prefix, 0, 1)
    raw = bytearray(open(data, "rb").read()); raw[100] ^= 1; open(data, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        T.load_checkpoint(prefix)
    raw[100] ^= 1; open(data, "wb").write(bytes(raw))
    idx = bytearray(open(prefix + ".index", "rb").read()); idx[10] ^= 1; open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError, match="checksum"):
        T.read_index(prefix)
    with pytest.raises(ValueError, match="magic"):
        open(prefix + ".index", "wb").write(bytes(idx[:-1]) + b"\x00")
        T.read_index(prefix)


def test_tf_checkpoint_snappy_block():
    from geometric_adv_amd import tf_checkpoint as T
    # literal "abcd", copy(offset 4, len 8) via a 1-byte-offset tag, literal "xyz"  -> "abcdabcdabcdxyz"
    comp = bytes([15, (4 - 1) << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4]) + bytes([(3 - 1) << 2]) + b"xyz"
    assert T._snappy_decompress(comp) == b"abcdabcdabcdxyz"


# ---------------------------------------------------------------- training-step oracle
def test_train_model_gradients_match_finite_differences():
    """oracle/train_model.py is unpinned against TF; its backward (BN in training mode, differentiated through the
    batch statistics; max-pool; Chamfer with pinned matches) is checked against central differences in fp64."""
    from geometric_adv_amd import weights as W
    from oracle.train_model import TrainModel, PARAM_GROUPS
    from oracle.attack_model import _o
    n, b = 64, 3
    tm = TrainModel(W.canonical(W.randomized_weights(n, seed=11), n), n)
    rng = np.random.default_rng(0)
    x = (rng.random((b, n, 3), dtype=np.float32) - 0.5)
    loss, G, c = tm.loss_and_grads(x)
    _, i1, _, i2 = _o().nn_distance(c["recon"].astype(np.float32), x)
    idx = (i1.astype(np.int64), i2.astype(np.int64))

    def f():
        return tm.chamfer_loss_fixed(tm.forward(x)["recon"], x.astype(np.float64), *idx)

    assert abs(f() - loss) < 1e-12
    errs = []
    for k in PARAM_GROUPS:
        for j, a in enumerate(tm.p[k]):
            flat = a.reshape(-1)
            for t in rng.choice(flat.size, size=min(4, flat.size), replace=False):
                old, h = flat[t], 1e-8
                flat[t] = old + h; fp = f(); flat[t] = old - h; fm = f(); flat[t] = old
                an = G[k][j].reshape(-1)[t]
                if abs(an) > 1e-4:
                    errs.append(abs((fp - fm) / (2 * h) - an) / abs(an))
    errs = np.array(errs)
    assert len(errs) > 40 and np.median(errs) < 1e-5 and np.percentile(errs, 90) < 1e-3    # a few kinks (ReLU, max) are expected
    for g in G["enc_b"]:                                  # a bias in front of a batch norm has exactly zero gradient
        assert np.abs(g).max() < 1e-12
    l0 = tm.step(x)[0]
    for _ in range(6):
        l1 = tm.step(x)[0]
    assert l1 < 0.5 * l0


def test_trainer_initial_weights_follow_the_reference_initialisers():
    """conv_1d: tflearn 'uniform_scaling' U(+-sqrt(3/fan_in)); fully_connected: weights_init='xavier'
    (encoders_decoders.py:107,132) = U(+-sqrt(6/(fan_in+fan_out))) -- per-layer std and bounds."""
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.trainer import initial_weights
    n = 256
    w = initial_weights(n, seed=3)
    ed, dd = W.enc_dims(), W.dec_dims(n)
    for i in range(5):
        a = w["autoencoder/encoder_conv_layer_%d/W" % i]
        lim = np.sqrt(3.0 / ed[i])
        assert np.abs(a).max() <= lim and abs(a.std() / (lim / np.sqrt(3)) - 1) < 0.1
    for k in range(3):
        a = w["autoencoder/decoder_fc_%d/W" % k]
        lim = np.sqrt(6.0 / (dd[k] + dd[k + 1]))
        assert a.shape == (dd[k], dd[k + 1]) and np.abs(a).max() <= lim
        assert abs(a.std() / (lim / np.sqrt(3)) - 1) < 0.03
    assert abs(w["autoencoder/decoder_fc_0/W"].std() - 0.072) < 0.003          # 128 -> 256: not the 0.02 of a truncated normal


def test_bench_parent_launcher_fails_cleanly_without_gpus():
    """bench.py --gpus 2 from a bare shell spawns its ranks itself (before touching any GPU); on a box without GPUs the ranks
    die on their assertion and the parent must relay a non-zero status and no JSON line -- no hang, no half-printed result.
    (GEOADV_BENCH_SHARE_GPU=1 skips the parent's device-count check, which would otherwise refuse before spawning.)"""
    import subprocess, sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for the GPU-less build container")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["GEOADV_BENCH_SHARE_GPU"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "2-rank run failed" in p.stderr


def test_bench_refuses_more_ranks_than_gpus_quickly():
    """`bench.py --gpus N` on a node with fewer than N GPUs (here: none) must end within seconds with a message naming
    the problem, before any rank is started -- not hang in a rendezvous."""
    import subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has the GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GEOADV_BENCH_SHARE_GPU")}
    t0 = time.time()
    o = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert o.returncode == 2 and "one rank per GPU" in o.stderr and time.time() - t0 < 60, (o.returncode, o.stderr[-300:])
