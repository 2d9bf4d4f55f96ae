"""Victim auto-encoder weights: the reference's TF variable names <-> arrays.

`restore_ae_model` (src/adversary_autoencoder.py:42-51) loads the variables prefixed with the AE
name ('autoencoder') from a TF1 checkpoint.  This build keeps them in an .npz keyed by the same
variable names (tflearn 0.3.2 naming):
    autoencoder/encoder_conv_layer_{i}/W   [1,1,Cin,Cout] (or [1,Cin,Cout] / [Cin,Cout])   i = 0..4
    autoencoder/encoder_conv_layer_{i}/b   [Cout]
    autoencoder/encoder_conv_layer_{i}_bnorm/{beta,gamma,moving_mean,moving_variance}  [Cout]
    autoencoder/decoder_fc_{k}/W [in,out], autoencoder/decoder_fc_{k}/b [out]               k = 0..2
No trained checkpoint ships with the reference (download_models_and_data.sh), so benchmarks and
tests use `synthetic_weights` (seed 7, Glorot-uniform W, zero b, near-identity BN).
"""
import numpy as np

ENC_FILTERS = [64, 128, 128, 256, 128]          # src/ae_templates.py:22 with bneck_size = 128
DEC_SIZES = [256, 256]                          # src/ae_templates.py:29 (+ [n_points * 3])
AE_NAME = "autoencoder"
KERNEL_BNECK = 128                              # the width the kernels are compiled for (ae.hip).  mlp_architecture(n_pc_points,
                                                # bneck_size, ...) (src/ae_templates.py:11-39) with bneck_size < 128 runs on them with
                                                # the missing channels as EXACT zeros (canonical(pad_to=...)); larger ones are refused


def enc_dims(bneck=128):
    return [3] + ENC_FILTERS[:-1] + [bneck]


def dec_dims(n_points, bneck=128):
    return [bneck] + DEC_SIZES + [3 * n_points]


def variable_names(ae_name=AE_NAME):
    names = []
    for i in range(5):
        names += ["%s/encoder_conv_layer_%d/W" % (ae_name, i), "%s/encoder_conv_layer_%d/b" % (ae_name, i)]
        names += ["%s/encoder_conv_layer_%d_bnorm/%s" % (ae_name, i, v)
                  for v in ("beta", "gamma", "moving_mean", "moving_variance")]
    for k in range(3):
        names += ["%s/decoder_fc_%d/W" % (ae_name, k), "%s/decoder_fc_%d/b" % (ae_name, k)]
    return names


def synthetic_weights(n_points, seed=7, ae_name=AE_NAME, bneck=128):
    """Seeded random-init weights of the reference architecture (dict name -> float32 array)."""
    rng = np.random.default_rng(seed)
    w = {}
    ed, dd = enc_dims(bneck), dec_dims(n_points, bneck)
    for i in range(5):
        cin, cout = ed[i], ed[i + 1]
        lim = np.sqrt(6.0 / (cin + cout))
        p = "%s/encoder_conv_layer_%d" % (ae_name, i)
        w[p + "/W"] = rng.uniform(-lim, lim, size=(1, 1, cin, cout)).astype(np.float32)
        w[p + "/b"] = np.zeros(cout, np.float32)
        w[p + "_bnorm/gamma"] = (1.0 + 0.002 * rng.standard_normal(cout)).astype(np.float32)
        w[p + "_bnorm/beta"] = np.zeros(cout, np.float32)
        w[p + "_bnorm/moving_mean"] = (0.1 * rng.standard_normal(cout)).astype(np.float32)
        w[p + "_bnorm/moving_variance"] = rng.uniform(0.5, 1.5, size=cout).astype(np.float32)
    for k in range(3):
        cin, cout = dd[k], dd[k + 1]
        lim = np.sqrt(6.0 / (cin + cout))
        p = "%s/decoder_fc_%d" % (ae_name, k)
        w[p + "/W"] = rng.uniform(-lim, lim, size=(cin, cout)).astype(np.float32)
        w[p + "/b"] = np.zeros(cout, np.float32)
    return w


def randomized_weights(n_points, seed=3, ae_name=AE_NAME, bneck=128):
    """Like synthetic_weights but with non-trivial biases / BN offsets (stress for parity tests)."""
    w = synthetic_weights(n_points, seed=seed, ae_name=ae_name, bneck=bneck)
    rng = np.random.default_rng(seed + 1000)
    for name in list(w):
        if name.endswith("/b") or name.endswith("/beta"):
            w[name] = (0.05 * rng.standard_normal(w[name].shape)).astype(np.float32)
        if name.endswith("/gamma"):
            w[name] = rng.uniform(0.7, 1.3, size=w[name].shape).astype(np.float32)
    return w


def save_npz(path, weights):
    np.savez(path, **{k.replace("/", "__"): v for k, v in weights.items()})


def load_npz(path):
    with np.load(path) as z:
        return {k.replace("__", "/"): z[k] for k in z.files}


def load(path, ae_name=AE_NAME, restore_epoch=None):
    """Victim weights from wherever they live: an .npz (save_npz), a TF V2 checkpoint prefix
    ('.../models.ckpt-500', i.e. the argument of saver.restore in adversary_autoencoder.py:48), or a model
    directory plus `restore_epoch` (conf.ae_dir / conf.ae_restore_epoch, adv_ae.py:76)."""
    import os
    from . import tf_checkpoint
    if restore_epoch is not None:
        return tf_checkpoint.restore_ae_weights(path, restore_epoch, ae_name)
    if path.endswith(".npz"):
        return load_npz(path)
    if os.path.exists(path + ".index"):
        wanted = set(variable_names(ae_name))
        return tf_checkpoint.load_checkpoint(path, lambda n: n in wanted)
    raise FileNotFoundError("%s is neither an .npz nor a TF V2 checkpoint prefix" % path)


def bneck_of(weights, ae_name=AE_NAME):
    """The bottleneck size of a set of weights: output channels of the last encoder layer."""
    name = "%s/encoder_conv_layer_4/W" % ae_name
    if name not in weights:
        raise KeyError("missing variable %r (restore_ae_model needs every '%s/*' variable)" % (name, ae_name))
    return int(np.asarray(weights[name]).shape[-1])


def canonical(weights, n_points, ae_name=AE_NAME, pad_to=None):
    """Validate and reshape to what geoadv_ae_create takes: lists of contiguous float32 arrays, at the weights' own bottleneck
    size (mlp_architecture's bneck_size, src/ae_templates.py:11-39; bneck_of()).

    pad_to (the kernels' compiled width, KERNEL_BNECK): a smaller bottleneck is widened with channels that are EXACT zeros for
    every input -- zero weight columns and bias, BN gamma 1 / beta 0 / mean 0 / var 1 in the last encoder layer (relu(0 * s + 0) =
    +0), zero rows in the first decoder layer -- so every real channel, the reconstruction and every gradient are computed by the
    same instructions on the same numbers as at width bneck.  A zero channel is never a 'tied positive maximum' (decoder.hip) and
    never a critical point (ae_utils.py:21 drops max_val == 0)."""
    bneck = bneck_of(weights, ae_name)
    if pad_to is not None and bneck > pad_to:
        raise ValueError("bneck_size %d: the kernels are built for bottlenecks of up to %d channels (src/ae_templates.py:11-39 allows any; "
                         "the reference's own scripts use 128, autoencoder/train_ae.py:45)" % (bneck, pad_to))
    ed, dd = enc_dims(bneck), dec_dims(n_points, bneck)
    out = {"enc_w": [], "enc_b": [], "gamma": [], "beta": [], "mean": [], "var": [], "dec_w": [], "dec_b": []}

    def get(name, shape):
        if name not in weights:
            raise KeyError("missing variable %r (restore_ae_model needs every '%s/*' variable)" % (name, ae_name))
        a = np.asarray(weights[name], dtype=np.float32)
        if a.size != int(np.prod(shape)):
            raise ValueError("variable %r has shape %s, expected %s" % (name, a.shape, tuple(shape)))
        return np.ascontiguousarray(a.reshape(shape))

    for i in range(5):
        p = "%s/encoder_conv_layer_%d" % (ae_name, i)
        out["enc_w"].append(get(p + "/W", (ed[i], ed[i + 1])))
        out["enc_b"].append(get(p + "/b", (ed[i + 1],)))
        out["gamma"].append(get(p + "_bnorm/gamma", (ed[i + 1],)))
        out["beta"].append(get(p + "_bnorm/beta", (ed[i + 1],)))
        out["mean"].append(get(p + "_bnorm/moving_mean", (ed[i + 1],)))
        out["var"].append(get(p + "_bnorm/moving_variance", (ed[i + 1],)))
    for k in range(3):
        p = "%s/decoder_fc_%d" % (ae_name, k)
        out["dec_w"].append(get(p + "/W", (dd[k], dd[k + 1])))
        out["dec_b"].append(get(p + "/b", (dd[k + 1],)))
    if pad_to is not None and bneck < pad_to:
        extra = pad_to - bneck
        z = lambda *shape: np.zeros(shape, np.float32)
        out["enc_w"][4] = np.ascontiguousarray(np.concatenate([out["enc_w"][4], z(ed[4], extra)], axis=1))
        out["enc_b"][4] = np.concatenate([out["enc_b"][4], z(extra)])
        out["gamma"][4] = np.concatenate([out["gamma"][4], np.ones(extra, np.float32)])
        out["beta"][4] = np.concatenate([out["beta"][4], z(extra)])
        out["mean"][4] = np.concatenate([out["mean"][4], z(extra)])
        out["var"][4] = np.concatenate([out["var"][4], np.ones(extra, np.float32)])
        out["dec_w"][0] = np.ascontiguousarray(np.concatenate([out["dec_w"][0], z(extra, dd[1])], axis=0))
    return out
