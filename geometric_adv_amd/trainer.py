"""Training the victim auto-encoder on MI355X.

Host-side mirror of the training surface of the reference's `PointNetAutoEncoder`
(src/pointnet_ae.py:71-139) / `AutoEncoder` (src/autoencoder.py:105-125,196-227): `partial_fit`,
`_single_epoch_train`, `train`, plus what `saver.save` would write (`export_weights`, TF variable names, also
as a TF V2 checkpoint through tf_checkpoint.write_checkpoint).  The arithmetic runs in libgeoadv.so
(csrc/train.hip); there is no CPU fallback.

Data-parallel training: one process per GPU; each rank runs the step on its shard of the batch and the flat gradient
buffer is sum-all-reduced over RCCL (`torch.distributed`, backend "nccl").  With `sync_bn=True` (default) the encoder's
batch-norm statistics are shared too: the step runs in phases and ten <= 4 KB all-reduces carry the per-channel sums,
so `world` ranks with `batch_size` clouds each take exactly the reference's step on world * batch_size clouds.
`sync_bn=False` keeps per-rank statistics (one collective per step).
"""
import ctypes as C
import time

import numpy as np
import torch

from . import _lib, weights as W
from .autoencoder import _AEWeights


class _TrainConfig(C.Structure):
    _fields_ = [("batch", C.c_int), ("learning_rate", C.c_float), ("bn_decay", C.c_float), ("loss", C.c_int),
                ("max_workgroups", C.c_int)]


def _fill_weights_struct(hw, canon, n_points):
    hw.n_points = n_points
    hw.enc_dims[:] = W.enc_dims()
    hw.dec_dims[:] = W.dec_dims(n_points)
    for i in range(5):
        hw.enc_w[i] = canon["enc_w"][i].ctypes.data
        hw.enc_b[i] = canon["enc_b"][i].ctypes.data
        hw.bn_gamma[i] = canon["gamma"][i].ctypes.data
        hw.bn_beta[i] = canon["beta"][i].ctypes.data
        hw.bn_mean[i] = canon["mean"][i].ctypes.data
        hw.bn_var[i] = canon["var"][i].ctypes.data
    for k in range(3):
        hw.dec_w[k] = canon["dec_w"][k].ctypes.data
        hw.dec_b[k] = canon["dec_b"][k].ctypes.data


def initial_weights(n_points, seed=0, ae_name=W.AE_NAME):
    """Fresh variables with the initialisers the reference's layers are built with (tflearn 0.3.2, third-party, restated):
    conv_1d W: tflearn's default 'uniform_scaling' = U(+-sqrt(3/fan_in)), fan_in = Cin for a width-1 filter
    (encoders_decoders.py:43-44 passes no weights_init); fully_connected W: weights_init='xavier'
    (encoders_decoders.py:107,132) = tf.contrib.layers.xavier_initializer(uniform=True) = U(+-sqrt(6/(fan_in+fan_out)));
    biases 0; BN gamma ~ N(1, 0.002), beta 0, moving_mean 0, moving_variance 1."""
    rng = np.random.default_rng(seed)
    w = {}
    ed, dd = W.enc_dims(), W.dec_dims(n_points)
    for i in range(5):
        cin, cout = ed[i], ed[i + 1]
        lim = np.sqrt(3.0 / cin)
        p = "%s/encoder_conv_layer_%d" % (ae_name, i)
        w[p + "/W"] = rng.uniform(-lim, lim, size=(1, 1, cin, cout)).astype(np.float32)
        w[p + "/b"] = np.zeros(cout, np.float32)
        w[p + "_bnorm/gamma"] = (1.0 + 0.002 * rng.standard_normal(cout)).astype(np.float32)
        w[p + "_bnorm/beta"] = np.zeros(cout, np.float32)
        w[p + "_bnorm/moving_mean"] = np.zeros(cout, np.float32)
        w[p + "_bnorm/moving_variance"] = np.ones(cout, np.float32)
    for k in range(3):
        cin, cout = dd[k], dd[k + 1]
        p = "%s/decoder_fc_%d" % (ae_name, k)
        lim = np.sqrt(6.0 / (cin + cout))                  # Glorot / Xavier uniform
        w[p + "/W"] = rng.uniform(-lim, lim, size=(cin, cout)).astype(np.float32)
        w[p + "/b"] = np.zeros(cout, np.float32)
    return w


class PointNetAETrainer:
    """One model replica on one GPU with a fixed batch size."""

    GROUPS = ("enc_w", "enc_b", "gamma", "beta", "dec_w", "dec_b")

    def __init__(self, weights, n_points, batch_size=50, learning_rate=0.0005, bn_decay=0.9, ae_name=W.AE_NAME,
                 device=None, sync_bn=True, loss="chamfer", max_workgroups=0):
        if isinstance(weights, str):
            weights = W.load(weights, ae_name)
        self.n_points, self.batch_size, self.ae_name = int(n_points), int(batch_size), ae_name
        self.device = torch.device(device if device is not None else "cuda:0")
        canon = W.canonical(weights, self.n_points, ae_name)
        hw = _AEWeights()
        _fill_weights_struct(hw, canon, self.n_points)
        if loss not in ("chamfer", "emd"):                       # conf.loss (src/pointnet_ae.py:74-79)
            raise ValueError("loss must be 'chamfer' or 'emd'")
        self.loss = loss
        cfg = _TrainConfig(self.batch_size, float(learning_rate), float(bn_decay), 1 if loss == "emd" else 0, int(max_workgroups))
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().geoadv_trainer_create(C.byref(self._h), C.byref(hw), C.byref(cfg)), "trainer_create")
        offs = (C.c_size_t * 26)()
        _lib.check(_lib.lib().geoadv_trainer_layout(self._h, offs), "trainer_layout")
        self._offsets = [int(o) for o in offs]
        pp, gp, cnt = C.c_void_p(), C.c_void_p(), C.c_size_t()
        _lib.check(_lib.lib().geoadv_trainer_buffers(self._h, C.byref(pp), C.byref(gp), C.byref(cnt)), "trainer_buffers")
        self._count = int(cnt.value)
        self._params_ptr, self._grads_ptr = pp.value, gp.value
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.epoch = 0
        self.sync_bn = bool(sync_bn)
        self._world_set = 1

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                _lib.lib().geoadv_trainer_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    # ---- flat buffers -------------------------------------------------------------------------------------
    def _view(self, ptr):
        """The library-owned flat buffer as a torch tensor (no copy): what the gradient all-reduce runs on."""
        class _Arr:
            pass
        a = _Arr()
        a.__cuda_array_interface__ = {"shape": (self._count,), "typestr": "<f4", "data": (ptr, False), "version": 2}
        with torch.cuda.device(self.device):
            return torch.as_tensor(a, device=self.device)

    def _view_f64(self, ptr, count):
        class _Arr:
            pass
        a = _Arr()
        a.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}
        with torch.cuda.device(self.device):
            return torch.as_tensor(a, device=self.device)

    def _set_world(self, world):
        if world != self._world_set:
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib().geoadv_trainer_set_world(self._h, int(world)), "trainer_set_world")
            self._world_set = world

    def _step_synchronised(self, x, gt, want_recon, world):
        """The step in phases, batch-norm sums all-reduced in between (see include/geoadv.h)."""
        from .dist import all_reduce_sum_
        L = _lib.lib()
        self._set_world(world)
        buf, cnt = C.c_void_p(), C.c_size_t()
        with torch.cuda.device(self.device):
            for p in range(L.geoadv_trainer_num_phases()):
                _lib.check(L.geoadv_trainer_run_phase(self._h, p, _lib.ptr(x), _lib.ptr(gt), _lib.stream_handle()), "trainer_run_phase")
                _lib.check(L.geoadv_trainer_exchange(self._h, p, C.byref(buf), C.byref(cnt)), "trainer_exchange")
                if cnt.value:
                    all_reduce_sum_(self._view_f64(buf.value, int(cnt.value)))
            recon = torch.empty_like(x) if want_recon else None
            _lib.check(L.geoadv_trainer_fetch(self._h, _lib.ptr(self._loss), _lib.ptr(recon), _lib.stream_handle()), "trainer_fetch")
        all_reduce_sum_(self.gradient_buffer())
        loss = all_reduce_sum_(self._loss.clone())
        self.apply(1.0)
        return recon, loss

    def gradient_buffer(self):
        return self._view(self._grads_ptr)

    def parameter_buffer(self):
        return self._view(self._params_ptr)

    def _shapes(self):
        ed, dd = W.enc_dims(), W.dec_dims(self.n_points)
        shapes = [(ed[i], ed[i + 1]) for i in range(5)] + [(ed[i + 1],) for i in range(5)] * 3
        shapes += [(dd[k], dd[k + 1]) for k in range(3)] + [(dd[k + 1],) for k in range(3)]
        return shapes

    def _unflatten(self, flat):
        """flat host array -> {group: [arrays]} in the layout order of geoadv_trainer_layout."""
        out, idx = {}, 0
        for g, cnt in zip(self.GROUPS, (5, 5, 5, 5, 3, 3)):
            out[g] = []
            for _ in range(cnt):
                shp = self._shapes()[idx]
                o = self._offsets[idx]
                out[g].append(flat[o:o + int(np.prod(shp))].reshape(shp).copy())
                idx += 1
        return out

    def gradients(self):
        """d loss / d variable of the last forward_backward, {group: [arrays]} (host)."""
        torch.cuda.synchronize(self.device)
        return self._unflatten(self.gradient_buffer().cpu().numpy())

    # ---- the step ----------------------------------------------------------------------------------------
    def _dev(self, a):
        t = torch.as_tensor(np.asarray(a, dtype=np.float32)) if not isinstance(a, torch.Tensor) else a
        t = t.to(self.device, dtype=torch.float32).contiguous()
        if tuple(t.shape) != (self.batch_size, self.n_points, 3):
            raise ValueError("batch must be of shape (%d, %d, 3); got %s" % (self.batch_size, self.n_points, tuple(t.shape)))
        return t

    def forward_backward(self, X, GT=None, want_recon=True):
        x = self._dev(X)
        gt = self._dev(GT) if GT is not None else None
        recon = torch.empty_like(x) if want_recon else None
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().geoadv_trainer_forward_backward(self._h, _lib.ptr(x), _lib.ptr(gt), _lib.ptr(self._loss),
                                                                  _lib.ptr(recon), _lib.stream_handle()), "trainer_forward_backward")
        return recon, self._loss

    def apply(self, grad_scale=1.0):
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().geoadv_trainer_apply(self._h, C.c_float(grad_scale), _lib.stream_handle()), "trainer_apply")

    def partial_fit(self, X, GT=None, want_recon=True, sync=True):
        """AutoEncoder.partial_fit (autoencoder.py:105-125): -> (recon, loss) of the pre-update weights.
        sync=False returns the loss as a 1-element GPU tensor without synchronising (throughput loops)."""
        world = torch.distributed.get_world_size() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 1
        if world > 1 and self.sync_bn:
            recon, loss = self._step_synchronised(self._dev(X), self._dev(GT) if GT is not None else None, want_recon, world)
            return recon, (float(loss.item()) if sync else loss)
        self._set_world(1)
        recon, loss = self.forward_backward(X, GT, want_recon)
        if world > 1:
            from .dist import all_reduce_sum_
            all_reduce_sum_(self.gradient_buffer())                 # one flat buffer: a single collective per step
            loss = all_reduce_sum_(loss.clone()) / world            # loss of the global batch (mean of equal shards)
            self.apply(1.0 / world)
        else:
            self.apply(1.0)
        return recon, (float(loss.item()) if sync else loss)

    def _single_epoch_train(self, train_data, batch_size=None):
        """pointnet_ae.py:101-139 without augmentation / denoising: train_data is an array (n, N, 3) or an object with
        next_batch(batch_size) -> (batch, labels, noisy) and num_examples."""
        bs = self.batch_size
        start = time.time()
        losses = []
        if hasattr(train_data, "next_batch"):
            n_batches = int(train_data.num_examples / bs)
            for _ in range(n_batches):
                batch_i, _, _ = train_data.next_batch(bs)
                losses.append(self.partial_fit(batch_i, want_recon=False, sync=False)[1].clone())
        else:
            n_batches = len(train_data) // bs
            for i in range(n_batches):
                losses.append(self.partial_fit(train_data[i * bs:(i + 1) * bs], want_recon=False, sync=False)[1].clone())
        epoch_loss = float(torch.stack(losses).mean().item()) if losses else 0.0
        return epoch_loss, time.time() - start

    def train(self, train_data, training_epochs, log_file=None, loss_display_step=1):
        """AutoEncoder.train (autoencoder.py:196-227): -> [(epoch, loss, duration)]."""
        stats = []
        for _ in range(training_epochs):
            loss, duration = self._single_epoch_train(train_data)
            self.epoch += 1
            stats.append((self.epoch, loss, duration))
            if self.epoch % loss_display_step == 0 and log_file is not None:
                log_file.write('%04d\t%.9f\t%.4f\n' % (self.epoch, loss, duration / 60.0))
        return stats

    # ---- what saver.save writes ----------------------------------------------------------------------------
    def export_weights(self):
        """Current variables + BN moving averages under their TF names (feed to PointNetAE / AdvAE, save_npz, or
        tf_checkpoint.write_checkpoint)."""
        ed, dd = W.enc_dims(), W.dec_dims(self.n_points)
        canon = {"enc_w": [np.empty((ed[i], ed[i + 1]), np.float32) for i in range(5)],
                 "dec_w": [np.empty((dd[k], dd[k + 1]), np.float32) for k in range(3)],
                 "dec_b": [np.empty(dd[k + 1], np.float32) for k in range(3)]}
        for g in ("enc_b", "gamma", "beta", "mean", "var"):
            canon[g] = [np.empty(ed[i + 1], np.float32) for i in range(5)]
        hw = _AEWeights()
        _fill_weights_struct(hw, canon, self.n_points)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().geoadv_trainer_export(self._h, C.byref(hw), _lib.stream_handle()), "trainer_export")
        out = {}
        for i in range(5):
            p = "%s/encoder_conv_layer_%d" % (self.ae_name, i)
            out[p + "/W"] = canon["enc_w"][i].reshape(1, 1, ed[i], ed[i + 1])
            out[p + "/b"] = canon["enc_b"][i]
            out[p + "_bnorm/gamma"] = canon["gamma"][i]
            out[p + "_bnorm/beta"] = canon["beta"][i]
            out[p + "_bnorm/moving_mean"] = canon["mean"][i]
            out[p + "_bnorm/moving_variance"] = canon["var"][i]
        for k in range(3):
            p = "%s/decoder_fc_%d" % (self.ae_name, k)
            out[p + "/W"] = canon["dec_w"][k]
            out[p + "/b"] = canon["dec_b"][k]
        return out
