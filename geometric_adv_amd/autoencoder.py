"""Host-side handle of the victim auto-encoder (forward only).

Mirrors what the attack path uses of src/adversary_autoencoder.py (restore_ae_model :42-51,
reconstruct :75-91, get_latent_vectors) and src/autoencoder.py (get_loss_per_pc :150-168): the
weights are uploaded once into libgeoadv.so (geoadv_ae_create packs them for MFMA) and every call
runs the fused gfx950 kernels.  Training lives in trainer.py.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, ops, weights as W


class _AEWeights(C.Structure):
    _fields_ = [("n_points", C.c_int), ("enc_dims", C.c_int * 6), ("dec_dims", C.c_int * 4),
                ("enc_w", C.c_void_p * 5), ("enc_b", C.c_void_p * 5),
                ("bn_gamma", C.c_void_p * 5), ("bn_beta", C.c_void_p * 5),
                ("bn_mean", C.c_void_p * 5), ("bn_var", C.c_void_p * 5),
                ("dec_w", C.c_void_p * 3), ("dec_b", C.c_void_p * 3)]


ENCODER_ARITH = {"f32": 0, "bf16x3": 1, "f16x2": 2}     # GEOADV_ENC_ARITH_*


class PointNetAE:
    """PointNet-style encoder + FC decoder with frozen weights on one GPU."""

    def __init__(self, weights, n_points, ae_name=W.AE_NAME, device=None, encoder_arith=None):
        """weights: dict of TF-variable-name -> array (see weights.py), or a path to such an .npz, or a
        TF V2 checkpoint prefix like '<ae_dir>/models.ckpt-500' (read without TensorFlow, tf_checkpoint.py).
        encoder_arith: "f16x2" (fp32 products as three fp16 piece products of power-of-two-scaled operands on the fp16 matrix pipe,
        range-guarded: include/geoadv.h), "bf16x3" (six bf16 piece products) or "f32" (fp32 MFMA); None = the library default
        ("f16x2" for every model whose constants scale exactly).  Applies to everything that runs this model (forward, attack,
        defense)."""
        if isinstance(weights, str):
            weights = W.load(weights, ae_name)
        self.n_points = int(n_points)
        self.device = torch.device(device if device is not None else "cuda:0")
        self._canon = W.canonical(weights, self.n_points, ae_name, pad_to=W.KERNEL_BNECK)     # keeps the host arrays alive
        self.bneck = W.bneck_of(weights, ae_name)     # the model's own bottleneck size (<= 128: narrower ones run zero-padded)
        self._kb = W.KERNEL_BNECK
        hw = _AEWeights()
        hw.n_points = self.n_points
        hw.enc_dims[:] = W.enc_dims()
        hw.dec_dims[:] = W.dec_dims(self.n_points)
        for i in range(5):
            hw.enc_w[i] = self._canon["enc_w"][i].ctypes.data
            hw.enc_b[i] = self._canon["enc_b"][i].ctypes.data
            hw.bn_gamma[i] = self._canon["gamma"][i].ctypes.data
            hw.bn_beta[i] = self._canon["beta"][i].ctypes.data
            hw.bn_mean[i] = self._canon["mean"][i].ctypes.data
            hw.bn_var[i] = self._canon["var"][i].ctypes.data
        for k in range(3):
            hw.dec_w[k] = self._canon["dec_w"][k].ctypes.data
            hw.dec_b[k] = self._canon["dec_b"][k].ctypes.data
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().geoadv_ae_create(C.byref(self._h), C.byref(hw)), "ae_create")
        if encoder_arith is not None:
            self.set_encoder_arith(encoder_arith)
        self._ws = None

    def set_encoder_arith(self, arith):
        """Switch the encoder kernels' arithmetic (ENCODER_ARITH); not while another thread uses this model."""
        _lib.check(_lib.lib().geoadv_ae_set_encoder_arith(self._h, ENCODER_ARITH[arith]), "ae_set_encoder_arith")

    @property
    def encoder_arith(self):
        code = _lib.lib().geoadv_ae_encoder_arith(self._h)
        return next(k for k, v in ENCODER_ARITH.items() if v == code)

    def status(self):
        """Synchronises and raises if an "f16x2" forward of this model met an activation outside its range since the last call
        (geoadv_ae_status: the clouds concerned got +inf latents).  The numpy-returning methods call it."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().geoadv_ae_status(self._h, _lib.stream_handle()), "ae_status")

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                _lib.lib().geoadv_ae_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def _as_dev(self, a):
        t = torch.as_tensor(np.asarray(a, dtype=np.float32)) if not isinstance(a, torch.Tensor) else a
        t = t.to(self.device, dtype=torch.float32).contiguous()
        if t.dim() != 3 or t.shape[1] != self.n_points or t.shape[2] != 3:
            raise ValueError("point clouds must be of shape (batch, %d, 3); got %s" % (self.n_points, tuple(t.shape)))
        return t

    def forward(self, pc, want_recon=True):
        """pc (b,n,3) -> (recon (b,n,3) or None, latent (b,bneck)) as GPU tensors."""
        pc = self._as_dev(pc)
        b = pc.shape[0]
        latent = torch.empty((b, self._kb), dtype=torch.float32, device=self.device)
        recon = torch.empty((b, self.n_points, 3), dtype=torch.float32, device=self.device) if want_recon else None
        with torch.cuda.device(self.device):
            need = _lib.lib().geoadv_ae_workspace_bytes(self._h, b)
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
            st = _lib.lib().geoadv_ae_forward(self._h, b, _lib.ptr(pc), _lib.ptr(latent), _lib.ptr(recon),
                                              _lib.ptr(self._ws), _lib.stream_handle())
        _lib.check(st, "ae_forward")
        return recon, (latent if self.bneck == self._kb else latent[:, :self.bneck].contiguous())

    def max_and_argmax(self, pc):
        """(max_val (b,128), max_idx (b,128) int32) of the last encoder layer over the points of each cloud:
        what src/ae_utils.py:19-20 computes with np.max / np.argmax from get_pre_symmetry_data
        (autoencoder.py:309-319), without ever materialising the (b, n, 128) pre-symmetry tensor."""
        pc = self._as_dev(pc)
        b = pc.shape[0]
        latent = torch.empty((b, self._kb), dtype=torch.float32, device=self.device)
        idx = torch.empty((b, self._kb), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            need = _lib.lib().geoadv_ae_workspace_bytes(self._h, b)
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
            st = _lib.lib().geoadv_ae_critical(self._h, b, _lib.ptr(pc), _lib.ptr(latent), _lib.ptr(idx),
                                               _lib.ptr(self._ws), _lib.stream_handle())
        _lib.check(st, "ae_critical")
        if self.bneck != self._kb:
            latent, idx = latent[:, :self.bneck].contiguous(), idx[:, :self.bneck].contiguous()
        return latent, idx

    # -- the reference's method names -----------------------------------------------------
    def reconstruct(self, X, GT=None, compute_loss=True):
        """adversary_autoencoder.py:75-91: returns (reconstructions, mean Chamfer loss or None) as numpy."""
        recon, _ = self.forward(X)
        loss = None
        if compute_loss:
            gt = self._as_dev(X if GT is None else GT)
            loss = float(self.loss_per_pc_tensor(recon, gt).mean().item())
        self.status()
        return recon.cpu().numpy(), loss

    def transform(self, X):
        """Latent codes (autoencoder.py: transform / get_latent_vectors) as numpy."""
        _, z = self.forward(X, want_recon=False)
        self.status()
        return z.cpu().numpy()

    get_latent_vectors = transform

    def decode(self, z):
        """autoencoder.py:191-194: latent codes (k,128) or one code (128,) -> reconstructions (k,n,3) as numpy.  Decoder half of
        the fused forward through geoadv_ae_decode: decode(transform(X)) == reconstruct(X)[0] bit for bit."""
        return self.decode_tensor(z).cpu().numpy()

    def decode_tensor(self, z):
        z = torch.as_tensor(np.asarray(z, dtype=np.float32)) if not isinstance(z, torch.Tensor) else z
        z = z.to(self.device, dtype=torch.float32)
        if z.dim() == 1:                                          # single example
            z = z[None]
        z = z.contiguous()
        if z.dim() != 2 or z.shape[1] != self.bneck:
            raise ValueError("latent codes must be of shape (batch, %d); got %s" % (self.bneck, tuple(z.shape)))
        b = z.shape[0]
        if self.bneck != self._kb:                                # the absent channels are zeros (and meet zero decoder rows)
            z = torch.cat([z, torch.zeros((b, self._kb - self.bneck), dtype=torch.float32, device=self.device)], dim=1).contiguous()
        recon = torch.empty((b, self.n_points, 3), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            need = _lib.lib().geoadv_ae_workspace_bytes(self._h, b)
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
            st = _lib.lib().geoadv_ae_decode(self._h, b, _lib.ptr(z), _lib.ptr(recon), _lib.ptr(self._ws), _lib.stream_handle())
        _lib.check(st, "ae_decode")
        return recon

    def interpolate(self, x, y, steps):
        """autoencoder.py:178-189: decode `steps + 2` codes on the segment between the codes of clouds x and y (n,3).
        The reference forms the codes in a float64 numpy array and feeds them to the float32 placeholder; so does this."""
        z1, z2 = self.transform(np.stack([np.asarray(x, np.float32), np.asarray(y, np.float32)]))
        all_z = np.zeros((steps + 2, len(z1)))
        for i, alpha in enumerate(np.linspace(0, 1, steps + 2)):
            all_z[i, :] = (alpha * z2) + ((1.0 - alpha) * z1)
        return self.decode(all_z.astype(np.float32))

    def get_reconstructions(self, pclouds, batch_size=50):
        """autoencoder.py:296-307: reconstructions of (N,K,3) clouds, fed `batch_size` at a time like the reference (the result
        does not depend on the chunking: every cloud is reconstructed on its own)."""
        out = [self.forward(pclouds[s:s + batch_size])[0].cpu().numpy() for s in range(0, len(pclouds), batch_size)]
        self.status()
        return np.vstack(out)

    def get_loss(self, X, GT=None):
        """autoencoder.py:140-148 with the loss of pointnet_ae.py:75-79: reduce_mean(dist1) + reduce_mean(dist2) of
        nn_distance(reconstruct(X), GT or X) over the whole batch, a python float."""
        recon, _ = self.forward(X)
        d1, _, d2, _ = ops.nn_distance(recon, self._as_dev(X if GT is None else GT))
        self.status()
        return float((d1.mean() + d2.mean()).item())

    def gradient_of_input_wrt_loss(self, in_points, gt_points=None):
        """pointnet_ae.py:140-143: tf.gradients(self.loss, self.x) -- d [reduce_mean(dist1) + reduce_mean(dist2)] / d in_points,
        a list with one (b,n,3) array.  Evaluated by the attack loop's own backward (one iteration at learning rate 0 with a
        zero perturbation and zero distance weight: its gradient of sum_b loss_ae[b], divided by the batch)."""
        from .adv_ae import AdvAE, Configuration
        x = self._as_dev(in_points)
        gt = x if gt_points is None else self._as_dev(gt_points)
        b = x.shape[0]
        if not hasattr(self, "_grad_handles"):
            self._grad_handles = {}
        at = self._grad_handles.get(b)
        if at is None:
            at = AdvAE("gradient", Configuration(batch_size=b, n_points=self.n_points, weights=None, learning_rate=0.0,
                                                 num_iterations=1, num_iterations_thresh=2), device=self.device, ae=self)
            self._grad_handles[b] = at
        at.set_inputs(x, gt, None, 0.0)
        at.init_pert(torch.zeros_like(x), reset_optimizer=True)
        at.run(0, 1, 2)
        return [(at.peek()["grad"] / float(b)).cpu().numpy()]

    def loss_per_pc_tensor(self, recon, gt):
        d1, _, d2, _ = ops.nn_distance(recon, gt)
        return d1.mean(1) + d2.mean(1)                      # pointnet_ae.py:75-79 / adv_ae.py:120-121

    def get_loss_per_pc(self, feed_data, orig_data=None):
        """autoencoder.py:150-168: per-cloud Chamfer reconstruction error as numpy."""
        recon, _ = self.forward(feed_data)
        gt = self._as_dev(feed_data if orig_data is None else orig_data)
        self.status()
        return self.loss_per_pc_tensor(recon, gt).cpu().numpy()
