"""AdvAE: the geometric adversarial attack on a point-cloud auto-encoder (src/adv_ae.py:25-251).

Same constructor / attack() contract as the reference class; the TF graph + session are replaced
by a device-resident loop in libgeoadv.so (geoadv_attack_*): one fused forward per iteration,
sparse encoder backward, CPU-ordered Chamfer gradients, TF-1.13-form Adam on the perturbation.
"""
import ctypes as C
import time

import numpy as np
import torch

from . import _lib
from .adversary import init_pert_value
from .autoencoder import PointNetAE


class Configuration:
    """The fields AdvAE reads from the reference's pickled Configuration (src/autoencoder.py:19-82,
    patched in attacker/run_attack.py:84-107).  Plain attributes; `weights` replaces
    ae_dir/ae_restore_epoch (a dict or .npz keyed by TF variable names, see weights.py)."""

    def __init__(self, batch_size, n_points, weights, loss="chamfer", loss_adv_type="chamfer",
                 loss_dist_type="chamfer", dist_weight_list=(1.0,), max_point_pert_weight=0.0,
                 max_point_dist_weight=0.0, num_iterations=500, num_iterations_thresh=400,
                 learning_rate=0.01, ae_name="autoencoder", emd_weight=0.0, verbose=False, batch_slots=1,
                 chamfer_prune=True, emd_reference_weights=False, recompute_backward=False, separate_adam=False,
                 chamfer_kernel="auto", encoder_backward="auto", emd_dense_levels=False, encoder_arith=None, loss_in_scan=True):
        self.batch_size = int(batch_size)
        self.n_input = [int(n_points), 3]
        self.n_output = [int(n_points), 3]
        self.weights = weights
        self.loss = loss
        self.loss_adv_type = loss_adv_type
        self.loss_dist_type = loss_dist_type
        self.dist_weight_list = list(dist_weight_list)
        self.max_point_pert_weight = float(max_point_pert_weight)
        self.max_point_dist_weight = float(max_point_dist_weight)
        self.num_iterations = int(num_iterations)
        self.num_iterations_thresh = int(num_iterations_thresh)
        self.learning_rate = float(learning_rate)
        self.ae_name = ae_name
        self.emd_weight = float(emd_weight)
        self.verbose = verbose
        self.batch_slots = int(batch_slots)      # batches attacked concurrently on this GPU (AdvAE.attack); 1 = the reference's order
        if chamfer_prune not in (True, False, "always", "pinned"):
            raise ValueError("chamfer_prune must be True (grid search except for tiny batches; adaptive), 'pinned' (the same, never "
                             "switched off by adapt_source_search), False or 'always'")
        self.chamfer_prune = chamfer_prune       # False: nn_distance(adv, x) always by the all-pairs kernel; "always": the paired
                                                 # grid search at every batch size (same results either way); True: the search, which
                                                 # a batch that hands most clouds back switches off FOR THAT BATCH (adapt_source_search);
                                                 # "pinned": True without that policy (timing independent of the data)
        self.emd_reference_weights = bool(emd_reference_weights)   # True: the EMD term's plan from the CPU op's expf arguments (ops.approx_match)
        self.encoder_arith = encoder_arith                         # None (library default: "bf16x3"), "bf16x3" or "f32": autoencoder.ENCODER_ARITH; of the model this handle creates
        self.emd_dense_levels = bool(emd_dense_levels)             # True: this handle's EMD sweeps all dense (GEOADV_EMD_DENSE_LEVELS; per handle, nothing process-wide)
        # alternative code paths with the same results (geoadv_attack_config; the parity tests run each against the default)
        self.recompute_backward = bool(recompute_backward)   # encoder backward re-runs the forward instead of reading ReLU masks
        self.separate_adam = bool(separate_adam)             # Adam step as its own launch
        if chamfer_kernel not in CHAMFER_KERNELS:
            raise ValueError("chamfer_kernel must be one of %s" % sorted(CHAMFER_KERNELS))
        self.chamfer_kernel = chamfer_kernel                 # "auto" (by batch size), "two_scan" or "symmetric"
        if encoder_backward not in ENCODER_BACKWARDS:
            raise ValueError("encoder_backward must be one of %s" % sorted(ENCODER_BACKWARDS))
        self.encoder_backward = encoder_backward             # "auto", "masked" (back-propagate dz) or "jacobian" (pool Jacobian)
        if loss_in_scan not in (True, False, "always"):
            raise ValueError("loss_in_scan must be True (where it pays), False or 'always'")
        self.loss_in_scan = loss_in_scan                     # False: the loss + gradient pass always as a launch of its own; "always": riding wherever possible


CHAMFER_KERNELS = {"auto": 0, "two_scan": 1, "symmetric": 2}
ENCODER_BACKWARDS = {"auto": 0, "masked": 1, "jacobian": 2}


class _AttackConfig(C.Structure):
    _fields_ = [("batch", C.c_int), ("loss_adv_type", C.c_int), ("loss_dist_type", C.c_int),
                ("max_point_pert_weight", C.c_float), ("max_point_dist_weight", C.c_float),
                ("learning_rate", C.c_float), ("emd_weight", C.c_float), ("all_pairs_source_dist", C.c_int),
                ("emd_weight_mode", C.c_int), ("recompute_backward", C.c_int), ("encoder_backward", C.c_int),
                ("separate_adam", C.c_int), ("chamfer_kernel", C.c_int), ("loss_in_scan", C.c_int)]


PROF_NAMES = ["encoder_fwd", "decoder_fwd", "chamfer_fwd", "loss_grad", "decoder_bwd", "encoder_bwd", "adam"]


class AdvAE:
    def __init__(self, adversary_name, configuration, device=None, ae=None):
        c = configuration
        self.configuration = c
        self.name = adversary_name
        if c.loss != "chamfer":
            raise ValueError("loss=%r: AdvAE as shipped only builds with loss='chamfer' (adv_ae.py:124 fails for 'emd')" % c.loss)
        if c.loss_adv_type not in ("chamfer", "latent"):
            raise ValueError("loss_adv_type must be 'latent' or 'chamfer' (run_attack.py:49)")
        if c.loss_dist_type not in ("pert", "chamfer"):
            raise ValueError("loss_dist_type must be 'pert' or 'chamfer' (run_attack.py:50)")
        self.device = torch.device(device if device is not None else "cuda:0")
        self.ae = ae if ae is not None else PointNetAE(c.weights, c.n_input[0], c.ae_name, self.device,
                                                       encoder_arith=getattr(c, "encoder_arith", None))
        self.n = c.n_input[0]
        self.B = c.batch_size
        cfg = _AttackConfig(self.B, 1 if c.loss_adv_type == "latent" else 0, 1 if c.loss_dist_type == "pert" else 0,
                            c.max_point_pert_weight, c.max_point_dist_weight, c.learning_rate, c.emd_weight,
                            {True: 0, "pinned": 0, False: 1, "always": 2}[getattr(c, "chamfer_prune", True)], (1 if getattr(c, "emd_reference_weights", False) else 0) | (0x100 if getattr(c, "emd_dense_levels", False) else 0),
                            1 if getattr(c, "recompute_backward", False) else 0, ENCODER_BACKWARDS[getattr(c, "encoder_backward", "auto")],
                            1 if getattr(c, "separate_adam", False) else 0,
                            CHAMFER_KERNELS[getattr(c, "chamfer_kernel", "auto")], {True: 0, False: 1, "always": 2}[getattr(c, "loss_in_scan", True)])
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().geoadv_attack_create(C.byref(self._h), self.ae.handle, C.byref(cfg)), "attack_create")
        self.last_history = None

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                _lib.lib().geoadv_attack_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    # ---- low-level device API (used by bench.py and the tests) ---------------------------
    def _dev(self, a, shape=None):
        t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, dtype=np.float32))
        t = t.to(self.device, dtype=torch.float32).contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError("expected shape %s, got %s" % (tuple(shape), tuple(t.shape)))
        return t

    def set_inputs(self, source_pc, target_pc, target_latent, dist_weight):
        B, n = self.B, self.n
        x = self._dev(source_pc, (B, n, 3)); gt = self._dev(target_pc, (B, n, 3))
        tz = None if target_latent is None else self._dev(target_latent, (B, self.ae.bneck))
        if tz is not None and self.ae.bneck != 128:             # the kernels' width: the absent channels are zeros on both sides
            tz = torch.cat([tz, torch.zeros((B, 128 - self.ae.bneck), dtype=torch.float32, device=self.device)], dim=1).contiguous()
        w = self._dev(np.ones(B, np.float32) * dist_weight if np.isscalar(dist_weight) else dist_weight, (B,))
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().geoadv_attack_set_inputs(self._h, _lib.ptr(x), _lib.ptr(gt), _lib.ptr(tz), _lib.ptr(w),
                                                           _lib.stream_handle()), "attack_set_inputs")
        # new clouds: an earlier batch's verdict says nothing about these -- the adaptive policy starts over with the search on
        if self.configuration.chamfer_prune is True and not getattr(self, "_search_on", True):
            self.set_source_search(True)

    def init_pert(self, init=None, reset_optimizer=False):
        """Adversary.init_pert (adversary.py:27-28).  Adam's slots are NOT reset by default: the
        reference initialises them once per graph (adv_ae.py:74) and never again."""
        if init is None:
            init = init_pert_value(self.B, self.n)
        p = self._dev(init, (self.B, self.n, 3))
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().geoadv_attack_init_pert(self._h, _lib.ptr(p), int(bool(reset_optimizer)),
                                                          _lib.stream_handle()), "attack_init_pert")

    def search_state(self):
        """(searched, handed_back): whether nn_distance(adv, x) goes through the paired grid search, and how many clouds of the
        batch it currently hands back to the all-pairs kernel (synchronises)."""
        a, b = C.c_int(0), C.c_int(0)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().geoadv_attack_search_state(self._h, C.byref(a), C.byref(b), _lib.stream_handle()), "attack_search_state")
        return bool(a.value), b.value

    def run(self, first_iteration, iterations, thresh, history=None):
        """Enqueue `iterations` attack iterations (no host sync).  history: optional GPU tensor
        [iterations, 6, B] receiving loss_adv, loss_dist, loss_pert, loss_max, input_dist, loss_ae."""
        if history is not None and tuple(history.shape) != (iterations, 6, self.B):
            raise ValueError("history must be of shape (iterations, 6, batch)")
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().geoadv_attack_run(self._h, int(first_iteration), int(iterations), int(thresh),
                                                    _lib.ptr(history), _lib.stream_handle()), "attack_run")

    def get_best(self, target_ae_loss_ref, clouds=True):
        """(metrics [B, 5], best adversarial clouds, their reconstructions) as GPU tensors; clouds=False: the metrics only (the
        two cloud arrays are None and not copied -- geoadv_attack_get_best takes null pointers for them)."""
        B, n = self.B, self.n
        ref = self._dev(target_ae_loss_ref, (B,))
        metrics = torch.empty((B, 5), dtype=torch.float32, device=self.device)
        adv = torch.empty((B, n, 3), dtype=torch.float32, device=self.device) if clouds else None
        recon = torch.empty((B, n, 3), dtype=torch.float32, device=self.device) if clouds else None
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().geoadv_attack_get_best(self._h, _lib.ptr(ref), _lib.ptr(metrics), _lib.ptr(adv),
                                                         _lib.ptr(recon), _lib.stream_handle()), "attack_get_best")
        return metrics, adv, recon

    def set_source_search(self, on):
        """nn_distance(adv, x) from the next forward on: paired grid search (True) or all-pairs kernel (False); same results."""
        _lib.check(_lib.lib().geoadv_attack_set_source_search(self._h, int(bool(on))), "attack_set_source_search")
        self._search_on = bool(on)

    def adapt_source_search(self):
        """Configuration.chamfer_prune=True is a default, not a promise: when the search hands more than half of the batch back to
        the all-pairs kernel (a victim whose perturbations leave the 1/16-box cells -- every trained victim measured so far), its
        workgroups only cost time; switch it off until the next set_inputs (new clouds start with the search on again; results
        are identical either way, so ranks of a sharded run may decide differently).  Configuration(chamfer_prune="pinned")
        disables the policy.  Called where the host synchronises anyway (end of a dist-weight run).  Returns the number of clouds
        handed back, or None if the search is not in use."""
        if self.configuration.chamfer_prune is not True or not getattr(self, "_search_on", True):
            return None
        searched, handed_back = self.search_state()
        if not searched:
            return None
        if 2 * handed_back > self.B:
            self.set_source_search(False)
        return handed_back

    def status(self):
        """Raises GeoAdvError if an in-launch hand-off of the loop timed out since the last set_inputs / init_pert (synchronises)."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().geoadv_attack_status(self._h, _lib.stream_handle()), "attack_status")

    def peek(self):
        """Current device state (test introspection): dict of GPU tensors."""
        B, n, dev = self.B, self.n, self.device
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        i = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
        out = dict(pert=f(B, n, 3), adv=f(B, n, 3), recon=f(B, n, 3), latent=f(B, 128), grad=f(B, n, 3),
                   idx_r1=i(B, n), idx_r2=i(B, n), idx_a1=i(B, n), idx_a2=i(B, n))
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().geoadv_attack_peek(self._h, *[_lib.ptr(out[k]) for k in
                       ("pert", "adv", "recon", "latent", "grad", "idx_r1", "idx_r2", "idx_a1", "idx_a2")],
                       _lib.stream_handle()), "attack_peek")
        if self.ae.bneck != 128:
            out["latent"] = out["latent"][:, :self.ae.bneck].contiguous()
        return out

    def profile(self, classes, stride=1):
        """classes: True/False for all/none, or an iterable of PROF_NAMES to time; stride: time every stride-th launch."""
        _lib.check(_lib.lib().geoadv_attack_profile_stride(self._h, int(stride)), "attack_profile_stride")
        if classes is True:
            mask = -1
        elif not classes:
            mask = 0
        else:
            mask = sum(1 << PROF_NAMES.index(c) for c in classes)
        _lib.check(_lib.lib().geoadv_attack_profile(self._h, int(mask)), "attack_profile")

    def markers(self, enable=True):
        """roctx ranges 'geoadv:<class>' around the launches of every kernel class (rocprofv3 --marker-trace)."""
        _lib.check(_lib.lib().geoadv_attack_markers(self._h, int(bool(enable))), "attack_markers")

    def profile_read(self):
        """{kernel class: (launches, total_ms)} measured with HIP events on the launch stream."""
        out = {}
        for k, name in enumerate(PROF_NAMES):
            n_, ms = C.c_int(0), C.c_float(0)
            _lib.check(_lib.lib().geoadv_attack_profile_read(self._h, k, C.byref(n_), C.byref(ms)), "attack_profile_read")
            out[name] = (n_.value, ms.value)
        return out

    # ---- the reference's public surface ----------------------------------------------------
    def reconstruct(self, X, GT=None, compute_loss=True):
        return self.ae.reconstruct(X, GT, compute_loss)

    def attack(self, source_pc, target_latent, target_pc, target_ae_loss_ref, configuration=None, log_file=None):
        """adv_ae.py:155-189.  Returns (adversarial_metrics [W,n,5], adversarial_pc_input [W,n,N,3],
        adversarial_pc_recon [W,n,N,3]) as numpy arrays, W = len(dist_weight_list)."""
        c = configuration if configuration is not None else self.configuration
        n_examples = len(source_pc)
        batch_size = c.batch_size
        assert n_examples % batch_size == 0, \
            'The number of examples (%d) should be divided by the batch size (%d)' % (n_examples, batch_size)
        n_batches = n_examples // batch_size
        if min(int(getattr(c, "batch_slots", 1)), n_batches) > 1:
            return self._attack_slots(source_pc, target_latent, target_pc, target_ae_loss_ref, n_batches, log_file)
        metrics, pcs_in, pcs_rec = [], [], []
        for i in range(n_batches):
            start_time = time.time()
            s, e = i * batch_size, (i + 1) * batch_size
            m_, a_, r_ = self._attack_one_batch(source_pc[s:e], None if target_latent is None else target_latent[s:e],
                                                target_pc[s:e], target_ae_loss_ref[s:e], log_file)
            metrics.append(m_); pcs_in.append(a_); pcs_rec.append(r_)
            duration = time.time() - start_time
            print("Batch: %04d out of %04d, attack time (minutes): %.4f" % (i + 1, n_batches, duration / 60.0))
            if log_file is not None:
                log_file.write('Batch %04d\tDuration %.4f\n' % (i + 1, duration / 60.0))
        return np.concatenate(metrics, axis=1), np.concatenate(pcs_in, axis=1), np.concatenate(pcs_rec, axis=1)

    def _attack_slots(self, source_pc, target_latent, target_pc, target_ae_loss_ref, n_batches, log_file):
        """The batch loop with `batch_slots` batches in flight on this GPU.  A B = 32 iteration is nine dependent launches and
        leaves the chip part idle between them; independent batches fill those gaps (two slots: +31 % iterations/s on
        MI355X, tools/two_slots.py).  Every slot is one attack handle with its own stream and host thread and takes a
        contiguous run of batches, exactly like one rank of dist.shard_batches -- including that rank's Adam slots, which
        the reference never resets between batches (adv_ae.py:74): results equal a `batch_slots`-process run, bit for bit."""
        import io
        import threading
        from .dist import shard_batches
        c = self.configuration
        bs = c.batch_size
        slots = min(int(c.batch_slots), n_batches)
        if not hasattr(self, "_slot_workers"):
            self._slot_workers = []
        while len(self._slot_workers) < slots - 1:
            self._slot_workers.append(AdvAE(self.name, c, self.device, ae=self.ae))
        workers = [self] + self._slot_workers[:slots - 1]
        results, logs, errors = [None] * n_batches, [None] * n_batches, []
        lock = threading.Lock()

        def work(slot):
            try:
                with torch.cuda.device(self.device), torch.cuda.stream(torch.cuda.Stream(self.device)):
                    for i in shard_batches(n_batches, slot, slots):
                        start_time = time.time()
                        s, e = i * bs, (i + 1) * bs
                        buf = io.StringIO() if log_file is not None else None
                        results[i] = workers[slot]._attack_one_batch(source_pc[s:e], None if target_latent is None else target_latent[s:e],
                                                                     target_pc[s:e], target_ae_loss_ref[s:e], buf)
                        duration = time.time() - start_time
                        with lock:
                            print("Batch: %04d out of %04d, attack time (minutes): %.4f" % (i + 1, n_batches, duration / 60.0))
                        if buf is not None:
                            logs[i] = buf.getvalue() + 'Batch %04d\tDuration %.4f\n' % (i + 1, duration / 60.0)
            except BaseException as exc:     # re-raised in the caller's thread
                errors.append(exc)

        threads = [threading.Thread(target=work, args=(k,)) for k in range(slots)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        if log_file is not None:
            for text in logs:
                log_file.write(text)
        return (np.concatenate([r[0] for r in results], axis=1), np.concatenate([r[1] for r in results], axis=1),
                np.concatenate([r[2] for r in results], axis=1))

    def _attack_one_batch(self, source_pc, target_latent, target_pc, target_ae_loss_ref, log_file=None, init_pert=None):
        """adv_ae.py:191-251."""
        c = self.configuration
        W = len(c.dist_weight_list)
        B, n = self.B, self.n
        metrics_all = np.zeros((W, B, 5), np.float32)
        adv_all = np.zeros((W, B, n, 3), np.float32)
        rec_all = np.zeros((W, B, n, 3), np.float32)
        hist = torch.empty((c.num_iterations, 6, B), dtype=torch.float32, device=self.device)
        self.last_history = []
        for i, dist_weight in enumerate(c.dist_weight_list):
            self.set_inputs(source_pc, target_pc, target_latent, float(dist_weight))
            self.init_pert(init_pert)
            self.run(0, c.num_iterations, c.num_iterations_thresh, hist)
            m_, a_, r_ = self.get_best(target_ae_loss_ref)
            self.status()                                            # the host sync of the run; raises if a hand-off ever timed out
            self.adapt_source_search()
            h = hist.cpu().numpy()
            self.last_history.append(h)
            step = (c.num_iterations // 10) or 1
            for it in range(c.num_iterations):
                if (it + 1) % step == 0:
                    la, ld, lp, lm = (h[it, k].mean() for k in range(4))
                    loss = (h[it, 0] + dist_weight * h[it, 1]).mean()
                    if c.verbose:
                        print("Weight {} of {}, Iteration {} of {}, loss={} loss_adv={} loss_dist={} loss_pert={} loss_max={}"
                              .format(i + 1, W, it + 1, c.num_iterations, loss, la, ld, lp, lm))
                    if log_file is not None:
                        log_file.write('Dist weight %.4f\tIteration %.04d\tloss: %.4f\tloss_adv: %.4f\tloss_dist: %.4f\t'
                                       'loss_pert: %.4f\tloss_max: %.4f\n' % (dist_weight, it + 1, loss, la, ld, lp, lm))
            metrics_all[i] = m_.cpu().numpy(); adv_all[i] = a_.cpu().numpy(); rec_all[i] = r_.cpu().numpy()
        return metrics_all, adv_all, rec_all
