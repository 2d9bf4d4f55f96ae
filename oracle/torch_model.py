"""Second, independently written model of the victim auto-encoder and of one attack iteration, on
torch-CPU library ops and autograd -- TEST INFRASTRUCTURE (never imported by geometric_adv_amd).

Why it exists: oracle/attack_model.py (numpy, hand-derived backward) is the oracle of record for the
network rows of SURVEY 8a (a4, a5, a11), and it is UNPINNED against the reference because TensorFlow 1.13 /
tflearn 0.3.2 cannot be installed here.  This file states the same arithmetic a second way -- library
convolution / batch-norm / linear layers and reverse-mode autograd instead of hand-written matrix products
and a hand-written backward -- so that the HIP kernels are not checked against one author's single reading
of tflearn.  tests/test_torch_second_opinion.py asserts the two agree to 1e-10 (fp64).  It does NOT lift
the "parity unpinned" status: only a TF-written vector could.

TF / tflearn pieces restated (file:function of the third-party sources the reference calls into):
  tflearn/layers/conv.py conv_1d              -> F.conv1d(kernel 1) + bias      (encoders_decoders.py:43-44)
  tflearn/layers/normalization.py batch_normalization, is_training False
      -> tf.nn.batch_normalization(x, moving_mean, moving_variance, beta, gamma, 1e-5)
         (tensorflow/python/ops/nn_impl.py batch_normalization: x*inv + (beta - mean*inv), inv = gamma*rsqrt(var+eps))
      -> F.batch_norm(training=False, eps=1e-5)                                   (encoders_decoders.py:52)
  tf.nn.relu / ReluGrad (gradient gated on the OUTPUT > 0)    -> F.relu (its backward gates on output > 0 too)
  tf.reduce_max(axis=1) and math_grad.py _MinOrMaxGrad (equal split among ties) -> torch.amax (same split)
  tflearn fully_connected = matmul(x, W[in,out]) + b          -> F.linear(x, W.T, b) (encoders_decoders.py:107-132)
  tensorflow/core/kernels/training_ops.cc ApplyAdam (non-Nesterov), optimizer defaults of adv_ae.py:152:
      alpha = lr*sqrt(1-beta2^t)/(1-beta1^t); m += (g-m)*(1-beta1); v += (g*g-v)*(1-beta2); var -= m*alpha/(sqrt(v)+eps)

It is also bench.py's BLAS-backed fp32 CPU baseline ("GEMMs via a BLAS-backed fp32 path").
"""
import numpy as np
import torch
import torch.nn.functional as F

from .cpu_oracle import Oracle

_AE = "autoencoder"


class TorchAE:
    """weights: dict keyed by the reference's TF variable names (geometric_adv_amd/weights.py docstring)."""

    def __init__(self, weights, n_points, dtype=torch.float64, ae_name=_AE, impl="conv"):
        """impl: "conv" = F.conv1d on [B, C, N] (the form closest to tflearn's conv_1d; the second-opinion tests use it);
        "mm" = the same layer as one [B*N, Cin] x [Cin, Cout] addmm, which is what runs at BLAS speed on a many-core host
        (bench.py's cpu_baseline: conv1d through oneDNN reached 14 GFLOP/s on 128 threads, addmm an order of magnitude more)."""
        self.n = n_points
        self.dt = dtype
        self.impl = impl
        t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32)).to(dtype)
        self.conv = []
        for i in range(5):
            p = "%s/encoder_conv_layer_%d" % (ae_name, i)
            w = t(weights[p + "/W"])
            w = w.reshape(w.shape[-2], w.shape[-1])                  # [Cin, Cout] of the [1,1,Cin,Cout] conv2d filter
            self.conv.append(dict(w2=w.contiguous(),                 # [Cin, Cout] for addmm
                                  w=w.t().contiguous()[:, :, None],  # conv1d weight [Cout, Cin, 1]
                                  b=t(weights[p + "/b"]), gamma=t(weights[p + "_bnorm/gamma"]), beta=t(weights[p + "_bnorm/beta"]),
                                  mean=t(weights[p + "_bnorm/moving_mean"]), var=t(weights[p + "_bnorm/moving_variance"])))
        self.fc = []
        for k in range(3):
            p = "%s/decoder_fc_%d" % (ae_name, k)
            self.fc.append((t(weights[p + "/W"]).t().contiguous(), t(weights[p + "/b"])))   # F.linear wants [out, in]

    def encode(self, pc):
        if self.impl == "mm":
            b, n = pc.shape[:2]
            h = pc.to(self.dt).reshape(b * n, 3)
            for L in self.conv:
                h = torch.addmm(L["b"], h, L["w2"])
                h = F.relu(F.batch_norm(h, L["mean"], L["var"], L["gamma"], L["beta"], training=False, eps=1e-5))
            return torch.amax(h.reshape(b, n, -1), dim=1)
        h = pc.to(self.dt).transpose(1, 2)                           # [B, 3, N] channels-first for conv1d
        for L in self.conv:
            h = F.conv1d(h, L["w"], L["b"])
            h = F.batch_norm(h, L["mean"], L["var"], L["gamma"], L["beta"], training=False, eps=1e-5)
            h = F.relu(h)
        return torch.amax(h, dim=2)                                  # symmetric max-pool over the points

    def decode(self, z):
        d = F.relu(F.linear(z, *self.fc[0]))
        d = F.relu(F.linear(d, *self.fc[1]))
        return F.linear(d, *self.fc[2]).reshape(z.shape[0], self.n, 3)


def _gather(cloud, idx):
    return torch.gather(cloud, 1, torch.as_tensor(idx, dtype=torch.int64)[:, :, None].expand(-1, -1, 3))


def chamfer_from_matches(p, q, i1, i2):
    """per-cloud mean_j |p_j - q_i1[j]|^2 + mean_k |q_k - p_i2[k]|^2 with the matches held constant (the idx outputs of
    nn_distance get no gradient, tf_nndistance.py:35-41), differentiable in p and q."""
    d1 = ((p - _gather(q, i1)) ** 2).sum(-1)
    d2 = ((q - _gather(p, i2)) ** 2).sum(-1)
    return d1.mean(1) + d2.mean(1), d1


class TorchAttack:
    """One batch of AdvAE's loop (adv_ae.py:78-153, 216-221) with autograd for d loss / d pert."""

    def __init__(self, ae, x, gt, tz, w, loss_adv_type="chamfer", loss_dist_type="chamfer", lr=0.01,
                 max_point_pert_weight=0.0, max_point_dist_weight=0.0, oracle=None):
        self.ae, self.dt = ae, ae.dt
        self.x32, self.gt32 = np.asarray(x, np.float32), np.asarray(gt, np.float32)
        self.x, self.gt = torch.as_tensor(self.x32).to(self.dt), torch.as_tensor(self.gt32).to(self.dt)
        self.tz = None if tz is None else torch.as_tensor(np.asarray(tz)).to(self.dt)
        self.w = torch.as_tensor(np.asarray(w, np.float64)).to(self.dt)
        self.adv_type, self.dist_type, self.lr = loss_adv_type, loss_dist_type, lr
        self.mppw, self.mpdw = max_point_pert_weight, max_point_dist_weight
        self.pert = torch.zeros_like(self.x)
        self.m, self.v = torch.zeros_like(self.x), torch.zeros_like(self.x)
        self.b1p, self.b2p = np.float32(0.9), np.float32(0.999)
        self.o = oracle if oracle is not None else Oracle()

    def init_pert(self, pert):
        self.pert = torch.as_tensor(np.asarray(pert)).to(self.dt).clone()

    def loss(self, pert, idx=None):
        """Returns (total loss, dict).  adv is one fp32 addition (an fp32 tensor in the reference's graph, adversary.py:35);
        the straight-through form keeps d adv / d pert = 1."""
        adv32 = torch.as_tensor(self.x32) + pert.detach().to(torch.float32)
        adv = pert + (adv32.to(self.dt) - pert.detach())
        z = self.ae.encode(adv)
        recon = self.ae.decode(z)
        if idx is None:
            iR1, iR2 = self.o.nn_distance(recon.detach().to(torch.float32).numpy(), self.gt32)[1::2]
            iA1, iA2 = self.o.nn_distance(adv32.numpy(), self.x32)[1::2]
        else:
            iR1, iR2, iA1, iA2 = idx
        loss_ae, _ = chamfer_from_matches(recon, self.gt, iR1, iR2)
        input_dist, A1 = chamfer_from_matches(adv, self.x, iA1, iA2)
        p2 = (pert ** 2).sum(-1)
        loss_pert, loss_max = torch.sqrt(p2.sum(1)), torch.sqrt(p2.amax(1))
        loss_adv = torch.sqrt(((z - self.tz) ** 2).sum(1)) if self.adv_type == "latent" else loss_ae
        if self.dist_type == "pert":
            loss_dist = loss_pert + self.mppw * loss_max if self.mppw > 0 else loss_pert
        else:
            loss_dist = input_dist + self.mpdw * A1.amax(1) if self.mpdw > 0 else input_dist
        total = (loss_adv + self.w * loss_dist).sum()                                       # adv_ae.py:105
        return total, dict(adv=adv, z=z, recon=recon, idx=(iR1, iR2, iA1, iA2), loss_ae=loss_ae, input_dist=input_dist,
                           loss_adv=loss_adv, loss_dist=loss_dist, loss_pert=loss_pert, loss_max=loss_max)

    def gradient(self, idx=None):
        p = self.pert.clone().requires_grad_(True)
        total, f = self.loss(p, idx)
        (g,) = torch.autograd.grad(total, p)
        return g, f

    def adam(self, g):
        f32 = np.float32
        dt = self.dt
        T = lambda v: torch.tensor(float(v), dtype=dt)
        omb1, omb2, eps = T(f32(1) - f32(0.9)), T(f32(1) - f32(0.999)), T(f32(1e-8))
        alpha = T(f32(self.lr) * np.sqrt(f32(1) - self.b2p) / (f32(1) - self.b1p))
        self.m = self.m + (g - self.m) * omb1
        self.v = self.v + (g * g - self.v) * omb2
        self.pert = self.pert - (self.m * alpha) / (torch.sqrt(self.v) + eps)
        rt = lambda t: t.to(torch.float32).to(dt)                     # pert and the slots are fp32 variables in the graph
        self.m, self.v, self.pert = rt(self.m), rt(self.v), rt(self.pert)
        self.b1p = f32(self.b1p * f32(0.9))
        self.b2p = f32(self.b2p * f32(0.999))

    def step(self, idx=None):
        g, f = self.gradient(idx)
        self.adam(g)
        return f, g

    @torch.no_grad()
    def forward(self):
        return self.loss(self.pert)[1]
