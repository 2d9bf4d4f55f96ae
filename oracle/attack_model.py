"""numpy model of the victim auto-encoder and of one attack iteration -- TEST INFRASTRUCTURE.

Restates SURVEY.md Appendix A (derived from src/adv_ae.py:36-48,78-153, src/adversary.py:33-57,
src/encoders_decoders.py:37-72,100-132 plus the tflearn 0.3.2 / TF 1.13 semantics the reference
relies on: conv_1d(k=1) = x@W + b, batch_normalization inference branch with eps 1e-5,
reduce_max and its equal-split gradient, ReluGrad on the output, ApplyAdam).

Parity status of THIS part: UNPINNED against the reference -- TensorFlow/tflearn are not installed
in the build container, the reference ships no trained weights and no recorded outputs for the
network, so there is nothing to pin it to.  It is the oracle of record for the
network arithmetic; the Chamfer pieces it calls (oracle/geoadv_oracle.c) ARE pinned.
"""
import numpy as np

from .cpu_oracle import Oracle

_oracle = None


def _o():
    global _oracle
    if _oracle is None:
        _oracle = Oracle()
    return _oracle


class AEModel:
    def __init__(self, canon, n_points, dtype=np.float64):
        """canon: geometric_adv_amd.weights.canonical(...) dict."""
        self.n = n_points
        self.dt = dtype
        c = lambda a: np.asarray(a, dtype=dtype)
        self.W = [c(a) for a in canon["enc_w"]]
        self.b = [c(a) for a in canon["enc_b"]]
        self.scale, self.offset = [], []
        for g, be, mu, var in zip(canon["gamma"], canon["beta"], canon["mean"], canon["var"]):
            inv = c(g) / np.sqrt(c(var) + dtype(1e-5))          # gamma * rsqrt(var + eps)
            self.scale.append(inv)
            self.offset.append(c(be) - c(mu) * inv)
        self.V = [c(a) for a in canon["dec_w"]]
        self.c = [c(a) for a in canon["dec_b"]]

    # ---- forward ------------------------------------------------------------------------
    def encode(self, pc, keep=False):
        h = np.asarray(pc, dtype=self.dt)                        # [B,N,3]
        hs = []
        for W, b, s, o in zip(self.W, self.b, self.scale, self.offset):
            a = h @ W + b
            h = np.maximum(a * s + o, 0)
            hs.append(h)
        z = h.max(axis=1)                                        # encoders_decoders.py:72
        return (z, hs) if keep else z

    def decode(self, z, keep=False):
        d1 = np.maximum(z @ self.V[0] + self.c[0], 0)
        d2 = np.maximum(d1 @ self.V[1] + self.c[1], 0)
        out = d2 @ self.V[2] + self.c[2]
        recon = out.reshape(z.shape[0], self.n, 3)               # adv_ae.py:48
        return (recon, d1, d2) if keep else recon

    def reconstruct(self, pc):
        z = self.encode(pc)
        return self.decode(z), z

    # ---- backward to the input ------------------------------------------------------------
    def decoder_backward(self, g_recon, d1, d2):
        g = g_recon.reshape(g_recon.shape[0], -1).astype(self.dt)
        dd2 = (g @ self.V[2].T) * (d2 > 0)
        dd1 = (dd2 @ self.V[1].T) * (d1 > 0)
        return dd1 @ self.V[0].T

    def encoder_backward(self, dz, z, hs):
        h5 = hs[-1]
        ind = (h5 == z[:, None, :])                               # TF _MinOrMaxGrad: equal split among ties
        cnt = ind.sum(axis=1, keepdims=True)
        dh = ind / cnt * dz[:, None, :]
        for i in range(4, -1, -1):
            da = dh * (hs[i] > 0) * self.scale[i]                 # ReluGrad uses output > 0
            dh = da @ self.W[i].T
        return dh                                                 # [B,N,3]


def chamfer_per_pc(d1, d2):
    """adv_ae.py:121 / :132: mean over points of both directions (squared distances)."""
    return d1.mean(axis=1, dtype=np.float64) + d2.mean(axis=1, dtype=np.float64)


class AttackModel:
    """One batch of the attack in numpy; state layout as the reference's graph variables."""

    def __init__(self, model, x, gt, tz, w, loss_adv_type="chamfer", loss_dist_type="chamfer",
                 lr=0.01, max_point_pert_weight=0.0, max_point_dist_weight=0.0, fp32_state=True,
                 emd_weight=0.0):
        self.emd_weight = emd_weight          # build-defined: loss_adv += emd_weight * match_cost(recon, gt) / N
        self.fp32_state = fp32_state          # False: pure fp64 (finite-difference checks of the backward)
        self.m_ = model
        self.dt = model.dt
        self.x = np.asarray(x, np.float32)
        self.gt = np.asarray(gt, np.float32)
        self.tz = None if tz is None else np.asarray(tz, self.dt)
        self.w = np.asarray(w, self.dt)
        self.adv_type, self.dist_type = loss_adv_type, loss_dist_type
        self.lr = lr
        self.mppw, self.mpdw = max_point_pert_weight, max_point_dist_weight
        self.pert = np.zeros_like(self.x, dtype=self.dt)
        self.m = np.zeros_like(self.pert)
        self.v = np.zeros_like(self.pert)
        self.b1p, self.b2p = np.float32(0.9), np.float32(0.999)

    def init_pert(self, pert):
        self.pert = np.asarray(pert, self.dt).copy()

    def forward(self, idx_override=None):
        """Returns a dict of everything one sess.run of the metrics list yields (adv_ae.py:219-221)
        plus intermediates.  idx_override = (iR1,iR2,iA1,iA2) pins the matches (so that a gradient
        check is not derailed by a legitimately flipped near-tie)."""
        M = self.m_
        # adv is an fp32 tensor in the reference (placeholder x + variable pert, adversary.py:35): form it
        # with ONE fp32 addition; with pert ~ 1e-7 against coordinates ~ 0.3 that rounding is a 10 % effect
        # on (adv - x), which an fp64 sum would hide
        if self.fp32_state:
            adv32 = (self.x + self.pert.astype(np.float32)).astype(np.float32)
            adv = adv32.astype(self.dt)
        else:
            adv = self.x.astype(self.dt) + self.pert
            adv32 = adv.astype(np.float32)
        z, hs = M.encode(adv, keep=True)
        recon, d1, d2 = M.decode(z, keep=True)
        recon32 = recon.astype(np.float32)
        R1, iR1, R2, iR2 = _o().nn_distance(recon32, self.gt)
        A1, iA1, A2, iA2 = _o().nn_distance(adv32, self.x)
        if idx_override is not None:
            iR1, iR2, iA1, iA2 = idx_override
        B = self.x.shape[0]
        ar = np.arange(B)[:, None]
        # distances in the model's dtype from the (possibly pinned) matches
        R1 = ((recon - self.gt[ar, iR1]) ** 2).sum(-1); R2 = ((self.gt - recon[ar, iR2]) ** 2).sum(-1)
        A1 = ((adv - self.x[ar, iA1]) ** 2).sum(-1);    A2 = ((self.x - adv[ar, iA2]) ** 2).sum(-1)
        loss_ae = R1.mean(1) + R2.mean(1)
        input_dist = A1.mean(1) + A2.mean(1)
        max_dist = A1.max(1)
        p2 = (self.pert ** 2).sum(-1)
        loss_pert, loss_max = np.sqrt(p2.sum(1)), np.sqrt(p2.max(1))
        match = None
        if self.adv_type == "latent":
            loss_adv = np.sqrt(((z - self.tz) ** 2).sum(1))
        else:
            loss_adv = loss_ae
            if self.emd_weight > 0:
                match = _o().approx_match(recon32, self.gt)                 # (b, n, m), no gradient (tf_approxmatch.py:19)
                cost = _o().match_cost(recon32, self.gt, match)
                loss_adv = loss_ae + self.emd_weight * cost.astype(self.dt) / self.x.shape[1]
        if self.dist_type == "pert":
            loss_dist = loss_pert + (self.mppw * loss_max if self.mppw > 0 else 0)
        else:
            loss_dist = input_dist + (self.mpdw * max_dist if self.mpdw > 0 else 0)
        return dict(match=match, adv=adv, recon=recon, z=z, hs=hs, d1=d1, d2=d2, idx=(iR1, iR2, iA1, iA2), A1=A1,
                    loss_ae=loss_ae, input_dist=input_dist, max_dist=max_dist, loss_pert=loss_pert,
                    loss_max=loss_max, loss_adv=loss_adv, loss_dist=loss_dist, p2=p2)

    def _chamfer_grad_first(self, P, Q, i1, i2, gd1, gd2):
        """d/dP of sum_j gd1*|P_j - Q_i1[j]|^2 + sum_k gd2*|Q_k - P_i2[k]|^2 (tf_nndistance.cpp:130-163)."""
        B = P.shape[0]
        ar = np.arange(B)[:, None]
        g = 2 * gd1[:, None, None] * (P - Q[ar, i1])
        t = 2 * gd2[:, None, None] * (Q - P[ar, i2])               # subtracted from P[i2[k]]
        for b in range(B):
            np.subtract.at(g[b], i2[b], t[b])
        return g

    def gradient(self, f):
        M = self.m_
        B, N = self.x.shape[:2]
        iR1, iR2, iA1, iA2 = f["idx"]
        one = np.ones(B, self.dt)
        if self.adv_type == "latent":
            dz = (f["z"] - self.tz) / f["loss_adv"][:, None]
        else:
            g_recon = self._chamfer_grad_first(f["recon"], self.gt.astype(self.dt), iR1, iR2, one / N, one / N)
            if self.emd_weight > 0:
                g1, _ = _o().match_cost_grad(f["recon"].astype(np.float32), self.gt, f["match"])
                g_recon = g_recon + (self.emd_weight / N) * g1.astype(self.dt)
            dz = M.decoder_backward(g_recon, f["d1"], f["d2"])
        g = M.encoder_backward(dz, f["z"], f["hs"])
        if self.dist_type == "pert":
            gd = self.w[:, None, None] * self.pert / f["loss_pert"][:, None, None]
            if self.mppw > 0:
                j = f["p2"].argmax(1)
                ar = np.arange(B)
                gd[ar, j] += (self.w * self.mppw)[:, None] * self.pert[ar, j] / f["loss_max"][:, None]
        else:
            gd = self._chamfer_grad_first(f["adv"], self.x.astype(self.dt), iA1, iA2, self.w / N, self.w / N)
            if self.mpdw > 0:
                j = f["A1"].argmax(1)
                ar = np.arange(B)
                gd[ar, j] += 2 * (self.w * self.mpdw)[:, None] * (f["adv"][ar, j] - self.x[ar, iA1[ar, j]])
        return g + gd

    def adam(self, g):
        """TF 1.13 ApplyAdam with the optimizer defaults of adv_ae.py:152."""
        dt = self.dt
        # the scalar coefficients are fp32 quantities in TF (T(1) - beta1 etc. with T = float)
        f32 = np.float32
        omb1, omb2, eps = dt(f32(1) - f32(0.9)), dt(f32(1) - f32(0.999)), dt(f32(1e-8))
        alpha = dt(f32(self.lr) * np.sqrt(f32(1) - self.b2p) / (f32(1) - self.b1p))
        self.m += (g - self.m) * omb1
        self.v += (g * g - self.v) * omb2
        self.pert -= (self.m * alpha) / (np.sqrt(self.v) + eps)
        # pert and the Adam slots are fp32 variables in the reference graph
        if self.fp32_state:
            self.m = self.m.astype(f32).astype(dt); self.v = self.v.astype(f32).astype(dt)
            self.pert = self.pert.astype(f32).astype(dt)
        self.b1p = np.float32(self.b1p * np.float32(0.9))
        self.b2p = np.float32(self.b2p * np.float32(0.999))

    def step(self):
        f = self.forward()
        g = self.gradient(f)
        self.adam(g)
        return f, g
