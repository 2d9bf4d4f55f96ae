"""numpy model of one TRAINING step of the victim auto-encoder -- TEST INFRASTRUCTURE.

Restates `PointNetAutoEncoder` (src/pointnet_ae.py:71-99: loss = reduce_mean(dist1) + reduce_mean(dist2) of
nn_distance(x_reconstr, gt), or with conf.loss == 'emd' reduce_mean(match_cost(x_reconstr, gt, approx_match(x_reconstr, gt)));
AdamOptimizer(lr).minimize(loss) over every trainable variable) and
`AutoEncoder.partial_fit` (src/autoencoder.py:105-125: one sess.run of (train_step, loss, x_reconstr) with
tflearn's is_training(True)), with the architecture of src/ae_templates.py:22-33 and the defaults of
default_train_params (:43-51: batch 50, lr 0.0005).

Third-party semantics restated (tflearn==0.3.2 `batch_normalization`, tensorflow-gpu==1.13.2; neither installed):
  training branch: mean, var = tf.nn.moments(a, axes=[0, 1])   (population variance of the B*N rows)
                   moving_x  -= (moving_x - batch_x) * (1 - decay), decay = 0.9, zero_debias=False
                   y = tf.nn.batch_normalization(a, mean, var, beta, gamma, 1e-5)
                     = a * inv + (beta - mean * inv),  inv = gamma * rsqrt(var + 1e-5)
  gradients flow through mean and var (tf.nn.moments is differentiable);
  Adam = TF 1.13 ApplyAdam (see attack_model.py), one shared pair of beta powers.
Parity status: UNPINNED against the reference (same reason as attack_model.py); self-checked against finite
differences in tests/test_host_logic.py.  The Chamfer pieces are the pinned C restatement.
"""
import numpy as np

from .attack_model import _o

EPS = 1e-5
PARAM_GROUPS = ("enc_w", "enc_b", "gamma", "beta", "dec_w", "dec_b")      # trainable; mean/var are moving averages


class TrainModel:
    def __init__(self, canon, n_points, lr=0.0005, decay=0.9, dtype=np.float64, loss="chamfer"):
        assert loss in ("chamfer", "emd")                          # conf.loss, src/pointnet_ae.py:74-79
        self.n, self.lr, self.decay, self.dt, self.loss = n_points, lr, decay, dtype, loss
        self.p = {k: [np.asarray(a, dtype=dtype).copy() for a in canon[k]] for k in canon}
        self.m = {k: [np.zeros_like(a) for a in self.p[k]] for k in PARAM_GROUPS}
        self.v = {k: [np.zeros_like(a) for a in self.p[k]] for k in PARAM_GROUPS}
        self.b1p, self.b2p = np.float32(0.9), np.float32(0.999)

    # ---- forward in training mode; returns everything the backward needs -----------------------------
    def forward(self, x):
        P, dt = self.p, self.dt
        B, N, _ = x.shape
        h = np.asarray(x, dt).reshape(B * N, 3)
        cache = {"h": [h], "a": [], "xhat": [], "inv_std": [], "mu": [], "var": []}
        for i in range(5):
            a = h @ P["enc_w"][i] + P["enc_b"][i]
            mu = a.mean(axis=0)
            var = ((a - mu) ** 2).mean(axis=0)                 # tf.nn.moments: mean of squared differences
            inv_std = 1.0 / np.sqrt(var + dt(EPS))
            xhat = (a - mu) * inv_std
            h = np.maximum(xhat * P["gamma"][i] + P["beta"][i], 0)
            cache["a"].append(a); cache["mu"].append(mu); cache["var"].append(var)
            cache["inv_std"].append(inv_std); cache["xhat"].append(xhat); cache["h"].append(h)
        h5 = h.reshape(B, N, -1)
        z = h5.max(axis=1)
        d1 = np.maximum(z @ P["dec_w"][0] + P["dec_b"][0], 0)
        d2 = np.maximum(d1 @ P["dec_w"][1] + P["dec_b"][1], 0)
        recon = (d2 @ P["dec_w"][2] + P["dec_b"][2]).reshape(B, N, 3)
        cache.update(z=z, h5=h5, d1=d1, d2=d2, recon=recon)
        return cache

    @staticmethod
    def chamfer_loss_fixed(recon, gt, idx1, idx2):
        """loss with the matches pinned (smooth; used by the finite-difference check and, with the matches of the
        fp32 search, as the loss itself): reduce_mean over all B*N elements of each direction."""
        B = recon.shape[0]
        ar = np.arange(B)[:, None]
        d1 = ((recon - gt[ar, idx1]) ** 2).sum(-1)
        d2 = ((gt - recon[ar, idx2]) ** 2).sum(-1)
        return d1.mean() + d2.mean()

    def loss_and_grads(self, x, gt=None, idx=None):
        """-> (loss, grads dict like self.p without mean/var, cache).  idx = (idx1, idx2) pins the matches."""
        P, dt = self.p, self.dt
        gt = np.asarray(x if gt is None else gt, np.float32)
        B, N, _ = gt.shape
        c = self.forward(x)
        recon = c["recon"]
        if self.loss == "emd":
            # src/pointnet_ae.py:77-79: match = approx_match(x_reconstr, gt); loss = reduce_mean(match_cost(x_reconstr, gt, match)).
            # approx_match is NoGradient (external/structural_losses/tf_approxmatch.py:19) and MatchCost's registered gradient is
            # match_cost_grad times the upstream gradient (:44-50): d loss / d recon = grad1 / B with the match held constant.
            # All three are the PINNED C restatements of the reference's CPU ops (oracle/geoadv_oracle.c), fed the fp32 clouds.
            r32 = recon.astype(np.float32)
            match = _o().approx_match(r32, gt)
            loss = float(_o().match_cost(r32, gt, match).astype(np.float64).mean())
            g1, _ = _o().match_cost_grad(r32, gt, match)
            g = g1.astype(dt) / B
        else:
            if idx is None:
                _, i1, _, i2 = _o().nn_distance(recon.astype(np.float32), gt)
                idx = (i1.astype(np.int64), i2.astype(np.int64))
            i1, i2 = idx
            gt64 = gt.astype(dt)
            loss = self.chamfer_loss_fixed(recon, gt64, i1, i2)
            ar = np.arange(B)[:, None]
            # NnDistanceGrad w.r.t. xyz1 = recon with grad_dist = 1/(B*N) (tf_nndistance.cpp:130-163)
            g = 2.0 / (B * N) * (recon - gt64[ar, i1])
            t2 = 2.0 / (B * N) * (gt64 - recon[ar, i2])              # scattered with a minus sign onto recon[idx2]
            for b in range(B):
                np.subtract.at(g[b], i2[b], t2[b])
        G = {k: [None] * len(P[k]) for k in PARAM_GROUPS}
        g = g.reshape(B, 3 * N)
        G["dec_w"][2] = c["d2"].T @ g; G["dec_b"][2] = g.sum(0)
        dd2 = (g @ P["dec_w"][2].T) * (c["d2"] > 0)
        G["dec_w"][1] = c["d1"].T @ dd2; G["dec_b"][1] = dd2.sum(0)
        dd1 = (dd2 @ P["dec_w"][1].T) * (c["d1"] > 0)
        G["dec_w"][0] = c["z"].T @ dd1; G["dec_b"][0] = dd1.sum(0)
        dz = dd1 @ P["dec_w"][0].T
        ind = (c["h5"] == c["z"][:, None, :])                    # _MinOrMaxGrad: equal split among ties
        dh = (ind / ind.sum(axis=1, keepdims=True) * dz[:, None, :]).reshape(B * N, -1)
        R = B * N
        for i in range(4, -1, -1):
            dy = dh * (c["h"][i + 1] > 0)                        # ReluGrad on the output
            xhat = c["xhat"][i]
            G["beta"][i] = dy.sum(0)
            G["gamma"][i] = (dy * xhat).sum(0)
            da = P["gamma"][i] * c["inv_std"][i] * (dy - G["beta"][i] / R - xhat * (G["gamma"][i] / R))
            G["enc_w"][i] = c["h"][i].T @ da
            G["enc_b"][i] = da.sum(0)
            dh = da @ P["enc_w"][i].T
        return loss, G, c

    def step(self, x, gt=None):
        """partial_fit: -> (loss, recon) of the PRE-update weights; updates weights, Adam slots, moving averages."""
        loss, G, c = self.loss_and_grads(x, gt)
        dt = self.dt
        one = dt(1.0)
        alpha = dt(self.lr) * np.sqrt(one - dt(self.b2p)) / (one - dt(self.b1p))
        for k in PARAM_GROUPS:
            for j in range(len(self.p[k])):
                g = G[k][j]
                self.m[k][j] += (g - self.m[k][j]) * dt(np.float32(1) - np.float32(0.9))
                self.v[k][j] += (g * g - self.v[k][j]) * dt(np.float32(1) - np.float32(0.999))
                self.p[k][j] -= (self.m[k][j] * alpha) / (np.sqrt(self.v[k][j]) + dt(np.float32(1e-8)))
        self.b1p = np.float32(self.b1p * np.float32(0.9))
        self.b2p = np.float32(self.b2p * np.float32(0.999))
        for i in range(5):                                       # assign_moving_average(zero_debias=False)
            self.p["mean"][i] -= (self.p["mean"][i] - c["mu"][i]) * dt(1.0 - self.decay)
            self.p["var"][i] -= (self.p["var"][i] - c["var"][i]) * dt(1.0 - self.decay)
        return loss, c["recon"]
