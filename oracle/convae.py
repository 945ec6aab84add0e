"""Oracle: CPU restatement of the reference ConvAutoencoder + TDNN sex classifier.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows /root/reference
``models/ConvAutoEncoder.py``:
  * GradReverse                 :12-28
  * TDNNSexClassifier           :30-69   (reshape-not-transpose before pooling, :61)
  * GLU == x*sigmoid(x)         :113-120
  * ConvAutoencoder             :136-200
and speechbrain ``nnet/pooling.py:StatisticsPooling`` (un-vendored dependency, restated
from the published v0.5.x algorithm: mean(dim=1) + gauss-noise offset in eps*[1,9],
unbiased std(dim=1) + eps, eps = 1e-5, concatenated and unsqueezed to [B,1,2C]).

Module / attribute names are the reference's so ``state_dict()`` keys match
(``encoder.0.weight`` ...).  gen_golden.py shows this file equals the reference import
bit-for-bit on CPU when the pooling noise is disabled on both sides.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class GradReverse(torch.autograd.Function):
    """identity forward, grad * -1 backward (ConvAutoEncoder.py:12-28)."""

    @staticmethod
    def forward(ctx, x):
        return x

    @staticmethod
    def backward(ctx, g):
        return -1 * g.clone()


class StatisticsPooling(nn.Module):
    """speechbrain.nnet.pooling.StatisticsPooling (lengths=None branch used by the
    reference, ConvAutoEncoder.py:64).  ``noise``: None -> deterministic form (the
    oracle's definition); "random" -> draw the reference's gaussian-derived offset;
    a tensor [B, C] in [0,1] -> eps*((1-9)*noise+9) is added to the mean (lets a test
    feed the same draw to the HIP path)."""

    def __init__(self, noise=None):
        super().__init__()
        self.eps = 1e-5
        self.noise = noise

    def gauss_noise(self, shape):
        g = torch.randn(shape)
        g = g - torch.min(g)
        g = g / torch.max(g)
        return g

    def forward(self, x, lengths=None):
        if lengths is None:
            mean = x.mean(dim=1)
            std = x.std(dim=1)
        else:
            mean, std = [], []
            for i in range(x.shape[0]):
                n = int(torch.round(lengths[i] * x.shape[1]))
                mean.append(x[i, 0:n].mean(dim=0))
                std.append(x[i, 0:n].std(dim=0))
            mean, std = torch.stack(mean), torch.stack(std)
        if self.noise is not None:
            g = self.gauss_noise(mean.shape) if isinstance(self.noise, str) else self.noise
            mean = mean + self.eps * ((1 - 9) * g + 9)
        std = std + self.eps
        return torch.cat((mean, std), dim=1).unsqueeze(1)


class TDNNSexClassifier(nn.Module):
    def __init__(self, num_classes=2, pooling_noise=None):
        super().__init__()
        self.tdnn = nn.Sequential(
            nn.Conv1d(128, 128, kernel_size=5, dilation=1), nn.ReLU(), nn.BatchNorm1d(128),
            nn.Conv1d(128, 128, kernel_size=3, dilation=2), nn.ReLU(), nn.BatchNorm1d(128),
            nn.Conv1d(128, 128, kernel_size=3, dilation=3), nn.ReLU(), nn.BatchNorm1d(128),
        )
        self.norm = nn.BatchNorm1d(128)
        self.stats_pooling = StatisticsPooling(pooling_noise)
        self.classify = nn.Sequential(
            nn.Linear(256, 128), nn.ReLU(), nn.BatchNorm1d(128),
            nn.Linear(128, 64), nn.ReLU(), nn.BatchNorm1d(64),
            nn.Linear(64, num_classes),
        )

    def forward(self, x):
        x = GradReverse.apply(x)
        x = self.norm(x)
        x = self.tdnn(x)
        # memory reinterpretation, NOT a transpose (ConvAutoEncoder.py:61)
        x = x.reshape(x.shape[0], x.shape[2], x.shape[1])
        p = self.stats_pooling(x).squeeze(1)
        return F.log_softmax(self.classify(p), 1)


class GLU(nn.Module):
    """the reference's "GLU" is swish / SiLU: x*sigmoid(x), no halving (:113-120)."""

    def forward(self, x):
        return x * torch.sigmoid(x)


class ConvAutoencoder(nn.Module):
    def __init__(self, pooling_noise=None):
        super().__init__()
        self.encoder = nn.Sequential(
            nn.Conv1d(1, 32, 15, 1, 7), GLU(),
            nn.Conv1d(32, 64, 5, 2, 2), nn.InstanceNorm1d(64, affine=True), GLU(),
            nn.Conv1d(64, 64, 5, 1, 2), nn.InstanceNorm1d(64, affine=True), GLU(),
            nn.Conv1d(64, 128, 5, 2, 2), nn.InstanceNorm1d(128, affine=True), GLU(),
            nn.Conv1d(128, 128, 5, 1, 2), nn.InstanceNorm1d(128, affine=True), GLU(),
        )
        self.decoder = nn.Sequential(
            nn.Conv1d(128, 128, 5, 1, 2),
            nn.ConvTranspose1d(128, 64, 5, 2, 2, output_padding=1),
            nn.InstanceNorm1d(64, affine=True), GLU(),
            nn.Conv1d(64, 64, 5, 1, 2),
            nn.ConvTranspose1d(64, 32, 5, 2, 2, output_padding=1),
            nn.InstanceNorm1d(32, affine=True), GLU(),
            nn.Conv1d(32, 1, 15, 1, 7),
        )
        self.sex_classifier = TDNNSexClassifier(2, pooling_noise)

    def forward(self, feats, return_latent=False):
        B, T, Fd = feats.shape
        x = feats.reshape(B, T * Fd).unsqueeze(1)
        z = self.encoder(x)
        logp = self.sex_classifier(z)
        y = self.decoder(z).squeeze(1).reshape(B, T, Fd)
        if return_latent:
            return y, logp, z
        return y, logp


def numpy_params(seed=8886):
    """Deterministic, platform-independent parameter set for ConvAutoencoder keyed like
    its state_dict (numpy RandomState is stable across versions; torch initialisers are
    not guaranteed to be).  Scales follow PyTorch's default init magnitudes so the
    activations stay O(1).  Used by gen_golden.py AND the tests, so the fixtures need
    not carry 2 MB of weights."""
    import numpy as np
    rs = np.random.RandomState(seed)
    ref = ConvAutoencoder()
    out = {}
    for k, v in ref.state_dict().items():
        shp = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_mean"):
            out[k] = torch.zeros(shp)
        elif k.endswith("running_var"):
            out[k] = torch.ones(shp)
        elif v.dim() == 1 and k.endswith("weight"):      # norm gammas
            out[k] = torch.from_numpy((1.0 + 0.1 * rs.standard_normal(shp)).astype("float32"))
        elif v.dim() == 1:                                # biases / betas
            out[k] = torch.from_numpy((0.05 * rs.standard_normal(shp)).astype("float32"))
        else:
            fan_in = int(v[0].numel()) if "decoder.1." not in k and "decoder.5." not in k \
                else int(v.shape[0] * v.shape[2])
            out[k] = torch.from_numpy(
                (rs.standard_normal(shp) / (fan_in ** 0.5)).astype("float32"))
    return out
