"""x-vector gender classifier, forward only, on libsa_hip.so: drop-in for the
``embedding_model`` / ``classifier`` pair of speechbrain_configs/evaluator_inference.yaml:34-48
(``speechbrain.lobes.models.Xvector.Xvector`` / ``.Classifier``, restated in the reference at
models/external_gender_classifiers.py:24-183).  Same constructor defaults as that config, same
parameter names (``blocks.0.conv.weight`` ... ``blocks.16.w.weight``; ``norm.norm.weight``,
``DNN.block_0.linear.w.weight``, ``out.w.weight`` as in the reference's classifier.ckpt), eval
mode (BatchNorm running statistics).  ``classify_batch_feats(feats, lens)`` is the call the
reference's fork adds (speechbrain_convae_train.py:139,146): returns (log_probs, score, index).
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L
from . import ops


class _Conv(nn.Module):
    def __init__(self, cin, cout, k, d):
        super().__init__()
        self.kernel_size, self.dilation = k, d
        self.conv = nn.Conv1d(cin, cout, k, dilation=d)


class _BN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.norm = nn.BatchNorm1d(c)

    def affine(self):
        n = self.norm
        return ops.fin_bn_eval(n.num_features, n.weight, n.bias, n.running_mean, n.running_var, n.eps)


class _Lin(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.w = nn.Linear(cin, cout)


class _Marker(nn.Module):
    pass


def _packed(conv):
    """split-bf16 operand image of the (frozen, eval-mode) Conv1d weight, output channels zero-padded
    to a multiple of 128; rebuilt when the parameter changes."""
    w = conv.conv.weight
    key = (w.data_ptr(), w._version, str(w.device))
    if getattr(conv, "_img_key", None) != key:
        Cout, Cin, K = w.shape
        npad = -(-Cout // 128) * 128
        wpad = w.detach()
        if npad != Cout:
            wpad = torch.cat([wpad, torch.zeros(npad - Cout, Cin, K, dtype=w.dtype, device=w.device)])
        conv._img = ops.pack_weights(wpad.contiguous().float(), "conv_fwd", torch.float32, L.BF16X3)
        conv._img_key, conv._npad = key, npad
    return conv._img, conv._npad


def _tdnn(x, conv, bn, slope=0.01):
    lib = L.load()
    B, T, Cin = x.shape
    Cout = conv.conv.out_channels
    _, _, s, t = bn.affine()
    img, npad = _packed(conv)
    y = torch.empty(B, T, Cout, dtype=torch.float32, device=x.device)
    L.check(lib.sa_tdnn_fwd(L.ptr(x), L.ptr(img), L.ptr(conv.conv.bias), L.ptr(s), L.ptr(t),
                            L.ptr(y), B, T, Cin, Cout, npad, conv.kernel_size, conv.dilation,
                            C.c_float(slope), L.stream()), "sa_tdnn_fwd")
    return y


class Xvector(nn.Module):
    def __init__(self, in_channels=80, lin_neurons=128, tdnn_channels=(512, 512, 512, 512, 1500),
                 tdnn_kernel_sizes=(5, 3, 3, 1, 1), tdnn_dilations=(1, 2, 3, 1, 1), pooling_noise=True,
                 **_ignored):
        super().__init__()
        self.blocks = nn.ModuleList()
        for c, k, d in zip(tdnn_channels, tdnn_kernel_sizes, tdnn_dilations):
            self.blocks.extend([_Conv(in_channels, c, k, d), _Marker(), _BN(c)])
            in_channels = c
        self.blocks.append(_Marker())                       # StatisticsPooling
        self.blocks.append(_Lin(in_channels * 2, lin_neurons))
        self.pooling_noise = pooling_noise
        self.eval()

    @torch.no_grad()
    def forward(self, x, lens=None):
        lib = L.load()
        x = x.contiguous().float()
        nb = (len(self.blocks) - 2) // 3
        for i in range(nb):
            x = _tdnn(x, self.blocks[3 * i], self.blocks[3 * i + 2])
        B, T, Cc = x.shape
        noise = None
        if torch.is_tensor(self.pooling_noise):
            noise = self.pooling_noise.to(x.device).float().contiguous()
        elif self.pooling_noise:
            g = torch.randn(B, Cc, device=x.device)
            g = g - g.min()
            noise = (g / g.max()).contiguous()
        pooled = torch.empty(B, 2 * Cc, dtype=torch.float32, device=x.device)
        lens_d = None if lens is None else lens.to(x.device).float().contiguous()
        L.check(lib.sa_time_pool(L.ptr(x), L.ptr(lens_d), L.ptr(noise), B, T, Cc, C.c_float(1e-5),
                                 L.ptr(pooled), L.stream()), "sa_time_pool")
        lin = self.blocks[-1].w
        emb = ops.dense(pooled, lin.weight, lin.bias, lin.out_features, lin.in_features)
        return emb.unsqueeze(1)                             # [B, 1, emb]


class _Block(nn.Module):
    def __init__(self, n_in, n_out):
        super().__init__()
        self.linear, self.act, self.norm = _Lin(n_in, n_out), _Marker(), _BN(n_out)


class Classifier(nn.Module):
    def __init__(self, input_shape=None, lin_blocks=1, lin_neurons=128, out_neurons=2, **_ignored):
        super().__init__()
        emb = input_shape[-1] if input_shape else 128
        assert lin_blocks == 1
        self.act = _Marker()
        self.norm = _BN(emb)
        self.DNN = nn.ModuleDict({"block_0": _Block(emb, lin_neurons)})
        self.out = _Lin(lin_neurons, out_neurons)
        self.eval()

    @staticmethod
    def _leaky_bn(x, bn):
        lib = L.load()
        _, _, s, t = bn.affine()
        y = torch.empty_like(x)
        L.check(lib.sa_leaky_affine(L.ptr(x), L.ptr(s), L.ptr(t), C.c_float(0.01), x.shape[0], x.shape[1],
                                    L.ptr(y), L.stream()), "sa_leaky_affine")
        return y

    @torch.no_grad()
    def forward(self, x):
        v = x.reshape(x.shape[0], -1).contiguous().float()
        v = self._leaky_bn(v, self.norm)
        blk = self.DNN["block_0"]
        v = ops.dense(v, blk.linear.w.weight, blk.linear.w.bias, blk.linear.w.out_features,
                      blk.linear.w.in_features)
        v = self._leaky_bn(v, blk.norm)
        v = ops.dense(v, self.out.w.weight, self.out.w.bias, self.out.w.out_features, self.out.w.in_features)
        return ops.log_softmax(v).unsqueeze(1)              # [B, 1, classes]


class EncoderClassifier(nn.Module):
    """embedding_model + classifier with the fork's classify_batch_feats()."""

    def __init__(self, embedding_model=None, classifier=None):
        super().__init__()
        self.embedding_model = embedding_model or Xvector()
        self.classifier = classifier or Classifier()

    @torch.no_grad()
    def classify_batch_feats(self, feats, wav_lens=None):
        out_prob = self.classifier(self.embedding_model(feats, wav_lens)).squeeze(1)
        score, index = torch.max(out_prob, dim=-1)
        return out_prob, score, index
