"""x-vector gender classifier on libsa_hip.so (in the training graph: 8f-1): drop-in for the
``embedding_model`` / ``classifier`` pair of speechbrain_configs/evaluator_inference.yaml:34-48
(``speechbrain.lobes.models.Xvector.Xvector`` / ``.Classifier``, restated in the reference at
models/external_gender_classifiers.py:24-183).  Same constructor defaults as that config, same
parameter names (``blocks.0.conv.weight`` ... ``blocks.16.w.weight``; ``norm.norm.weight``,
``DNN.block_0.linear.w.weight``, ``out.w.weight`` as in the reference's classifier.ckpt), eval
mode (BatchNorm running statistics).  ``classify_batch_feats(feats, lens)`` is the call the
reference's fork adds (speechbrain_convae_train.py:139,146): returns (log_probs, score, index).

``EncoderClassifier.forward(feats)`` is the same computation INSIDE the training graph
(models/EndToEnd.py:57-61,81: ``self.sex_classifier(input)`` on the reconstructed features, the
classifier pretrained and frozen): differentiable with respect to the features only
(sa_tdnn_bwd_input / sa_tdnn_fold / sa_time_pool_bwd / sa_leaky_affine_bwd); its parameters never
receive gradients.
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


def _packed_dgrad(conv):
    """the same weight as a data-gradient operand: reduction over the block's output channels
    (zero-padded to a multiple of 16), produced = its input channels (zero-padded to 128)."""
    w = conv.conv.weight
    key = (w.data_ptr(), w._version, str(w.device))
    if getattr(conv, "_dimg_key", None) != key:
        Cout, Cin, K = w.shape
        cred, npad = -(-Cout // 16) * 16, -(-Cin // 128) * 128
        wpad = torch.zeros(cred, npad, K, dtype=torch.float32, device=w.device)
        wpad[:Cout, :Cin] = w.detach().float()
        conv._dimg = ops.pack_weights(wpad, "conv_dgrad", torch.float32, L.BF16X3)
        conv._dimg_key, conv._dgeom = key, (cred, npad)
    return conv._dimg, conv._dgeom


def _tdnn_bwd(dy, mask, conv, bn, slope=0.01):
    """d loss / d x of one frozen TDNN block from d loss / d y and the forward's LeakyReLU mask."""
    lib = L.load()
    B, T, Cy = mask.shape
    Cin, K, dil = conv.conv.in_channels, conv.kernel_size, conv.dilation
    _, _, s, _ = bn.affine()
    img, (cred, npad) = _packed_dgrad(conv)
    pad = dil * (K - 1) // 2
    dxe = torch.empty(B, T + 2 * pad, Cin, dtype=torch.float32, device=dy.device)
    L.check(lib.sa_tdnn_bwd_input(L.ptr(dy), L.ptr(mask), L.ptr(s), L.ptr(img), L.ptr(dxe), B, T,
                                  Cy, cred, Cin, npad, K, dil, C.c_float(slope), L.stream()),
            "sa_tdnn_bwd_input")
    if pad == 0:
        return dxe
    dx = torch.empty(B, T, Cin, dtype=torch.float32, device=dy.device)
    L.check(lib.sa_tdnn_fold(L.ptr(dxe), L.ptr(dx), B, T, Cin, pad, L.stream()), "sa_tdnn_fold")
    return dx


def _tdnn(x, conv, bn, slope=0.01, want_mask=False):
    lib = L.load()
    B, T, Cin = x.shape
    Cout = conv.conv.out_channels
    _, _, s, t = bn.affine()
    img, npad = _packed(conv)
    y = torch.empty(B, T, Cout, dtype=torch.float32, device=x.device)
    mask = torch.empty(B, T, Cout, dtype=torch.uint8, device=x.device) if want_mask else None
    L.check(lib.sa_tdnn_fwd(L.ptr(x), L.ptr(img), L.ptr(conv.conv.bias), L.ptr(s), L.ptr(t),
                            L.ptr(y), B, T, Cin, Cout, npad, conv.kernel_size, conv.dilation,
                            C.c_float(slope), L.ptr(mask), L.stream()), "sa_tdnn_fwd")
    return (y, mask) if want_mask else y


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


def _leaky_bn_bwd(dy, x, bn):
    lib = L.load()
    _, _, s, _ = bn.affine()
    dx = torch.empty_like(x)
    L.check(lib.sa_leaky_affine_bwd(L.ptr(dy), L.ptr(x), L.ptr(s), C.c_float(0.01), x.shape[0], x.shape[1],
                                    L.ptr(dx), L.stream()), "sa_leaky_affine_bwd")
    return dx


class _XvFn(torch.autograd.Function):
    """log-probabilities of the frozen x-vector classifier as a function of the features."""

    @staticmethod
    def forward(ctx, enc, feats, lens):
        lib = L.load()
        xv, cl = enc.embedding_model, enc.classifier
        x = feats.detach().contiguous().float()
        nb = (len(xv.blocks) - 2) // 3
        h, masks = x, []
        for i in range(nb):
            h, m = _tdnn(h, xv.blocks[3 * i], xv.blocks[3 * i + 2], want_mask=True)
            masks.append(m)
        B, T, Cc = h.shape
        lens_d = None if lens is None else lens.to(h.device).float().contiguous()
        pooled0 = torch.empty(B, 2 * Cc, dtype=torch.float32, device=h.device)
        L.check(lib.sa_time_pool(L.ptr(h), L.ptr(lens_d), None, B, T, Cc, C.c_float(1e-5), L.ptr(pooled0),
                                 L.stream()), "sa_time_pool")
        pooled = pooled0
        noise = xv.pooling_noise
        if noise is not None and noise is not False:
            if torch.is_tensor(noise):
                g = noise.to(h.device).float()
            else:
                g = torch.randn(B, Cc, device=h.device)
                g = g - g.min()
                g = g / g.max()
            pooled = pooled0.clone()
            pooled[:, :Cc] += 1e-5 * ((1.0 - 9.0) * g + 9.0)
        lin = xv.blocks[-1].w
        emb = ops.dense(pooled, lin.weight, lin.bias, lin.out_features, lin.in_features)
        v1 = Classifier._leaky_bn(emb, cl.norm)
        blk = cl.DNN["block_0"]
        h1 = ops.dense(v1, blk.linear.w.weight, blk.linear.w.bias, blk.linear.w.out_features,
                       blk.linear.w.in_features)
        v2 = Classifier._leaky_bn(h1, blk.norm)
        logits = ops.dense(v2, cl.out.w.weight, cl.out.w.bias, cl.out.w.out_features, cl.out.w.in_features)
        logp = ops.log_softmax(logits)
        ctx.enc, ctx.saved = enc, (masks, h, lens_d, pooled0, emb, h1, logp)
        return logp

    @staticmethod
    def backward(ctx, d_logp):
        lib = L.load()
        enc = ctx.enc
        xv, cl = enc.embedding_model, enc.classifier
        masks, h, lens_d, pooled0, emb, h1, logp = ctx.saved
        blk = cl.DNN["block_0"]
        g = ops.log_softmax_bwd(d_logp.contiguous().float(), logp)
        g = ops.dense(g, cl.out.w.weight, None, cl.out.w.in_features, cl.out.w.out_features, transpose_w=True)
        g = _leaky_bn_bwd(g, h1, blk.norm)
        g = ops.dense(g, blk.linear.w.weight, None, blk.linear.w.in_features, blk.linear.w.out_features,
                      transpose_w=True)
        g = _leaky_bn_bwd(g, emb, cl.norm)
        lin = xv.blocks[-1].w
        gp = ops.dense(g, lin.weight, None, lin.in_features, lin.out_features, transpose_w=True)
        B, T, Cc = h.shape
        gh = torch.empty_like(h)
        L.check(lib.sa_time_pool_bwd(L.ptr(h), L.ptr(lens_d), L.ptr(gp), L.ptr(pooled0), B, T, Cc,
                                     C.c_float(1e-5), L.ptr(gh), L.stream()), "sa_time_pool_bwd")
        nb = (len(xv.blocks) - 2) // 3
        for i in reversed(range(nb)):
            gh = _tdnn_bwd(gh, masks[i], xv.blocks[3 * i], xv.blocks[3 * i + 2])
        ctx.saved = None
        return None, gh, None


class EncoderClassifier(nn.Module):
    """embedding_model + classifier with the fork's classify_batch_feats()."""

    def __init__(self, embedding_model=None, classifier=None):
        super().__init__()
        self.embedding_model = embedding_model or Xvector()
        self.classifier = classifier or Classifier()

    def forward(self, feats, wav_lens=None):
        """(log_probs [B, classes], score, index) like classify_batch_feats, but part of the
        autograd graph through `feats` (models/EndToEnd.py:81); the classifier stays frozen."""
        if not feats.is_cuda:
            raise L.SaHipError("the x-vector classifier runs on the GPU only (no CPU fallback)")
        out_prob = _XvFn.apply(self, feats, wav_lens)
        score, index = torch.max(out_prob.detach(), dim=-1)
        return out_prob, score, index

    @torch.no_grad()
    def classify_batch_feats(self, feats, wav_lens=None):
        out_prob = self.classifier(self.embedding_model(feats, wav_lens)).squeeze(1)
        score, index = torch.max(out_prob, dim=-1)
        return out_prob, score, index
