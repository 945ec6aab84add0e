"""ConvReconstruction: drop-in for ``models.EndToEnd.ConvReconstruction`` of the reference
(models/EndToEnd.py:36-87, BASELINE config 4 / SURVEY 8f-1) with forward + backward on libsa_hip.so.

  feats [B, T, 80] -> reshape [B, 1, T*80] -> encoder (:40-54):
      Conv1d(1->32, k15, p7) -> InstanceNorm(32) -> x*sigmoid(x) -> Conv1d(32->64, k5, s2, p2) -> IN(64) -> act
      -> Conv1d(64->64, k5, p2) -> IN(64) -> act -> ConvTranspose1d(64->32, k5, s2, p2, op1) -> IN(32) -> act
      -> Conv1d(32->1, k15, p7)                                    -> recon [B, T, 80]
  sex_classifier (:57-61,81): the PRETRAINED, frozen x-vector EncoderClassifier applied to the
      reconstruction -> (log_probs, score, index); only the gradient with respect to the
      reconstruction flows back (xvector.EncoderClassifier.forward).

Same parameter names / shapes as the reference module (``encoder.0.weight`` ... ``encoder.12.bias``;
the torch.nn layers are parameter containers only).  The conv stack is a subset of the
ConvAutoencoder's kernels (sa_conv1toC / sa_conv_gemm 32->64 s2, 64->64, ConvT 64->32 /
sa_convCto1, statistics in the producers' epilogues, normalisation + activation in the consumers'
prologues); the backward uses the two-pass normalisation backward (statistics in the data-gradient
epilogue, sa_ew_apply) -- this "next" row is built for parity first.

The reference constructs the classifier from absolute paths on its authors' machine
(``EncoderClassifier.from_hparams(source="/home/ubuntu/...")``, :57-61); here it is passed in
(``ConvReconstruction(sex_classifier=...)``) or built with random weights.
"""
import functools

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from ._lib import SaHipError
from .convae import GLU, CONVT_WG_TAPS, K5
from .xvector import EncoderClassifier


class ConvReconstruction(nn.Module):
    def __init__(self, sex_classifier=None, precision="bf16x3"):
        super().__init__()
        if precision not in ("bf16x3", "f32"):
            raise SaHipError("ConvReconstruction runs in precision bf16x3 or f32")
        self.precision = precision
        self.act_dtype, self.kcode = ops.PRECISIONS[precision]
        self.encoder = nn.Sequential(
            nn.Conv1d(1, 32, 15, 1, 7), nn.InstanceNorm1d(32, affine=True), GLU(),
            nn.Conv1d(32, 64, 5, 2, 2), nn.InstanceNorm1d(64, affine=True), GLU(),
            nn.Conv1d(64, 64, 5, 1, 2), nn.InstanceNorm1d(64, affine=True), GLU(),
            nn.ConvTranspose1d(64, 32, 5, 2, 2, output_padding=1), nn.InstanceNorm1d(32, affine=True), GLU(),
            nn.Conv1d(32, 1, 15, 1, 7),
        )
        self.sex_classifier = sex_classifier if sex_classifier is not None else EncoderClassifier()
        for p in self.sex_classifier.parameters():          # pretrained and frozen in the reference
            p.requires_grad = False
        self.sex_classifier.eval()

    def train(self, mode=True):
        super().train(mode)
        self.sex_classifier.eval()                          # BatchNorm running statistics, always
        return self

    def forward(self, feats):
        names, params = zip(*((k, p) for k, p in self.named_parameters() if k.startswith("encoder.")))
        recon = _ConvRecFn.apply(self, names, feats, *params)
        logp, score, index = self.sex_classifier(recon)
        return recon, logp


class _ConvRecFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, names, feats, *params):
        ctx.set_materialize_grads(False)
        P = dict(zip(names, params))
        dt, code = model.act_dtype, model.kcode
        B, T, Fd = feats.shape
        Ltot = T * Fd
        if Fd != 80 or Ltot % 2:
            raise SaHipError("ConvReconstruction expects feats [B, T, 80]")
        if not feats.is_cuda:
            raise SaHipError("ConvReconstruction runs on the GPU only (no CPU fallback)")
        L2 = Ltot // 2
        x0 = feats.detach().reshape(B, Ltot).contiguous().float()

        def pack(key, kind):
            return ops.pack_weights(P[key].detach(), kind, dt, code)

        def inorm(stats, n, prefix, C):
            return ops.fin_in_fwd(ops.sum_partials(stats, B), B, C, n, P[prefix + ".weight"], P[prefix + ".bias"])

        W = {("encoder.3.weight", "conv_fwd"): pack("encoder.3.weight", "conv_fwd"),
             ("encoder.6.weight", "conv_fwd"): pack("encoder.6.weight", "conv_fwd"),
             ("encoder.9.weight", "convT_fwd"): pack("encoder.9.weight", "convT_fwd")}
        y0, st = ops.conv1toC(x0, P["encoder.0.weight"], P["encoder.0.bias"], dt, want_stats=True)
        n0 = inorm(st, Ltot, "encoder.1", 32)
        y1, st = ops.conv_gemm(y0, W[("encoder.3.weight", "conv_fwd")], P["encoder.3.bias"], 32, 64, 2, 1,
                               ops.taps_conv(K5, 1, 2), L2, s1=n0[2], t1=n0[3], swish=True, want_stats=True,
                               code=code)
        n1 = inorm(st, L2, "encoder.4", 64)
        y2, st = ops.conv_gemm(y1, W[("encoder.6.weight", "conv_fwd")], P["encoder.6.bias"], 64, 64, 1, 1,
                               ops.taps_conv(K5, 1, 2), L2, s1=n1[2], t1=n1[3], swish=True, want_stats=True,
                               code=code)
        n2 = inorm(st, L2, "encoder.7", 64)
        y3, st = ops.conv_gemm(y2, W[("encoder.9.weight", "convT_fwd")], P["encoder.9.bias"], 64, 32, 1, 2,
                               ops.UP2, Ltot, s1=n2[2], t1=n2[3], swish=True, want_stats=True, code=code)
        n3 = inorm(st, Ltot, "encoder.10", 32)
        recon = ops.convCto1(y3, P["encoder.12.weight"], P["encoder.12.bias"], n3[2], n3[3], True)
        ctx.S = dict(x0=x0, y=[y0, y1, y2, y3], n=[n0, n1, n2, n3], dims=(B, T, Ltot, L2))
        ctx.model, ctx.names, ctx.params = model, names, params
        ctx.need_input_grad = feats.requires_grad
        return recon.view(B, T, Fd)

    @staticmethod
    def backward(ctx, d_recon):
        S, model, names = ctx.S, ctx.model, ctx.names
        if S is None:
            raise SaHipError("ConvReconstruction backward called twice (saved tensors were released)")
        P = dict(zip(names, ctx.params))
        dt, code = model.act_dtype, model.kcode
        B, T, Ltot, L2 = S["dims"]
        y0, y1, y2, y3 = S["y"]
        n0, n1, n2, n3 = S["n"]
        dev = y0.device
        G = {k: None for k in names}
        need = {k: p.requires_grad for k, p in P.items()}
        if d_recon is None:
            ctx.S = None
            return (None, None, None) + tuple(None for _ in names)
        wg = functools.partial(ops.wgrad, code=ops.WGRAD_CODE[model.precision])

        def newg(key):
            return torch.empty_like(P[key])

        def pack(key, kind):
            return ops.pack_weights(P[key].detach(), kind, dt, code)

        def norm_bwd(g, st, y, nrm, C, Ln, prefix, bias_key):
            """g = d z (already multiplied by the activation derivative), st = partial (sum dz,
            sum dz*xhat): InstanceNorm backward coefficients, then d y = c1*dz + c2*y + c3 in place;
            also the gradients of the norm's affine parameters and of the conv bias in front."""
            sums = ops.sum_partials(st, B)
            dg, db = newg(prefix + ".weight"), newg(prefix + ".bias")
            c1, c2, c3 = ops.fin_norm_bwd(sums, sums, B * C, C, Ln, P[prefix + ".weight"], nrm[0], nrm[1],
                                          dgamma=dg, dbeta=db)
            G[prefix + ".weight"], G[prefix + ".bias"] = dg, db
            st2 = ops.ew("apply", g, y, C, out=g, c1=c1, c2=c2, c3=c3)
            if need[bias_key]:
                G[bias_key] = ops.fin_bias(ops.sum_partials(st2, B), B, C, newg(bias_key))
            return g

        def in_ep(y, nrm):
            return dict(mode=1, x=y, s1=nrm[2], t1=nrm[3], mean=nrm[0], rstd=nrm[1])

        g_rec = d_recon.reshape(B, Ltot).contiguous().float()
        if need["encoder.12.bias"]:
            G["encoder.12.bias"] = ops.sum_partials(g_rec.view(4 * B, Ltot // 4), 1, n=Ltot // 4).sum().float().reshape(1)
        if need["encoder.12.weight"]:
            G["encoder.12.weight"] = ops.wgrad1C(g_rec, y3, newg("encoder.12.weight"), flip=True,
                                                 s1=n3[2], t1=n3[3], swish=True)
        g, st = ops.conv1toC(g_rec, P["encoder.12.weight"], None, dt, flip=True, want_stats=True,
                             ep=dict(x=y3, s1=n3[2], t1=n3[3], mean=n3[0], rstd=n3[1]))          # d z3
        g = norm_bwd(g, st, y3, n3, 32, Ltot, "encoder.10", "encoder.9.bias")                    # d y3
        if need["encoder.9.weight"]:
            G["encoder.9.weight"] = wg(y2, g, 64, 32, 1, 2, CONVT_WG_TAPS, L2, newg("encoder.9.weight"),
                                       (32 * K5, K5, 1), s1=n2[2], t1=n2[3], swish=True)
        g, st = ops.conv_gemm(g, pack("encoder.9.weight", "convT_dgrad"), None, 32, 64, 2, 1,
                              ops.taps_convT_dgrad(), L2, want_stats=True, ep=in_ep(y2, n2), code=code)   # d z2
        g = norm_bwd(g, st, y2, n2, 64, L2, "encoder.7", "encoder.6.bias")                       # d y2
        taps5 = [(k - 2, 0) for k in range(K5)]
        if need["encoder.6.weight"]:
            G["encoder.6.weight"] = wg(y1, g, 64, 64, 1, 1, taps5, L2, newg("encoder.6.weight"),
                                       (K5, 64 * K5, 1), s1=n1[2], t1=n1[3], swish=True)
        g, st = ops.conv_gemm(g, pack("encoder.6.weight", "conv_dgrad"), None, 64, 64, 1, 1,
                              ops.taps_conv_dgrad_s1(K5, 1, 2), L2, want_stats=True, ep=in_ep(y1, n1),
                              code=code)                                                          # d z1
        g = norm_bwd(g, st, y1, n1, 64, L2, "encoder.4", "encoder.3.bias")                       # d y1
        if need["encoder.3.weight"]:
            G["encoder.3.weight"] = wg(y0, g, 32, 64, 2, 1, taps5, L2, newg("encoder.3.weight"),
                                       (K5, 32 * K5, 1), s1=n0[2], t1=n0[3], swish=True)
        g, st = ops.conv_gemm(g, pack("encoder.3.weight", "conv_dgrad"), None, 64, 32, 1, 2, ops.UP2, Ltot,
                              want_stats=True, ep=in_ep(y0, n0), code=code)                      # d z0
        g = norm_bwd(g, st, y0, n0, 32, Ltot, "encoder.1", "encoder.0.bias")                     # d y0
        if need["encoder.0.weight"]:
            G["encoder.0.weight"] = ops.wgrad1C(S["x0"], g, newg("encoder.0.weight"))
        d_feats = None
        if ctx.need_input_grad:
            d_feats = ops.convCto1(g, P["encoder.0.weight"], None, flip=True).view(B, T, 80)
        ctx.S = None
        return (None, None, d_feats) + tuple(G[k] if need[k] else None for k in names)
