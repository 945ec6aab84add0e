"""ConvAutoencoder: drop-in for ``models.ConvAutoEncoder.ConvAutoencoder`` of the reference
(models/ConvAutoEncoder.py:136-200) with the whole forward + backward on libsa_hip.so.

Same constructor (no arguments needed), same ``forward(feats[B,T,80]) -> (recon[B,T,80],
log_probs[B,2])``, same parameter / buffer names and shapes (``encoder.0.weight`` ...
``sex_classifier.classify.6.bias``; the torch.nn layers below are used ONLY as parameter
containers so that ``state_dict()``, ``named_parameters()`` with the reference's
``"sex_classifier" in name`` freeze logic, and default initialisation are the reference's).
Their ``forward`` is never called: one ``torch.autograd.Function`` runs the fused pipeline
  - activations channels-last [B, L, C] in ``dtype`` (bf16 default, fp32 available),
  - InstanceNorm / BatchNorm statistics from the producing conv's epilogue, normalisation +
    x*sigmoid(x) applied in the consumer's prologue (the normalised tensors are never stored),
  - GradReverse folded into the first classifier BatchNorm's backward coefficients,
  - the reshape-not-transpose before StatisticsPooling (models/ConvAutoEncoder.py:61).
"""
import functools
import os

import torch
import torch.nn as nn

from . import distributed as sdist
from . import ops
from . import _lib as L
from ._lib import SaHipError

K5 = 5
CONVT_WG_TAPS = [(1, 0), (1, 1), (0, 0), (0, 1), (-1, 0)]      # (input row offset, output phase) per tap


class _ParamOnly(nn.Module):
    """activation placeholders keep the Sequential indices of the reference."""

    def forward(self, x):                                        # pragma: no cover
        raise SaHipError("parameter container only; use ConvAutoencoder.forward")


class GLU(_ParamOnly):
    pass


class StatisticsPooling(_ParamOnly):
    pass


class TDNNSexClassifier(nn.Module):
    def __init__(self, num_classes=2):
        super().__init__()
        self.tdnn = nn.Sequential(
            nn.Conv1d(128, 128, kernel_size=5, dilation=1), nn.ReLU(), nn.BatchNorm1d(128),
            nn.Conv1d(128, 128, kernel_size=3, dilation=2), nn.ReLU(), nn.BatchNorm1d(128),
            nn.Conv1d(128, 128, kernel_size=3, dilation=3), nn.ReLU(), nn.BatchNorm1d(128),
        )
        self.norm = nn.BatchNorm1d(128)
        self.stats_pooling = StatisticsPooling()
        self.classify = nn.Sequential(
            nn.Linear(256, 128), nn.ReLU(), nn.BatchNorm1d(128),
            nn.Linear(128, 64), nn.ReLU(), nn.BatchNorm1d(64),
            nn.Linear(64, num_classes),
        )

    def forward(self, x):                                        # pragma: no cover
        raise SaHipError("parameter container only; use ConvAutoencoder.forward")


class ConvAutoencoder(nn.Module):
    def __init__(self, precision="bf16x3", pooling_noise=True, sync_bn=True, dtype=None,
                 cache_wgrad_operand=True):
        """precision: "bf16x3" (default: fp32 storage, split-bf16 operands on the bf16 MFMA --
        meets the 1e-4 parity bar), "bf16" (bf16 storage + single bf16 MFMA: fastest, ~2e-4 on
        recon), "f32" (exact fp32 MFMA).  dtype=torch.float32 / torch.bfloat16 selects "f32" /
        "bf16" (kept for callers that think in torch dtypes).  cache_wgrad_operand=False trades
        the bf16 operand cache (memory) for recomputation in the weight-gradient kernels."""
        super().__init__()
        if dtype is not None:
            precision = {torch.float32: "f32", torch.bfloat16: "bf16"}[dtype]
        if precision not in ops.PRECISIONS:
            raise SaHipError(f"unknown precision {precision!r}")
        self.precision = precision
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
        self.sex_classifier = TDNNSexClassifier(2)
        self.act_dtype, self.kcode = ops.PRECISIONS[precision]
        # kernel precision of the decoder convolutions (experiment knob, default = same as the rest)
        self.dec_kcode = self.kcode
        self.dgrad_kcode = ops.DGRAD_CODE[precision]
        # forward convs also store their transformed input in bf16 for the weight gradient
        # (bf16x3 / bf16x1f models; +1/2 of the saved activations in memory, identical results)
        self.cache_wgrad_operand = cache_wgrad_operand
        # with the cache: the norm-backward apply passes run in the prologue of the data-gradient
        # convolutions (False: separate sa_ew_apply launches)
        self.fuse_apply = True
        # option: weight-gradient GEMMs on a second stream beside the data-gradient convolutions of
        # the following layers.  Measured: the kernels do overlap, the step time does not change
        # (13.34 vs 13.35 ms at B=32), so it is off unless SA_OVERLAP_WGRAD=1.
        self.overlap_wgrad = os.environ.get("SA_OVERLAP_WGRAD", "0") == "1"
        # speechbrain's StatisticsPooling adds eps*U[1,9] to the pooled mean on every call
        # (train and eval); True reproduces that, a tensor [B,128] in [0,1] fixes the draw
        # (tests), False/None gives the deterministic form the oracle uses.
        self.pooling_noise = pooling_noise
        self.sync_bn = sync_bn
        # True: slab reductions and their finalisers in one launch each (sa_reduce_finalize: 46 fewer
        # launches per step).  Measured neutral to slightly slower (B = 32: 10.08 vs 10.04 ms; B = 10:
        # within the run-to-run spread), so the separate sa_sum_partials / sa_fin_* launches stay
        # the default (they are also what runs wherever the sums are all-reduced first).
        self.fused_finalize = os.environ.get("SA_FUSED_FINALIZE", "0") == "1"
        # the InstanceNorm FORWARD pairs alone (per utterance: no hand-off between workgroups in that mode)
        self.fused_in_fwd = os.environ.get("SA_FUSED_IN_FWD", "0") == "1"
        # The FC head of the classifier is ~12 (forward) / ~25 (backward) launches of 5-10 us each
        # that nothing else waits for until the branches merge: True runs them on a side stream
        # beside the decoder's convolutions (forward: after the pooling; backward: the decoder's
        # backward is issued first).  Measured neutral (B = 32: 9.75-9.78 vs 9.77-9.88 ms; B = 10:
        # 4.02-4.08 vs 4.02 ms), so off by default; never under SyncBatchNorm (the head then
        # contains collectives).
        self.overlap_head = os.environ.get("SA_OVERLAP_HEAD", "0") == "1"
        # the FC head as one forward and one backward launch (sa_head_fused.hip) where its BatchNorm
        # statistics are local and B fits one workgroup; SA_FUSED_HEAD=0: the separate launches
        self.fused_head = os.environ.get("SA_FUSED_HEAD", "1") == "1"
        # bias gradients of a backward stage as two launches at the end of the stage (sa_bias_multi)
        # instead of sa_sum_partials + sa_fin_bias per layer
        self.batch_bias = os.environ.get("SA_BATCH_BIAS", "1") == "1"
        self.batch_wred = os.environ.get("SA_BATCH_WRED", "1") == "1"
        # data-parallel FC head: None = every BatchNorm1d of the head exchanges its sums (any batch
        # split); "equal" = every rank holds a batch as large as this one (what data.shard_indices
        # deals); [b0, b1, ...] = the ranks' batch sizes.  With the sizes known the pooled rows are
        # exchanged ONCE and the head runs on the global batch on every rank (_head_plan).
        self.dp_batch_sizes = None
        # parity probe only (see _ConvAEFn.backward): the backward re-reads bf16-rounded stored tensors
        self.bwd_reload_bf16 = os.environ.get("SA_BWD_RELOAD_BF16", "0") == "1"
        # PARITY PROBE, not a mode (tools/bf16_reload_probe.py): what storing the convolution outputs in
        # bf16 would do -- 1: every stored forward tensor is rounded to bf16 right after the launch that
        # produced it (its statistics still come from the fp32 accumulators, operands stay split);
        # 2: the data gradients between the backward launches as well; 3: those gradients only;
        # 4: the decoder's stored outputs and the gradients of the decoder's backward only
        self.store_bf16_probe = int(os.environ.get("SA_STORE_BF16_PROBE", "0"))

    def forward(self, feats):
        # walking the module tree costs ~0.15 ms a call: the (names, parameters) lists are cached
        # and revalidated against the owners' registries (56 identity checks, ~5 us), so replacing
        # ANY parameter object (load into a sub-module, .to(), pruning, ...) is noticed
        c = self.__dict__.get("_np_cache")
        if c is None or any(o[n] is not p for o, n, p in c[2]):
            names, params, owners = [], [], []
            for mname, mod in self.named_modules():
                for pname, p in mod._parameters.items():
                    if p is not None:
                        names.append(f"{mname}.{pname}" if mname else pname)
                        params.append(p)
                        owners.append((mod._parameters, pname, p))
            c = self.__dict__["_np_cache"] = (tuple(names), tuple(params), owners)
        return _ConvAEFn.apply(self, c[0], feats, *c[1])

    def _wgrad_stream(self, device):
        if getattr(self, "_wgs", None) is None and device.type == "cuda":
            self._wgs = torch.cuda.Stream(device=device)
        return getattr(self, "_wgs", None)

    def _head_stream(self, device):
        if getattr(self, "_hs", None) is None and device.type == "cuda":
            self._hs = torch.cuda.Stream(device=device)
        return getattr(self, "_hs", None)

    def _side_stream(self, device):
        if getattr(self, "_side", None) is None and device.type == "cuda":
            self._side = torch.cuda.Stream(device=device)
        return getattr(self, "_side", None)

    # ---- SyncBatchNorm support: statistics sums are all-reduced across data-parallel ranks
    # (what speechbrain's Brain applies under DDP); identity on one process.
    def _bn_syncs(self):
        return self.sync_bn and sdist.dp_active()

    def _bn_allreduce(self, sums):
        if self._bn_syncs():
            sdist.all_reduce_now(sums)
            return sdist.world_size()
        return 1

    def _bn_global_counts(self, counts, device):
        """SyncBatchNorm element counts.  Ranks hold ragged batches (data.batches pads each rank's
        batch to its own longest utterance, the last batch may be short), so the per-layer counts
        are all-reduced beside the sums -- torch.nn.SyncBatchNorm, which speechbrain applies for
        the reference under DDP, exchanges counts the same way.  ONE collective per step for the
        six BatchNorms; the result stays on the device (the finalisers read it there:
        sa_fin_bn_fwd / sa_fin_norm_bwd / sa_bn2d_bwd `count_dev`).  None on a single process."""
        if not self._bn_syncs():
            return None
        key = (tuple(counts), str(device))
        cache = self.__dict__.setdefault("_count_cache", {})
        local = cache.get(key)
        if local is None:
            if len(cache) > 64:
                cache.clear()
            local = cache[key] = torch.tensor(counts, dtype=torch.float64).to(device)
        g = local.clone()
        sdist.all_reduce_now(g)
        return g

    def _bn_rows(self):
        """per-utterance partial rows may go straight to the finalisers when nothing is all-reduced"""
        return not self._bn_syncs()

    def _bn_global(self, local_sums):
        """(global sums, world): the all-reduced copy of the local sums, or the local sums
        themselves on one process (no copy)."""
        if not self._bn_syncs():
            return local_sums, 1
        g = local_sums.clone()
        return g, self._bn_allreduce(g)

    def _head_plan(self, B):
        """(row offset of this rank, global batch, world) when the FC head runs on the gathered
        global batch, else None.  The two BatchNorm1d exchanges of the forward and the two of the
        backward become one exchange of the pooled rows [B_global, 256] and one of d log p
        [B_global, 2]; the statistics are then local sums over the global rows."""
        if not self._bn_syncs() or self.dp_batch_sizes is None:
            return None
        W, r = sdist.world_size(), sdist.rank()
        sizes = [B] * W if isinstance(self.dp_batch_sizes, str) else [int(b) for b in self.dp_batch_sizes]
        if self.dp_batch_sizes != "equal" and isinstance(self.dp_batch_sizes, str):
            raise SaHipError(f"dp_batch_sizes: 'equal', a list of per-rank batch sizes or None, not {self.dp_batch_sizes!r}")
        if len(sizes) != W or sizes[r] != B:
            raise SaHipError(f"dp_batch_sizes {sizes} does not describe rank {r} of {W} holding {B} utterances")
        return sum(sizes[:r]), sum(sizes), W

    @staticmethod
    def _gather_rows(x, plan):
        """[B_global, C] with every rank's rows in rank order: this rank's rows into a zeroed
        buffer, one sum all-reduce (adds exact zeros, so the rows arrive bit-identical)."""
        off, Bg, _ = plan
        g = torch.zeros(Bg, x.shape[1], device=x.device, dtype=x.dtype)
        g[off:off + x.shape[0]].copy_(x)
        sdist.all_reduce_now(g)
        return g


class _W:
    """packed weight image + the kernel precision code it was packed for"""
    __slots__ = ("img", "code")

    def __init__(self, img, code):
        self.img, self.code = img, code


# every (weight, use) pair the fused forward / backward multiplies with
PACK_PLAN = (
    [(k, "conv_fwd") for k in ("encoder.2.weight", "encoder.5.weight", "encoder.8.weight",
                               "encoder.11.weight", "sex_classifier.tdnn.0.weight",
                               "sex_classifier.tdnn.3.weight", "sex_classifier.tdnn.6.weight",
                               "decoder.0.weight", "decoder.4.weight")]
    + [(k, "convT_fwd") for k in ("decoder.1.weight", "decoder.5.weight")]
    + [(k, "conv_dgrad") for k in ("encoder.2.weight", "encoder.5.weight", "encoder.8.weight",
                                   "encoder.11.weight", "sex_classifier.tdnn.0.weight",
                                   "sex_classifier.tdnn.3.weight", "sex_classifier.tdnn.6.weight",
                                   "decoder.0.weight", "decoder.4.weight")]
    + [(k, "convT_dgrad") for k in ("decoder.1.weight", "decoder.5.weight")])


def _kcode(model, key, kind):
    if kind.endswith("dgrad"):
        return model.dgrad_kcode
    return model.dec_kcode if key.startswith("decoder") else model.kcode


def _packed(model, P):
    """All operand images of the current weights, refreshed by ONE launch.  The image buffers
    are persistent (keyed by the parameters' storage): a backward uses the images its forward
    packed, which is what autograd's saved-tensor semantics ask for as long as the weights are
    not modified between the two."""
    items = [((k, kind), P[k], kind, _kcode(model, k, kind)) for k, kind in PACK_PLAN]
    key = tuple(w.data_ptr() for _, w, _, _ in items) + tuple(c for _, _, _, c in items)
    pk = getattr(model, "_pack_cache", None)
    if pk is None or pk[0] != key:
        pk = (key, ops.PackedWeights(items, model.act_dtype))
        model._pack_cache = pk
    pk[1].refresh()
    return {tag: _W(img, code) for tag, (img, code) in pk[1].images.items()}


def _conv(x, w, *args, **kw):
    return ops.conv_gemm(x, w.img, *args, code=w.code, **kw)


def _round_first(out):
    """store_bf16_probe: the launch's main output as a bf16-stored tensor would hold it"""
    y = out[0] if isinstance(out, tuple) else out
    if torch.is_tensor(y) and y.dtype == torch.float32:
        y.copy_(y.bfloat16())
    return out


class _PendingApply:
    """d z of a normalised layer together with the coefficients of d y = c1*dz + c2*y + c3 that the
    next data-gradient convolution applies in its prologue (instead of a sa_ew_apply pass)."""
    __slots__ = ("g", "y", "c", "per_c", "relu", "bias_key", "wgrads")

    def __init__(self, g, y, c, per_c, relu, bias_key):
        self.g, self.y, self.c, self.per_c, self.relu, self.bias_key = g, y, c, per_c, relu, bias_key
        self.wgrads = []


def _noise(model, B, device):
    n = model.pooling_noise
    if n is None or n is False:
        return None
    if torch.is_tensor(n):
        return n.to(device=device, dtype=torch.float32).contiguous()
    g = torch.randn(B, 128, device=device)                      # speechbrain _get_gauss_noise
    g = g - g.min()
    return (g / g.max()).contiguous()


class _ConvAEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, names, feats, *params):
        # an output the loss does not use arrives as None in backward (not as a zero tensor): the
        # parameters that only feed it then get no gradient at all, like in the reference's graph
        ctx.set_materialize_grads(False)
        P = dict(zip(names, params))
        dt = model.act_dtype
        train = model.training
        B, T, Fd = feats.shape
        Ltot = T * Fd
        if Fd != 80 or Ltot % 4:
            raise SaHipError("ConvAutoencoder expects feats [B, T, 80] with T*80 divisible by 4")
        if not feats.is_cuda:
            raise SaHipError("ConvAutoencoder runs on the GPU only (no CPU fallback)")
        L2, L4 = Ltot // 2, Ltot // 4
        S = {}                                                  # saved for backward
        x0 = feats.detach().reshape(B, Ltot).contiguous().float()
        W = _packed(model, P)
        pw = lambda k, kind: W[(k, kind)]
        # activation cache: each conv also writes its transformed input rows in bf16, the operand
        # its weight gradient multiplies with (saves the recomputation and half of the bytes there)
        A = {}
        cache_a = (train and model.cache_wgrad_operand and any(ctx.needs_input_grad)
                   and ops.WGRAD_CODE[model.precision] in (L.BF16X1F, L.BF16) and model.kcode != L.FP8)

        def cg(x, w, key, *args, **kw):
            if cache_a and key is not None and P[key].requires_grad:
                A[key] = kw["a_out"] = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
            out = _conv(x, w, *args, **kw)
            if model.store_bf16_probe in (1, 2) or (model.store_bf16_probe == 4 and str(key).startswith("decoder")):
                y_ = out[0] if isinstance(out, tuple) else out
                if y_.dtype == torch.float32:
                    y_.copy_(y_.bfloat16())
            return out

        ff = model.fused_finalize

        def inorm(stats, n, prefix, C):
            if ff or model.fused_in_fwd:
                return ops.reduce_finalize(L.FIN_IN_FWD, stats, B, C, count=n, gamma=P[prefix + ".weight"],
                                           beta=P[prefix + ".bias"])
            sums = ops.sum_partials(stats, B)
            return ops.fin_in_fwd(sums, B, C, n, P[prefix + ".weight"], P[prefix + ".bias"])

        def bn_stats(stats, count, mod, prefix, C, ci):
            """train-mode BatchNorm from the per-tile partial statistics of the producing launch"""
            if train and ff and not model._bn_syncs():
                out = ops.reduce_finalize(L.FIN_BN_FWD, stats, B, C, count=count, gamma=P[prefix + ".weight"],
                                          beta=P[prefix + ".bias"], run_mean=mod.running_mean,
                                          run_var=mod.running_var)
                tracked.append(mod.num_batches_tracked)
                return out
            sums = ops.sum_partials(stats, 1, rows=model._bn_rows()) if train else None
            return bnorm(sums, count, mod, prefix, C, ci)

        def bnorm(sums, count, mod, prefix, C, ci, local=False):
            """sums [C,2] local; returns (mean, rstd, scale, shift) per channel.  ci: index of
            this BatchNorm in the all-reduced count vector gc (SyncBatchNorm).  local: the sums
            already cover the global batch (gathered head rows), nothing to exchange."""
            if not train:
                return ops.fin_bn_eval(C, P[prefix + ".weight"], P[prefix + ".bias"],
                                       mod.running_mean, mod.running_var)
            if not local:
                model._bn_allreduce(sums)
            out = ops.fin_bn_fwd(sums, C, count, P[prefix + ".weight"], P[prefix + ".bias"],
                                 mod.running_mean, mod.running_var, count_dev=None if local else cdev(ci))
            tracked.append(mod.num_batches_tracked)
            return out

        tracked = []                                            # BatchNorm step counters, bumped together
        enc, dec, cls = model.encoder, model.decoder, model.sex_classifier
        La, Lb, Lc = L4 - 4, L4 - 8, L4 - 14
        # global element counts of the six BatchNorms (norm, tdnn.2/5/8, classify.2/5) under
        # SyncBatchNorm: one tiny all-reduce, device-resident; None on one process
        gc = model._bn_global_counts([B * L4, B * La, B * Lb, B * Lc, B, B], feats.device) if train else None
        cdev = lambda i: None if gc is None else gc[i:i + 1]
        # ---------------- encoder ----------------
        y0 = ops.conv1toC(x0, P["encoder.0.weight"], P["encoder.0.bias"], dt)
        if model.store_bf16_probe in (1, 2) and y0.dtype == torch.float32:
            y0.copy_(y0.bfloat16())
        y1, st = cg(y0, pw("encoder.2.weight", "conv_fwd"), "encoder.2.weight", P["encoder.2.bias"], 32, 64, 2, 1,
                               ops.taps_conv(K5, 1, 2), L2, swish=True, want_stats=True)
        n1 = inorm(st, L2, "encoder.3", 64)
        y2, st = cg(y1, pw("encoder.5.weight", "conv_fwd"), "encoder.5.weight", P["encoder.5.bias"], 64, 64, 1, 1,
                               ops.taps_conv(K5, 1, 2), L2, s1=n1[2], t1=n1[3], swish=True, want_stats=True)
        n2 = inorm(st, L2, "encoder.6", 64)
        y3, st = cg(y2, pw("encoder.8.weight", "conv_fwd"), "encoder.8.weight", P["encoder.8.bias"], 64, 128, 2, 1,
                               ops.taps_conv(K5, 1, 2), L4, s1=n2[2], t1=n2[3], swish=True, want_stats=True)
        n3 = inorm(st, L4, "encoder.9", 128)
        y4, st = cg(y3, pw("encoder.11.weight", "conv_fwd"), "encoder.11.weight", P["encoder.11.bias"], 128, 128, 1, 1,
                               ops.taps_conv(K5, 1, 2), L4, s1=n3[2], t1=n3[3], swish=True, want_stats=True)
        n4 = inorm(st, L4, "encoder.12", 128)
        # decoder.0 goes first: it stages the same transformed encoder output the classifier's input
        # BatchNorm needs statistics of, and leaves them as a by-product of its prologue
        y5 = cg(y4, pw("decoder.0.weight", "conv_fwd"), "decoder.0.weight", P["decoder.0.bias"], 128, 128, 1, 1,
                ops.taps_conv(K5, 1, 2), L4, s1=n4[2], t1=n4[3], swish=True, want_pro_stats=train)
        if train:
            y5, a4_stats = y5
        # ---------------- sex classifier (GradReverse = identity forward) ----------------
        bn_n = bn_stats(a4_stats if train else None, B * L4, cls.norm, "sex_classifier.norm", 128, 0)
        r0, st = cg(y4, pw("sex_classifier.tdnn.0.weight", "conv_fwd"), "sex_classifier.tdnn.0.weight",
                               P["sex_classifier.tdnn.0.bias"], 128, 128, 1, 1, ops.taps_conv(5, 1, 0), La,
                               s1=n4[2], t1=n4[3], swish=True, s2=bn_n[2], t2=bn_n[3], relu=True,
                               want_stats=True)
        bn0 = bn_stats(st, B * La, cls.tdnn[2], "sex_classifier.tdnn.2", 128, 1)
        r1, st = cg(r0, pw("sex_classifier.tdnn.3.weight", "conv_fwd"), "sex_classifier.tdnn.3.weight",
                               P["sex_classifier.tdnn.3.bias"], 128, 128, 1, 1, ops.taps_conv(3, 2, 0), Lb,
                               s2=bn0[2], t2=bn0[3], relu=True, want_stats=True)
        bn1 = bn_stats(st, B * Lb, cls.tdnn[5], "sex_classifier.tdnn.5", 128, 2)
        r2, st = cg(r1, pw("sex_classifier.tdnn.6.weight", "conv_fwd"), "sex_classifier.tdnn.6.weight",
                               P["sex_classifier.tdnn.6.bias"], 128, 128, 1, 1, ops.taps_conv(3, 3, 0), Lc,
                               s2=bn1[2], t2=bn1[3], relu=True, want_stats=True)
        bn2 = bn_stats(st, B * Lc, cls.tdnn[8], "sex_classifier.tdnn.8", 128, 3)
        pooled, pmean, psd = ops.pool_fwd(r2, bn2[2], bn2[3], noise=_noise(model, B, feats.device))
        def head_fwd(X, n, local=False):
            H1 = ops.dense(X, P["sex_classifier.classify.0.weight"], P["sex_classifier.classify.0.bias"],
                           128, 256, relu=True)
            f1 = bnorm(ops.colsums(H1) if train else None, n, cls.classify[2], "sex_classifier.classify.2", 128, 4, local)
            H2 = ops.dense(H1, P["sex_classifier.classify.3.weight"], P["sex_classifier.classify.3.bias"],
                           64, 128, ps=f1[2], pt=f1[3], relu=True)
            f2 = bnorm(ops.colsums(H2) if train else None, n, cls.classify[5], "sex_classifier.classify.5", 64, 5, local)
            logits = ops.dense(H2, P["sex_classifier.classify.6.weight"], P["sex_classifier.classify.6.bias"],
                               2, 64, ps=f2[2], pt=f2[3])
            return H1, f1, H2, f2, ops.log_softmax(logits)

        # data-parallel with known per-rank batch sizes: the pooled rows of all ranks, one exchange
        plan = model._head_plan(B) if train else None
        head_in, Bh = (model._gather_rows(pooled, plan), plan[1]) if plan else (pooled, B)
        # one launch for the whole head where the batch statistics are local (train mode, no
        # SyncBatchNorm exchange between the layers) and the batch fits one workgroup
        fused_head = (train and model.fused_head and (plan is not None or not model._bn_syncs())
                      and Bh <= ops.head_max_rows())
        clsP = {k[len("sex_classifier.classify."):]: v for k, v in P.items() if k.startswith("sex_classifier.classify.")}
        hs = (model._head_stream(feats.device) if (model.overlap_head and not model._bn_syncs() and not fused_head)
              else None)
        if fused_head:
            H1, f1, H2, f2, logp = ops.head_fwd(head_in, clsP, cls.classify[2], cls.classify[5])
            tracked += [cls.classify[2].num_batches_tracked, cls.classify[5].num_batches_tracked]
        elif hs is None:
            H1, f1, H2, f2, logp = head_fwd(head_in, Bh, plan is not None)
        else:                                  # beside the decoder's convolutions
            main = torch.cuda.current_stream()
            hs.wait_stream(main)
            with torch.cuda.stream(hs):
                H1, f1, H2, f2, logp = head_fwd(head_in, Bh)
        logp_all = logp
        if plan:                               # this rank's rows of the global log-probabilities
            logp = logp_all[plan[0]:plan[0] + B]
        # ---------------- decoder ----------------
        y6, st = cg(y5, pw("decoder.1.weight", "convT_fwd"), "decoder.1.weight", P["decoder.1.bias"], 128, 64, 1, 2,
                               ops.UP2, L2, want_stats=True)
        n6 = inorm(st, L2, "decoder.2", 64)
        y7 = cg(y6, pw("decoder.4.weight", "conv_fwd"), "decoder.4.weight", P["decoder.4.bias"], 64, 64, 1, 1,
                           ops.taps_conv(K5, 1, 2), L2, s1=n6[2], t1=n6[3], swish=True)
        y8, st = cg(y7, pw("decoder.5.weight", "convT_fwd"), "decoder.5.weight", P["decoder.5.bias"], 64, 32, 1, 2,
                               ops.UP2, Ltot, want_stats=True)
        n8 = inorm(st, Ltot, "decoder.6", 32)
        recon = ops.convCto1(y8, P["decoder.8.weight"], P["decoder.8.bias"], n8[2], n8[3], True)
        if hs is not None:
            main.wait_stream(hs)
            for tns in (H1, H2, logp_all, f1[0], f2[0]):      # allocated on the side stream, consumed on this one
                tns.record_stream(main)

        if tracked:
            torch._foreach_add_(tracked, 1)
        S.update(x0=x0, y=[y0, y1, y2, y3, y4, y5, y6, y7, y8], r=[r0, r1, r2],
                 n=[None, n1, n2, n3, n4, None, n6, None, n8], bn=[bn_n, bn0, bn1, bn2], f=[f1, f2],
                 pooled=head_in, pmean=pmean, psd=psd, H1=H1, H2=H2, logp=logp_all, head_plan=plan,
                 dims=(B, T, Ltot, L2, L4, La, Lb, Lc), train=train, W=W, A=A, gc=gc, fused_head=fused_head)
        ctx.S, ctx.model, ctx.names, ctx.params = S, model, names, params
        ctx.need_input_grad = feats.requires_grad
        return recon.view(B, T, Fd), logp

    @staticmethod
    def backward(ctx, d_recon, d_logp):
        S, model, names = ctx.S, ctx.model, ctx.names
        if S is None:
            raise SaHipError("ConvAutoencoder backward called twice (saved tensors were released)")
        P = dict(zip(names, ctx.params))
        dt = model.act_dtype
        B, T, Ltot, L2, L4, La, Lb, Lc = S["dims"]
        if not S["train"]:
            raise SaHipError("backward through eval-mode BatchNorm is not implemented")
        y0, y1, y2, y3, y4, y5, y6, y7, y8 = S["y"]
        r0, r1, r2 = S["r"]
        if model.bwd_reload_bf16:
            # PARITY PROBE, not a mode: what bf16 side copies of the stored forward
            # tensors would do to the gradients -- the tensors the backward only RE-READS (the
            # normalisation-backward prologues' y, the epilogues' x) are rounded to bf16 here, the
            # kernels and every other operand are unchanged (tools/bf16_reload_probe.py reports the result)
            rb = lambda t_: t_.bfloat16().float() if t_ is not None and t_.dtype == torch.float32 else t_
            y0, y1, y2, y3, y4, y5, y6, y7, y8 = (rb(t_) for t_ in (y0, y1, y2, y3, y4, y5, y6, y7, y8))
            r0, r1, r2 = (rb(t_) for t_ in (r0, r1, r2))
        n1, n2, n3, n4, n6, n8 = S["n"][1], S["n"][2], S["n"][3], S["n"][4], S["n"][6], S["n"][8]
        bn_n, bn0, bn1, bn2 = S["bn"]
        f1, f2 = S["f"]
        dev = y0.device
        G = {k: None for k in names}
        need = {k: p.requires_grad for k, p in P.items()}
        # which stages have anything to produce (the epoch-parity schedule of the reference freezes
        # either the classifier or everything else, speechbrain_convae_train.py:212-235): like
        # autograd in the reference, nothing is computed below the last tensor that needs a gradient
        need_stage = {st: any(v for k, v in need.items() if k.startswith(st))
                      for st in sdist.StageBuckets.STAGES}
        run_encoder = need_stage["encoder"] or ctx.need_input_grad
        run_decoder = need_stage["decoder"] or run_encoder
        buckets = sdist.StageBuckets(list(P.items()), dev, model._side_stream(dev))
        newg = buckets.view

        def setg(key, val):
            G[key] = newg(key).copy_(val.reshape(P[key].shape))
        W, A = S["W"], S["A"]
        gc = S["gc"]                                      # all-reduced BatchNorm counts (or None)
        cdev = lambda i: None if gc is None else gc[i:i + 1]
        pw = lambda k, kind: W[(k, kind)]
        # the split-K reducers of a stage's weight gradients wait for the end of the stage like the bias
        # gradients (nobody reads them earlier) and run as one launch (sa_wgrad_reduce_multi)
        pending_wred = []
        wg = functools.partial(ops.wgrad, code=ops.WGRAD_CODE[model.precision],
                               defer=pending_wred if model.batch_wred else None)
        ff = model.fused_finalize
        # with the bf16 operand caches in place the apply pass of every normalised layer whose
        # gradient feeds a convolution moves into that convolution's prologue
        fuse = bool(A) and model.fuse_apply and model.dgrad_kcode in (L.BF16X3, L.BF16)

        def cg(gin, w, *args, **kw):
            """data-gradient / forward-type launch; a _PendingApply input selects the
            normalisation-backward prologue, which also emits the bf16 d y for the deferred weight
            gradients and the column sums for the bias gradient."""
            if model.store_bf16_probe in (2, 3) or (model.store_bf16_probe == 4 and in_decoder[0]):
                return _round_first(cg_(gin, w, *args, **kw))
            return cg_(gin, w, *args, **kw)

        in_decoder = [False]

        def cg_(gin, w, *args, **kw):
            if not isinstance(gin, _PendingApply):
                return _conv(gin, w, *args, **kw)
            p = gin
            dyc = torch.empty(p.g.shape, dtype=torch.bfloat16, device=p.g.device) if p.wgrads else None
            want_cs = p.bias_key is not None and need[p.bias_key]
            out = _conv(p.g, w, *args, a_out=dyc, nb=dict(x=p.y, c1=p.c[0], c2=p.c[1], c3=p.c[2],
                                                           per_c=p.per_c, relu_mask=p.relu,
                                                           want_colsum=want_cs), **kw)
            if want_cs:
                cs = out[-1]
                out = out[:-1] if len(out) > 2 else out[0]
                nb_, nt_, cc_ = cs.shape
                if ff:
                    G[p.bias_key] = ops.reduce_finalize(L.FIN_BIAS, cs, nb_, cc_, ncomp=1, db=newg(p.bias_key))
                elif model.batch_bias:
                    G[p.bias_key] = newg(p.bias_key)
                    pending_bias.append((cs, nb_, cc_, 1, G[p.bias_key]))
                else:
                    G[p.bias_key] = ops.fin_bias(ops.sum_partials(cs.view(nb_, nt_, cc_, 1), nb_), nb_, cc_,
                                                 newg(p.bias_key), ncomp=1)
            for f in p.wgrads:
                f(dyc)
            p.wgrads = []
            return out

        def bias_from(stats, key, C):
            if need[key]:
                if ff:
                    G[key] = ops.reduce_finalize(L.FIN_BIAS, stats, B, C, db=newg(key))
                elif model.batch_bias:
                    G[key] = newg(key)
                    pending_bias.append((stats, B, C, 2, G[key]))
                else:
                    G[key] = ops.fin_bias(ops.sum_partials(stats, B), B, C, newg(key))

        # bias gradients wait for the end of their stage: nothing reads them before the stage's
        # bucket is reduced, so all of a stage's slab sums run as two launches (sa_bias_multi)
        pending_bias = []

        def flush_bias():
            if pending_bias:
                ops.bias_multi(pending_bias)
                pending_bias.clear()
            if pending_wred:
                ops.wgrad_reduce_multi(pending_wred)
                pending_wred.clear()

        def in_ep(y, nrm, g2=None):
            """fused-epilogue description of an [InstanceNorm -> swish] backward (stats pass)."""
            mean, rstd, scale, shift = nrm
            d = dict(mode=1, x=y, g2=g2, s1=scale, t1=shift, mean=mean, rstd=rstd)
            if isinstance(g2, _PendingApply):                 # apply of the `norm` BatchNorm rides along
                d.update(g2=g2.g, g2k=g2.c)
            return d

        def in_finish(g, st, y, nrm, C, Ln, prefix, bias_key):
            """g = d z (already multiplied by swish'), st = partial (sum dz, sum dz*yhat)."""
            mean, rstd = nrm[0], nrm[1]
            dg, db = newg(prefix + ".weight"), newg(prefix + ".bias")
            if ff:
                c1, c2, c3 = ops.reduce_finalize(L.FIN_IN_BWD, st, B, C, count=Ln, gamma=P[prefix + ".weight"],
                                                 mean=mean, rstd=rstd, dgamma=dg, dbeta=db)
            else:
                sums = ops.sum_partials(st, B)
                c1, c2, c3 = ops.fin_norm_bwd(sums, sums, B * C, C, Ln, P[prefix + ".weight"], mean, rstd,
                                              dgamma=dg, dbeta=db)
            G[prefix + ".weight"], G[prefix + ".bias"] = dg, db
            if fuse:
                return _PendingApply(g, y, (c1, c2, c3), False, False, bias_key)
            st2 = ops.ew("apply", g, y, C, out=g, c1=c1, c2=c2, c3=c3)
            bias_from(st2, bias_key, C)
            return g                                             # now d y

        def bn_ep(r, bn, xp=None):
            d = dict(mode=2, x=r, mean=bn[0], rstd=bn[1], per_c=True)
            if xp:
                d.update(s1=xp[0], t1=xp[1], xp_is_act=True)
            return d

        def bn_finish(g, st, r, bn, Ln, prefix, bias_key, ci, xp=None):
            """backward of [conv -> ReLU -> BatchNorm(prefix)] w.r.t. the conv output (stored r =
            relu output); xp=(s1,t1): the BN input is swish(r*s1+t1) instead (the `norm` BN on the
            encoder output, with GradReverse in front: sign -1, no ReLU mask)."""
            mean, rstd = bn[0], bn[1]
            kw = dict(s1=xp[0], t1=xp[1], xp_is_act=True) if xp else {}
            dg, db = newg(prefix + ".weight"), newg(prefix + ".bias")
            if ff and not model._bn_syncs():
                c1, c2, c3 = ops.reduce_finalize(L.FIN_BN_BWD, st, B, 128, count=float(B * Ln),
                                                 gamma=P[prefix + ".weight"], mean=mean, rstd=rstd,
                                                 sign=-1.0 if xp else 1.0, dgamma=dg, dbeta=db)
            else:
                lsums = ops.sum_partials(st, 1, rows=model._bn_rows())
                gsums, _ = model._bn_global(lsums)
                c1, c2, c3 = ops.fin_norm_bwd(gsums, lsums, 128, 128, float(B * Ln), P[prefix + ".weight"],
                                              mean, rstd, sign=-1.0 if xp else 1.0, dgamma=dg, dbeta=db,
                                              n_dev=cdev(ci))
            G[prefix + ".weight"], G[prefix + ".bias"] = dg, db
            if fuse:
                return _PendingApply(g, r, (c1, c2, c3), True, not xp, bias_key)
            st2 = ops.ew("apply", g, r, 128, out=g, c1=c1, c2=c2, c3=c3, relu_mask=not xp, per_c=True,
                         want_stats=bias_key is not None, **kw)
            if bias_key:
                bias_from(st2, bias_key, 128)
            return g

        ws = model._wgrad_stream(dev) if model.overlap_wgrad else None

        def on_side(fn, *tensors):
            """run fn on the weight-gradient stream, ordered after everything enqueued so far"""
            if ws is None:
                return fn()
            ws.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(ws):
                out = fn()
            for t in tensors:
                t.record_stream(ws)                  # the allocator must not recycle them early
            return out

        def side_join():
            if ws is not None:
                torch.cuda.current_stream().wait_stream(ws)

        def run_conv_wgrad(key, x, dy, cin, cout, sa, Mrows, K, dil, pad, pro):
            if key in A:
                x, pro = A[key], dict(x_pre=True, dy_pre=bool(pro.get("dy_pre")))
            G[key] = on_side(lambda: wg(x, dy, cin, cout, sa, 1, [(k * dil - pad, 0) for k in range(K)],
                                        Mrows, newg(key), (K, cin * K, 1), **pro), x, dy)

        def conv_wgrad(key, x, dy, cin, cout, sa, Mrows, K, dil, pad, **pro):
            if not need[key]:
                return
            if isinstance(dy, _PendingApply):                # runs once the bf16 d y exists
                dy.wgrads.append(lambda dyc: run_conv_wgrad(key, x, dyc, cin, cout, sa, Mrows, K, dil, pad,
                                                            dict(dy_pre=True)))
            else:
                run_conv_wgrad(key, x, dy, cin, cout, sa, Mrows, K, dil, pad, pro)

        def run_convT_wgrad(key, x, dy, cin, cout, Mrows, dy_pre):
            pro = {}
            if key in A:
                x, pro = A[key], dict(x_pre=True, dy_pre=dy_pre)
            G[key] = on_side(lambda: wg(x, dy, cin, cout, 1, 2, CONVT_WG_TAPS, Mrows, newg(key),
                                        (cout * K5, K5, 1), **pro), x, dy)

        def convT_wgrad(key, x, dy, cin, cout, Mrows):
            if not need[key]:
                return
            if isinstance(dy, _PendingApply):
                dy.wgrads.append(lambda dyc: run_convT_wgrad(key, x, dyc, cin, cout, Mrows, True))
            else:
                run_convT_wgrad(key, x, dy, cin, cout, Mrows, False)

        # an output the loss does not use (the endtoend "sex only" branch,
        # speechbrain_convae_train.py:112-113, leaves recon out of the graph): like autograd in the
        # reference, the parameters that only feed that output get NO gradient (None, so that Adam
        # skips them) rather than zeros
        recon_unused = d_recon is None
        if recon_unused:
            for k in need:
                if k.startswith("decoder"):
                    need[k] = False
            need_stage["decoder"] = False
        if d_logp is None:
            d_logp = torch.zeros(B, 2, device=dev)

        def finish():
            flush_bias()
            buckets.join()
            ctx.S = None
            grads = tuple(G[k] if need[k] else None for k in names)
            G.clear()
            buckets.views.clear()
            return (None, None, None) + grads

        # ======================= sex classifier: FC head =======================
        def head_bwd():
            c = "sex_classifier.classify."
            plan = S.get("head_plan")
            dlp, Bh = d_logp.contiguous().float(), B
            if plan:                                  # d log p of every rank's rows, one exchange
                dlp, Bh = model._gather_rows(dlp, plan), plan[1]
            dP = head_bwd_rows(c, dlp, Bh, plan is not None)
            if plan:
                # every rank now holds the head gradients of the SUM of the ranks' losses; the
                # data-parallel average that follows expects each rank's share
                hg = [G[k] for k in names if k.startswith(c) and k in G]
                if plan[2] > 1 and hg:
                    torch._foreach_mul_(hg, 1.0 / plan[2])
                dP = dP[plan[0]:plan[0] + B]
            return dP

        def head_bwd_rows(c, dlp, n, local):
            if S.get("fused_head"):                  # the whole chain in one launch (sa_head_bwd)
                short = ("0.weight", "0.bias", "2.weight", "2.bias", "3.weight", "3.bias", "5.weight", "5.bias",
                         "6.weight", "6.bias")
                views = {}
                for k in short:
                    if need[c + k]:
                        views[k] = G[c + k] = newg(c + k)
                clsP = {k: P[c + k] for k in short}
                return ops.head_bwd(dlp, S["logp"], S["pooled"], S["H1"], f1, S["H2"], f2, clsP, views)
            glob = (lambda l: l) if local else (lambda l: model._bn_global(l)[0])
            cd = (lambda i: None) if local else cdev
            dLG = ops.log_softmax_bwd(dlp, S["logp"])
            H1, H2 = S["H1"], S["H2"]
            G[c + "6.weight"] = ops.dense_wgrad(dLG, H2, newg(c + "6.weight"), ps=f2[2], pt=f2[3])
            G[c + "6.bias"] = newg(c + "6.bias"); ops.colsums(dLG, out0=G[c + "6.bias"])
            dN2 = ops.dense(dLG, P[c + "6.weight"], None, 64, 2, transpose_w=True)
            G[c + "5.weight"], G[c + "5.bias"] = newg(c + "5.weight"), newg(c + "5.bias")
            l2 = ops.colsums(dN2, H2, f2[0], f2[1], out0=G[c + "5.bias"], out1=G[c + "5.weight"])
            dH2 = ops.bn2d_bwd(dN2, H2, glob(l2), n, P[c + "5.weight"], f2[0], f2[1], True, count_dev=cd(5))
            G[c + "3.weight"] = ops.dense_wgrad(dH2, H1, newg(c + "3.weight"), ps=f1[2], pt=f1[3])
            G[c + "3.bias"] = newg(c + "3.bias"); ops.colsums(dH2, out0=G[c + "3.bias"])
            dN1 = ops.dense(dH2, P[c + "3.weight"], None, 128, 64, transpose_w=True)
            G[c + "2.weight"], G[c + "2.bias"] = newg(c + "2.weight"), newg(c + "2.bias")
            l1 = ops.colsums(dN1, H1, f1[0], f1[1], out0=G[c + "2.bias"], out1=G[c + "2.weight"])
            dH1 = ops.bn2d_bwd(dN1, H1, glob(l1), n, P[c + "2.weight"], f1[0], f1[1], True, count_dev=cd(4))
            G[c + "0.weight"] = ops.dense_wgrad(dH1, S["pooled"], newg(c + "0.weight"))
            G[c + "0.bias"] = newg(c + "0.bias"); ops.colsums(dH1, out0=G[c + "0.bias"])
            return ops.dense(dH1, P[c + "0.weight"], None, 256, 128, transpose_w=True)

        # ======================= sex classifier: pooling + TDNN (its input gradient is an addend
        # of the encoder-output gradient that the last decoder dgrad fuses in) =======================
        def tdnn_bwd(dP):
            t = "sex_classifier.tdnn."
            g, st = ops.pool_bwd(r2, bn2[2], bn2[3], dP, S["pmean"], S["psd"], bn=(bn2[0], bn2[1]))
            if model.store_bf16_probe in (2, 3):
                _round_first(g)
            g = bn_finish(g, st, r2, bn2, Lc, t + "8", t + "6.bias", 3)
            conv_wgrad(t + "6.weight", r1, g, 128, 128, 1, Lc, 3, 3, 0, s2=bn1[2], t2=bn1[3])
            g, st = cg(g, pw(t + "6.weight", "conv_dgrad"), None, 128, 128, 1, 1,
                       ops.taps_conv_dgrad_s1(3, 3, 0), Lb, want_stats=True, ep=bn_ep(r1, bn1))
            g = bn_finish(g, st, r1, bn1, Lb, t + "5", t + "3.bias", 2)
            conv_wgrad(t + "3.weight", r0, g, 128, 128, 1, Lb, 3, 2, 0, s2=bn0[2], t2=bn0[3])
            g, st = cg(g, pw(t + "3.weight", "conv_dgrad"), None, 128, 128, 1, 1,
                       ops.taps_conv_dgrad_s1(3, 2, 0), La, want_stats=True, ep=bn_ep(r0, bn0))
            g = bn_finish(g, st, r0, bn0, La, t + "2", t + "0.bias", 1)
            conv_wgrad(t + "0.weight", y4, g, 128, 128, 1, La, 5, 1, 0, s1=n4[2], t1=n4[3], swish=True,
                       s2=bn_n[2], t2=bn_n[3])
            xp4 = (n4[2], n4[3])
            g, st = cg(g, pw(t + "0.weight", "conv_dgrad"), None, 128, 128, 1, 1,
                       ops.taps_conv_dgrad_s1(5, 1, 0), L4, want_stats=True, ep=bn_ep(y4, bn_n, xp4))
            da = bn_finish(g, st, y4, bn_n, L4, "sex_classifier.norm", None, 0, xp=xp4)      # includes GRL
            side_join()
            flush_bias()
            if need_stage["sex_classifier"]:
                buckets.reduce_stage("sex_classifier")
            return da

        # ======================= decoder =======================
        def decoder_bwd():
            d_rec = torch.zeros(B, T, 80, device=dev) if recon_unused else d_recon
            g_rec = d_rec.reshape(B, Ltot).contiguous().float()
            if need["decoder.8.bias"]:
                setg("decoder.8.bias", ops.sum_partials(g_rec.view(4 * B, Ltot // 4), 1, n=Ltot // 4).sum())
            if need["decoder.8.weight"]:
                G["decoder.8.weight"] = ops.wgrad1C(g_rec, y8, newg("decoder.8.weight"), flip=True,
                                                    s1=n8[2], t1=n8[3], swish=True)
            g, st = ops.conv1toC(g_rec, P["decoder.8.weight"], None, dt, flip=True, want_stats=True,
                                 ep=dict(x=y8, s1=n8[2], t1=n8[3], mean=n8[0], rstd=n8[1]))     # d z8
            if model.store_bf16_probe >= 2:
                _round_first(g)
            in_decoder[0] = True
            g = in_finish(g, st, y8, n8, 32, Ltot, "decoder.6", "decoder.5.bias")               # d y8
            convT_wgrad("decoder.5.weight", y7, g, 64, 32, L2)
            g, st = cg(g, pw("decoder.5.weight", "convT_dgrad"), None, 32, 64, 2, 1,
                       ops.taps_convT_dgrad(), L2, want_stats=True)                              # d y7
            bias_from(st, "decoder.4.bias", 64)
            conv_wgrad("decoder.4.weight", y6, g, 64, 64, 1, L2, K5, 1, 2, s1=n6[2], t1=n6[3], swish=True)
            g, st = cg(g, pw("decoder.4.weight", "conv_dgrad"), None, 64, 64, 1, 1,
                       ops.taps_conv_dgrad_s1(K5, 1, 2), L2, want_stats=True, ep=in_ep(y6, n6))  # d z6
            g = in_finish(g, st, y6, n6, 64, L2, "decoder.2", "decoder.1.bias")                 # d y6
            convT_wgrad("decoder.1.weight", y5, g, 128, 64, L4)
            g, st = cg(g, pw("decoder.1.weight", "convT_dgrad"), None, 64, 128, 2, 1,
                       ops.taps_convT_dgrad(), L4, want_stats=True)                              # d y5
            bias_from(st, "decoder.0.bias", 128)
            conv_wgrad("decoder.0.weight", y4, g, 128, 128, 1, L4, K5, 1, 2, s1=n4[2], t1=n4[3], swish=True)
            side_join()
            flush_bias()
            if need_stage["decoder"]:
                buckets.reduce_stage("decoder")
            in_decoder[0] = False
            return g

        hs = (model._head_stream(dev) if (model.overlap_head and run_decoder and not model._bn_syncs()
                                         and dev.type == "cuda") else None)
        if hs is None:
            da4_cls = tdnn_bwd(head_bwd())
            if not run_decoder:                   # classifier-only step: nothing below needs a gradient
                return finish()
            g = decoder_bwd()
        else:
            # the head's ~25 small launches on the side stream beside the decoder's backward
            main = torch.cuda.current_stream()
            hs.wait_stream(main)
            with torch.cuda.stream(hs):
                dP = head_bwd()
            g = decoder_bwd()
            main.wait_stream(hs)
            dP.record_stream(main)
            da4_cls = tdnn_bwd(dP)
        if not run_encoder:
            return finish()

        # ======================= encoder =======================
        # d z4 = (decoder.0 dgrad + classifier branch) * swish'(z4), fused into the dgrad launch
        g, st = cg(g, pw("decoder.0.weight", "conv_dgrad"), None, 128, 128, 1, 1,
                   ops.taps_conv_dgrad_s1(K5, 1, 2), L4, want_stats=True, ep=in_ep(y4, n4, g2=da4_cls))
        g = in_finish(g, st, y4, n4, 128, L4, "encoder.12", "encoder.11.bias")               # d y4
        conv_wgrad("encoder.11.weight", y3, g, 128, 128, 1, L4, K5, 1, 2, s1=n3[2], t1=n3[3], swish=True)
        g, st = cg(g, pw("encoder.11.weight", "conv_dgrad"), None, 128, 128, 1, 1,
                   ops.taps_conv_dgrad_s1(K5, 1, 2), L4, want_stats=True, ep=in_ep(y3, n3))
        g = in_finish(g, st, y3, n3, 128, L4, "encoder.9", "encoder.8.bias")                 # d y3
        conv_wgrad("encoder.8.weight", y2, g, 64, 128, 2, L4, K5, 1, 2, s1=n2[2], t1=n2[3], swish=True)
        g, st = cg(g, pw("encoder.8.weight", "conv_dgrad"), None, 128, 64, 1, 2, ops.UP2, L2,
                   want_stats=True, ep=in_ep(y2, n2))
        g = in_finish(g, st, y2, n2, 64, L2, "encoder.6", "encoder.5.bias")                  # d y2
        conv_wgrad("encoder.5.weight", y1, g, 64, 64, 1, L2, K5, 1, 2, s1=n1[2], t1=n1[3], swish=True)
        g, st = cg(g, pw("encoder.5.weight", "conv_dgrad"), None, 64, 64, 1, 1,
                   ops.taps_conv_dgrad_s1(K5, 1, 2), L2, want_stats=True, ep=in_ep(y1, n1))
        g = in_finish(g, st, y1, n1, 64, L2, "encoder.3", "encoder.2.bias")                  # d y1
        conv_wgrad("encoder.2.weight", y0, g, 32, 64, 2, L2, K5, 1, 2, swish=True)
        g, st = cg(g, pw("encoder.2.weight", "conv_dgrad"), None, 64, 32, 1, 2, ops.UP2, Ltot,
                   want_stats=True, ep=dict(mode=1, x=y0))                                   # d y0
        bias_from(st, "encoder.0.bias", 32)
        if need["encoder.0.weight"]:
            G["encoder.0.weight"] = ops.wgrad1C(S["x0"], g, newg("encoder.0.weight"))
        d_feats = None
        if ctx.need_input_grad:
            d_feats = ops.convCto1(g, P["encoder.0.weight"], None, flip=True).view(B, T, 80)
        side_join()
        flush_bias()
        if need_stage["encoder"]:
            buckets.reduce_stage("encoder")
        buckets.join()
        ctx.S = None
        # (Brain.check_gradients clips the buckets directly when every .grad is still a view of one of them)
        model._last_flats = [buckets.flat[st] for st in sdist.StageBuckets.STAGES if need_stage[st]]
        grads = tuple(G[k] if need[k] else None for k in names)
        # autograd's AccumulateGrad adopts a gradient only when nothing else references it (it
        # clones otherwise: 56 copies per step): drop this frame's references explicitly rather than
        # rely on the frame being collected before the engine looks
        G.clear()
        buckets.views.clear()
        return (None, None, d_feats) + grads
