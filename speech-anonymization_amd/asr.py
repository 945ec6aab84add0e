"""Frozen-ASR utility branch (BASELINE config 4, second half).

Reference: models/SpeechBrain_ASR.py:16-46 (ASR.compute_forward: CNN -> Transformer, `do_ctc=False`
returns (enc_out, pred)), :101-103 (get_predictions), call sites
speechbrain_convae_train.py:97-103: the pretrained recogniser runs on the original and on the
reconstructed features and `loss_utility(recon_prob, orig_prob)` (CosineSimilarityLoss) pulls the
decoder states together.  Architecture: speechbrain_configs/convae.yaml:139-158 -- ConvolutionFrontEnd
(3 blocks, 128 / 256 / 512 channels, kernels 3 / 3 / 1, strides 2 / 2 / 1) -> 10240 -> TransformerASR
(d_model 768, 8 heads, 12 encoder + 6 decoder layers, d_ffn 3072, GELU, post-norm), 161.6 M
parameters with ctc_lin / seq_lin.

PARITY UNPINNED, throughput-only: the two classes live in speechbrain (an empty submodule of the
reference) and the weights are a hub fetch; the reference holds no vector, checkpoint or key list
for them.  What is restated here from speechbrain's published architecture [SB-MEM] is the layer
graph; what can be checked is checked as properties (tests/test_asr.py: parameter count 161.6 M,
causality of the decoder, padding masks, the gradient reaching the features and nothing else, zero
loss for identical inputs).

MI355X shape of it: a dense transformer is plain library GEMMs (hipBLASLt through torch; nothing
here is a hand-written kernel).  The weights are FROZEN, so they are held once in bf16 (323 MB,
resident in HBM), QKV packed into one [2304, 768] GEMM, no weight-gradient GEMM ever runs (only the
data gradient of the reconstruction branch), and the original-features branch runs under no_grad
(no activations kept).  The loss itself is the library's sa_cosine_loss (losses.CosineSimilarityLoss).
"""
import math
import os

import torch
import torch.nn.functional as F


def _frozen(t):
    return torch.nn.Parameter(t, requires_grad=False)


def _hip(x):
    """the hand-written passes (csrc/sa_asr.hip) take contiguous bf16 tensors on the GPU"""
    return x.is_cuda and x.dtype == torch.bfloat16


class _ReflectPad(torch.autograd.Function):
    """sa_reflect_pad_fwd / _bwd: one gather each way"""

    @staticmethod
    def forward(ctx, x):
        from . import ops
        return ops.reflect_pad(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        return ops.reflect_pad_bwd(dy.contiguous())


def _reflect_pad1(x):
    """F.pad(., (1, 1, 1, 1), mode="reflect") of the time and frequency dims of a [B, T, F, C]
    tensor.  On the GPU in bf16: the library's gather kernel and its adjoint.  Otherwise two
    concatenations: the backward of torch's reflection_pad2d is an atomic scatter (1.7 ms per call on
    the [B, 504, 40, 128] tensor of block 2); the backward of a concatenation is two narrow adds."""
    if _hip(x):
        return _ReflectPad.apply(x)
    x = torch.cat([x[:, :, 1:2], x, x[:, :, -2:-1]], dim=2)
    return torch.cat([x[:, 1:2], x, x[:, -2:-1]], dim=1)


class _AddLayerNorm(torch.autograd.Function):
    """norm(x + r) of a post-norm layer as one pass forward (sa_add_layernorm_fwd) and one backward
    (sa_layernorm_bwd): its result is the gradient of both addends; gamma / beta are frozen."""

    @staticmethod
    def forward(ctx, x, r, gamma, beta, eps):
        from . import ops
        save = any(ctx.needs_input_grad[:2])
        y, s, stat = ops.add_layernorm(x.contiguous(), None if r is None else r.contiguous(), gamma, beta, eps, save)
        if save:
            ctx.save_for_backward(s, stat, gamma)
        ctx.has_r = r is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        s, stat, gamma = ctx.saved_tensors
        ds = ops.layernorm_bwd(dy.contiguous(), s, stat, gamma)
        return ds, (ds if ctx.has_r else None), None, None, None


class ConvolutionFrontEnd(torch.nn.Module):
    """[SB-MEM] speechbrain.lobes.models.convolution.ConvolutionFrontEnd with the YAML's arguments:
    per block Conv2d ("same" reflect padding, stride s on time AND frequency) -> LayerNorm over
    (freq, channels) -> LeakyReLU; [B, T, 80] -> [B, ceil(T/4), 20, 512]."""

    def __init__(self, input_shape=(8, 10, 80), num_blocks=3, num_layers_per_block=1,
                 out_channels=(128, 256, 512), kernel_sizes=(3, 3, 1), strides=(2, 2, 1),
                 residuals=(False, False, False), **_):
        super().__init__()
        assert num_layers_per_block == 1 and not any(residuals)
        self.kernel_sizes, self.strides = tuple(kernel_sizes), tuple(strides)
        cin, f = 1, input_shape[-1]
        g = torch.Generator().manual_seed(1234)
        self.w, self.b, self.ln_w, self.ln_b = (torch.nn.ParameterList() for _ in range(4))
        for i in range(num_blocks):
            k, s, co = kernel_sizes[i], strides[i], out_channels[i]
            f = -(-f // s)
            self.w.append(_frozen(torch.randn(co, cin, k, k, generator=g) / math.sqrt(cin * k * k)))
            self.b.append(_frozen(torch.zeros(co)))
            self.ln_w.append(_frozen(torch.ones(f, co)))
            self.ln_b.append(_frozen(torch.zeros(f, co)))
            cin = co
        self.out_features = f * cin

    def forward(self, x):
        """activations stay [B, T, F, C] in memory (channels-last: what MIOpen's NHWC implicit-GEMM
        kernels take); the convolution sees the logical NCHW view of the same bytes"""
        first = 0
        if (_hip(x) and x.shape[-1] == 80 and x.shape[1] >= 3 and self.w[0].shape == (128, 1, 3, 3)
                and self.strides[0] == 2 and self.w[0].dtype == x.dtype):
            x = _Block0.apply(x, self.w[0], self.b[0], self.ln_w[0], self.ln_b[0], 1e-5, 0.01)
            first = 1
        else:
            x = x.unsqueeze(-1)                                   # [B, T, F, 1]
        for i, (k, s) in enumerate(zip(self.kernel_sizes, self.strides)):
            if i < first:
                continue
            if k > 1:
                x = _reflect_pad1(x)
            w = self.w[i].to(x.dtype)
            if x.shape[-1] == 1:          # one channel: NCHW and NHWC are the same bytes; canonical strides
                xin = x.reshape(x.shape[0], 1, x.shape[1], x.shape[2])
            else:
                xin, w = x.permute(0, 3, 1, 2), w.contiguous(memory_format=torch.channels_last)
            x = F.conv2d(xin, w, self.b[i].to(x.dtype), stride=s)
            x = x.permute(0, 2, 3, 1).contiguous()                # [B, T', F', C] (no copy when NHWC)
            lw, lb = self.ln_w[i], self.ln_b[i]
            if _hip(x) and lw.dtype == x.dtype and lw.numel() in (5120, 10240):
                x = _LNLeaky.apply(x, lw, lb, 1e-5, 0.01)
            else:
                x = F.leaky_relu(F.layer_norm(x, x.shape[2:], lw.to(x.dtype), lb.to(x.dtype), 1e-5), 0.01)
        return x                                                  # [B, T', F', C]


class _LNLeaky(torch.autograd.Function):
    """the front end's LayerNorm over (frequency, channel) + LeakyReLU: one pass each way
    (sa_ln_leaky_fwd / _bwd); keeps the convolution's output and two floats per row"""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, slope):
        from . import ops
        x = x.contiguous()
        save = ctx.needs_input_grad[0]
        y, stat = ops.ln_leaky(x, gamma, beta, eps, slope, save)
        if save:
            ctx.save_for_backward(x, stat, gamma, beta)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, stat, gamma, beta = ctx.saved_tensors
        return ops.ln_leaky_bwd(dy.contiguous(), x, stat, gamma, beta, ctx.slope), None, None, None, None


class _Block0(torch.autograd.Function):
    """block 0 of the front end (Conv2d 1 -> 128, 3 x 3, stride 2, reflect padding + LayerNorm + LeakyReLU)
    as one pass each way (sa_asr_block0_fwd / _bwd): nine multiply-adds per output, bound by its output"""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, eps, slope):
        from . import ops
        x = x.contiguous()
        save = ctx.needs_input_grad[0]
        y, stat = ops.asr_block0(x, w, b, gamma, beta, eps, slope, save)
        if save:
            ctx.save_for_backward(x, w, b, gamma, beta, stat)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, w, b, gamma, beta, stat = ctx.saved_tensors
        return (ops.asr_block0_bwd(dy.contiguous(), x, w, b, gamma, beta, stat, ctx.slope),) + (None,) * 6


def _sdpa_mask(bias, B, h, Tq, Tk, dtype):
    """additive mask [B | 1, 1, Tq | 1, Tk] -> what the fused attention kernel takes without copying it
    again: materialised [B, h, Tq, Tk] in the operand dtype inside a buffer whose rows are padded to a
    multiple of 8 elements (torch pads an unaligned mask on EVERY call otherwise: 18 us x 100 per step).
    Built once per forward and shared by the layers (the caller caches it on the bias tensor)."""
    cache = bias.__dict__.setdefault("_sa_sdpa", {}) if hasattr(bias, "__dict__") else {}
    key = (B, h, Tq, Tk, dtype)
    m = cache.get(key)
    if m is None:
        buf = torch.empty(B, h, Tq, -(-Tk // 8) * 8, dtype=dtype, device=bias.device)
        m = buf[..., :Tk]
        m.copy_(bias.expand(B, h, Tq, Tk))
        cache[key] = m
    return m


class _Attention(torch.nn.Module):
    """multi-head attention as library GEMMs (packed in-projection, scores, context, out-projection);
    the softmax accumulates in fp32.  kv=None: self-attention with the QKV GEMM packed.  The weights
    are frozen, so the 1/sqrt(d_head) of the scores is folded into the query rows of the
    in-projection once, at construction."""

    # "gemm": scores / softmax / context as separate launches on head-major copies of q, k, v;
    # "sdpa": torch's fused attention (a library kernel, like the GEMMs) on strided head VIEWS of the
    # packed projection -- no head-major copies, no [B, h, Tq, Tk] score tensor in HBM
    impl = os.environ.get("SA_ASR_ATTN", "sdpa")

    def __init__(self, d, nhead, g):
        super().__init__()
        self.d, self.h = d, nhead
        in_w = torch.randn(3 * d, d, generator=g) / math.sqrt(d)
        in_w[:d] *= 1.0 / math.sqrt(d // nhead)
        self.in_w = _frozen(in_w)
        self.in_b = _frozen(torch.zeros(3 * d))
        self.out_w = _frozen(torch.randn(d, d, generator=g) / math.sqrt(d))
        self.out_b = _frozen(torch.zeros(d))

    def forward(self, x, kv=None, bias=None):
        B, Tq, d = x.shape
        h, dh = self.h, d // self.h
        if self.impl == "sdpa" and x.is_cuda:
            # unbind, not three selects: its backward is ONE stack into the packed gradient (a select's is a
            # zero-filled tensor per operand plus the adds that merge them)
            if kv is None:
                q, k, v = (t.transpose(1, 2) for t in F.linear(x, self.in_w, self.in_b).view(B, Tq, 3, h, dh).unbind(2))
            else:
                Tk = kv.shape[1]
                q = F.linear(x, self.in_w[:d], self.in_b[:d]).view(B, Tq, h, dh).transpose(1, 2)
                k, v = (t.transpose(1, 2) for t in F.linear(kv, self.in_w[d:], self.in_b[d:]).view(B, Tk, 2, h, dh).unbind(2))
            m = None if bias is None else _sdpa_mask(bias, B, h, Tq, k.shape[2], q.dtype)
            o = F.scaled_dot_product_attention(q, k, v, attn_mask=m, scale=1.0)   # the scale sits in the query weights
            return F.linear(o.transpose(1, 2).reshape(B, Tq, d), self.out_w, self.out_b)
        if kv is None:                                            # one head-major copy for q, k, v
            q, k, v = F.linear(x, self.in_w, self.in_b).view(B, Tq, 3, h, dh).permute(2, 0, 3, 1, 4).contiguous()
        else:
            Tk = kv.shape[1]
            q = F.linear(x, self.in_w[:d], self.in_b[:d]).view(B, Tq, h, dh).transpose(1, 2)
            k, v = F.linear(kv, self.in_w[d:], self.in_b[d:]).view(B, Tk, 2, h, dh).permute(2, 0, 3, 1, 4).contiguous()
        s = torch.matmul(q, k.transpose(-1, -2))                  # the scale sits in the query weights
        if bias is not None:
            s = s + bias.to(s.dtype)                              # [B | 1, 1, Tq | 1, Tk], 0 / -inf
        p = torch.softmax(s, dim=-1)                              # fp32 accumulation inside the kernel
        o = torch.matmul(p, v).transpose(1, 2).reshape(B, Tq, d)
        return F.linear(o, self.out_w, self.out_b)


class _FFN(torch.nn.Module):
    def __init__(self, d, d_ffn, g):
        super().__init__()
        self.w1 = _frozen(torch.randn(d_ffn, d, generator=g) / math.sqrt(d))
        self.b1 = _frozen(torch.zeros(d_ffn))
        self.w2 = _frozen(torch.randn(d, d_ffn, generator=g) / math.sqrt(d_ffn))
        self.b2 = _frozen(torch.zeros(d))

    def forward(self, x):
        return F.linear(F.gelu(F.linear(x, self.w1, self.b1)), self.w2, self.b2)


class _LN(torch.nn.Module):
    """LayerNorm(x + r) (r optional): the residual add and the normalisation in one pass"""

    def __init__(self, d):
        super().__init__()
        self.w, self.b = _frozen(torch.ones(d)), _frozen(torch.zeros(d))

    def forward(self, x, r=None):
        if _hip(x) and x.shape[-1] in (256, 512, 768, 1024) and self.w.dtype == x.dtype:
            return _AddLayerNorm.apply(x, r, self.w, self.b, 1e-6)
        return F.layer_norm(x if r is None else x + r, x.shape[-1:], self.w, self.b, 1e-6)


class _EncLayer(torch.nn.Module):
    def __init__(self, d, nhead, d_ffn, g):
        super().__init__()
        self.att, self.ffn, self.n1, self.n2 = _Attention(d, nhead, g), _FFN(d, d_ffn, g), _LN(d), _LN(d)

    def forward(self, x, bias):                                   # post-norm (normalize_before: False)
        x = self.n1(x, self.att(x, bias=bias))
        return self.n2(x, self.ffn(x))


class _DecLayer(torch.nn.Module):
    def __init__(self, d, nhead, d_ffn, g):
        super().__init__()
        self.self_att, self.cross_att = _Attention(d, nhead, g), _Attention(d, nhead, g)
        self.ffn, self.n1, self.n2, self.n3 = _FFN(d, d_ffn, g), _LN(d), _LN(d), _LN(d)

    def forward(self, y, mem, self_bias, mem_bias):
        y = self.n1(y, self.self_att(y, bias=self_bias))
        y = self.n2(y, self.cross_att(y, kv=mem, bias=mem_bias))
        return self.n3(y, self.ffn(y))


def _sine_positions(T, d, device):
    pos = torch.arange(T, device=device, dtype=torch.float32).unsqueeze(1)
    den = torch.exp(torch.arange(0, d, 2, device=device, dtype=torch.float32) * (-math.log(10000.0) / d))
    pe = torch.zeros(T, d, device=device)
    pe[:, 0::2], pe[:, 1::2] = torch.sin(pos * den), torch.cos(pos * den)
    return pe


class TransformerASR(torch.nn.Module):
    """[SB-MEM] speechbrain.lobes.models.transformer.TransformerASR.TransformerASR with the YAML's
    arguments.  forward(src [B, T', F', C], tgt [B, U] token ids, wav_len [B] relative lengths,
    pad_idx) -> (encoder_out [B, T', d], decoder_out [B, U, d])."""

    def __init__(self, input_size=10240, tgt_vocab=5000, d_model=768, nhead=8, num_encoder_layers=12,
                 num_decoder_layers=6, d_ffn=3072, dropout=0.0, activation=None,
                 normalize_before=False, seed=4321, **_):
        super().__init__()
        assert not normalize_before, "the reference config is post-norm"
        g = torch.Generator().manual_seed(seed)
        self.d_model = d_model
        self.src_w = _frozen(torch.randn(d_model, input_size, generator=g) / math.sqrt(input_size))
        self.src_b = _frozen(torch.zeros(d_model))
        self.emb = _frozen(torch.randn(tgt_vocab, d_model, generator=g) / math.sqrt(d_model))
        self.enc = torch.nn.ModuleList(_EncLayer(d_model, nhead, d_ffn, g) for _ in range(num_encoder_layers))
        self.dec = torch.nn.ModuleList(_DecLayer(d_model, nhead, d_ffn, g) for _ in range(num_decoder_layers))
        self.enc_norm, self.dec_norm = _LN(d_model), _LN(d_model)

    @staticmethod
    def _key_bias(pad_mask):
        """[B, Tk] bool (True = padding) -> additive [B, 1, 1, Tk]"""
        return torch.zeros(pad_mask.shape, device=pad_mask.device).masked_fill(pad_mask, float("-inf"))[:, None, None, :]

    def forward(self, src, tgt, wav_len=None, pad_idx=0):
        B, T = src.shape[:2]
        x = src.reshape(B, T, -1)
        dt = self.src_w.dtype
        x = F.linear(x.to(dt), self.src_w, self.src_b)
        x = x + _sine_positions(T, self.d_model, x.device).to(dt)
        mem_bias = None
        if wav_len is not None:
            n = torch.round(wav_len.to(x.device).float() * T).long()
            mem_bias = self._key_bias(torch.arange(T, device=x.device)[None, :] >= n[:, None])
        for layer in self.enc:
            x = layer(x, mem_bias)
        enc_out = self.enc_norm(x)

        U = tgt.shape[1]
        y = self.emb[tgt] * math.sqrt(self.d_model)              # NormalizedEmbedding
        y = y + _sine_positions(U, self.d_model, y.device).to(dt)
        causal = torch.full((U, U), float("-inf"), device=y.device).triu(1)[None, None]
        self_bias = causal + self._key_bias(tgt == pad_idx)     # key 0 is <bos>: no row is fully masked
        for layer in self.dec:
            y = layer(y, enc_out, self_bias, mem_bias)
        return enc_out, self.dec_norm(y)


class ASR(torch.nn.Module):
    """The seam of models/SpeechBrain_ASR.py: `get_predictions(feats, wav_lens, tokens_bos, batch,
    eval=False, do_ctc=False)` -> (enc_out, pred).  Frozen and in eval mode by construction
    (on_evaluate_start, :96-99); only `do_ctc=False` is on the training path
    (speechbrain_convae_train.py:98-99) -- the CTC / beam-search half (:31-46) is the reference's
    evaluation control plane and stays out of scope."""

    def __init__(self, cnn=None, transformer=None, pad_index=0, dtype=None, output_neurons=5000):
        super().__init__()
        self.CNN = cnn if cnn is not None else ConvolutionFrontEnd()
        self.Transformer = transformer if transformer is not None else TransformerASR(
            input_size=self.CNN.out_features, tgt_vocab=output_neurons)
        d = self.Transformer.d_model
        # ctc_lin / seq_lin (convae.yaml:176-182): part of asr_model's 161.6 M parameters, unused
        # with do_ctc=False
        self.ctc_lin_w, self.ctc_lin_b = _frozen(torch.zeros(output_neurons, d)), _frozen(torch.zeros(output_neurons))
        self.seq_lin_w, self.seq_lin_b = _frozen(torch.zeros(output_neurons, d)), _frozen(torch.zeros(output_neurons))
        self.pad_index = pad_index
        if dtype is not None:
            self.to(dtype)
        self.eval()

    def train(self, mode=True):                                   # frozen: always eval
        return super().train(False)

    def compute_forward(self, feats, wav_lens, tokens_bos, batch=None, stage=None, do_ctc=False):
        if do_ctc:
            raise NotImplementedError("CTC / beam search (models/SpeechBrain_ASR.py:31-46) is the "
                                      "reference's evaluation path, out of scope (SURVEY 8f-2)")
        dt = self.Transformer.src_w.dtype
        src = self.CNN(feats.to(dt))
        return self.Transformer(src, tokens_bos, wav_lens, pad_idx=self.pad_index)

    def get_predictions(self, feats, wav_lens, tokens_bos, batch=None, eval=False, do_ctc=False):
        if eval:
            with torch.no_grad():
                return self.compute_forward(feats, wav_lens, tokens_bos, batch, do_ctc=do_ctc)
        return self.compute_forward(feats, wav_lens, tokens_bos, batch, do_ctc=do_ctc)


def utility_loss(asr, loss_utility, feats, reconstructed, wav_lens, tokens_bos, batch=None):
    """speechbrain_convae_train.py:97-102.  The reference runs both branches with grad enabled; the
    original-features branch reaches no trainable parameter (the recogniser is frozen, feats has no
    grad), so it runs under no_grad here -- same loss, same gradient, no activations kept."""
    _, orig_prob = asr.get_predictions(feats.detach(), wav_lens, tokens_bos, batch, eval=True)
    _, recon_prob = asr.get_predictions(reconstructed, wav_lens, tokens_bos, batch, eval=False)
    return loss_utility(recon_prob.float(), orig_prob.float())
