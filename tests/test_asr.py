"""Frozen-ASR utility branch: PARITY UNPINNED -- speechbrain's TransformerASR /
ConvolutionFrontEnd are not in the reference tree and the weights are a hub fetch, so there is
nothing to compare numbers with.  These are the properties the architecture implies
(speechbrain_configs/convae.yaml:139-182; models/SpeechBrain_ASR.py:16-30,101-103; call sites
speechbrain_convae_train.py:97-102)."""
import pytest
import torch

from speech_anonymization_amd import asr as A


def _small(dtype=torch.float32):
    cnn = A.ConvolutionFrontEnd(out_channels=(8, 16, 32))
    tr = A.TransformerASR(input_size=cnn.out_features, tgt_vocab=50, d_model=64, nhead=4,
                          num_encoder_layers=2, num_decoder_layers=2, d_ffn=128)
    return A.ASR(cnn, tr, output_neurons=50, dtype=dtype)


def test_parameter_count_of_the_reference_configuration():
    """convae.yaml:139-182: CNN + Transformer + seq_lin + ctc_lin = 161.6 M"""
    with torch.device("meta"):
        m = A.ASR()
    n = sum(p.numel() for p in m.parameters())
    assert abs(n - 161.6e6) < 0.3e6, n
    assert not any(p.requires_grad for p in m.parameters())
    assert m.CNN.out_features == 10240                           # Transformer input_size in the YAML


def test_shapes_masks_and_causality():
    torch.manual_seed(0)
    m = _small()
    B, T, U = 3, 72, 7
    feats = torch.randn(B, T, 80)
    lens = torch.tensor([1.0, 0.5, 0.75])
    tok = torch.randint(3, 50, (B, U))
    tok[:, 0] = 1
    tok[2, 5:] = 0                                               # padding
    enc, pred = m.get_predictions(feats, lens, tok, eval=True)
    assert enc.shape == (B, T // 4, 64) and pred.shape == (B, U, 64)
    assert torch.isfinite(enc).all() and torch.isfinite(pred).all()
    # decoder is causal: changing token u changes nothing before u
    tok2 = tok.clone()
    tok2[:, 4] = (tok2[:, 4] + 7) % 47 + 3
    _, pred2 = m.get_predictions(feats, lens, tok2, eval=True)
    assert torch.allclose(pred[:, :4], pred2[:, :4], atol=1e-6)
    assert not torch.allclose(pred[:, 4:], pred2[:, 4:], atol=1e-6)
    # encoder key-padding mask: encoder states of utterance 1 beyond its length do not reach its
    # valid positions through attention -- perturb the transformer input there, not the features
    # (the CNN's receptive field straddles the boundary)
    src = m.CNN(feats)
    src2 = src.clone()
    src2[1, (T // 4) // 2 + 1:] += 1.0
    e1, _ = m.Transformer(src, tok, lens)
    e2, _ = m.Transformer(src2, tok, lens)
    n1 = (T // 4) // 2
    assert torch.allclose(e1[1, :n1], e2[1, :n1], atol=1e-6)
    assert torch.allclose(e1[0], e2[0], atol=1e-6)


def test_gradient_reaches_the_reconstruction_only():
    torch.manual_seed(1)
    m = _small()

    class Cos(torch.nn.Module):                                  # the arithmetic of utils/cosine_similarity_loss.py:53-56
        def forward(self, a, b):
            l = 1 - torch.nn.functional.cosine_similarity(a, b, dim=2, eps=1e-6)
            return l.sum() / l.shape[1]
    B, T, U = 2, 36, 5
    feats = torch.randn(B, T, 80)
    recon = (feats + 0.1 * torch.randn(B, T, 80)).requires_grad_()
    tok = torch.randint(3, 50, (B, U))
    tok[:, 0] = 1
    lens = torch.ones(B)
    loss = A.utility_loss(m, Cos(), feats, recon, lens, tok)
    loss.backward()
    assert recon.grad is not None and float(recon.grad.abs().sum()) > 0
    assert all(p.grad is None for p in m.parameters())
    assert float(loss) > 0
    same = A.utility_loss(m, Cos(), feats, feats.clone().requires_grad_(), lens, tok)
    assert abs(float(same)) < 1e-5                               # identical inputs: zero loss
    m.train()
    assert not m.training                                        # frozen recogniser stays in eval mode
    with pytest.raises(NotImplementedError):
        m.get_predictions(feats, lens, tok, do_ctc=True)


@pytest.mark.gpu
def test_utility_loss_in_the_train_step_on_gpu():
    """the branch inside compute_objectives with the library's sa_cosine_loss as loss_utility: bf16
    recogniser against the same weights in fp32 (loss within 5 %; 1 - cos cancels, so the
    perturbation is large), gradient direction the same"""
    from speech_anonymization_amd.losses import CosineSimilarityLoss
    torch.manual_seed(2)
    m32 = _small().cuda()
    m16 = _small(torch.bfloat16).cuda()
    B, T, U = 4, 72, 6
    feats = torch.randn(B, T, 80, device="cuda")
    tok = torch.randint(3, 50, (B, U), device="cuda")
    tok[:, 0] = 1
    lens = torch.ones(B, device="cuda")
    out = []
    for m in (m32, m16):
        recon = (feats + 1.5 * torch.sin(7 * feats)).requires_grad_()
        loss = A.utility_loss(m, CosineSimilarityLoss(), feats, recon, lens, tok)
        loss.backward()
        out.append((float(loss), recon.grad.clone()))
    assert out[0][0] > 0.01 and abs(out[1][0] - out[0][0]) < 0.05 * out[0][0], (out[0][0], out[1][0])
    g32, g16 = out[0][1], out[1][1]
    cos = float((g32 * g16).sum() / (g32.norm() * g16.norm()))
    assert cos > 0.95, cos


@pytest.mark.gpu
def test_brain_adds_the_utility_term():
    """compute_objectives with an attached recogniser (speechbrain_convae_train.py:97-102,122-128):
    loss = 0.1 recon + 0.9 sex + w * utility, and the utility gradient reaches the decoder"""
    import bench
    from speech_anonymization_amd.brain import Stage
    from speech_anonymization_amd.losses import CosineSimilarityLoss
    dev = torch.device("cuda", 0)
    brain = bench.build_brain(dev, "bf16x3", 2)
    batch = bench.synthetic_batch(2, 0, dev, n_samples=36 * 160 * 2 - 160)
    tok = torch.randint(3, 50, (2, 5), device=dev)
    tok[:, 0] = 1
    batch.tokens_bos = (tok, torch.ones(2, device=dev))
    brain.hparams.loss_utility = CosineSimilarityLoss()
    brain.modules.ConvAE.pooling_noise = None
    brain.hparams.epoch_counter.current = 10         # past update_until_epoch: the normaliser is frozen
    torch.manual_seed(3)
    base = brain.compute_objectives(brain.compute_forward(batch, Stage.TRAIN), batch, Stage.TRAIN)
    brain.asr_brain = _small(torch.bfloat16).to(dev)
    brain.hparams.utility_loss_weight = 0.5
    pred = brain.compute_forward(batch, Stage.TRAIN)
    with_u = brain.compute_objectives(pred, batch, Stage.TRAIN)
    from speech_anonymization_amd import asr as A2
    feats = brain.features(*batch.sig)
    u = A2.utility_loss(brain.asr_brain, brain.hparams.loss_utility, feats, pred[0].detach(),
                        batch.sig[1], tok)
    assert float(u) > 0
    assert abs(float(with_u) - float(base) - 0.5 * float(u)) < 2e-3 * abs(float(with_u)) + 1e-4
    with_u.backward()
    g = brain.modules.ConvAE.decoder[8].weight.grad if hasattr(brain.modules.ConvAE, "decoder") else None
    assert g is None or torch.isfinite(g).all()
    assert any(p.grad is not None and float(p.grad.abs().sum()) > 0 for p in brain.modules.ConvAE.parameters())


@pytest.mark.gpu
def test_utility_retention_reaches_the_valid_stats():
    """speechbrain_convae_train.py:156-163 + :338-343: at VALID with a recogniser attached, every
    batch appends the per-utterance cosine similarity of the recogniser's encoder outputs on the
    reconstructed and on the original features (the library's sa_cosine_loss row output);
    on_stage_end reports their mean as Utility_Retention."""
    import bench
    from speech_anonymization_amd.brain import Stage
    dev = torch.device("cuda", 0)
    brain = bench.build_brain(dev, "bf16x3", 3)
    brain.asr_brain = _small(torch.bfloat16).to(dev)
    brain.modules.ConvAE.pooling_noise = None
    # (a fresh InputNormalization has no statistics: one training-mode pass in an updating epoch
    # fills them, as the epochs before a validation stage do)
    brain.hparams.epoch_counter.current = 1
    brain.features(*bench.synthetic_batch(3, 7, dev, n_samples=36 * 160 * 2 - 160).sig)
    brain.hparams.epoch_counter.current = 10
    brain.modules.eval()
    brain.on_stage_start(Stage.VALID, 1)
    want = []
    for seed in (0, 1):
        batch = bench.synthetic_batch(3, seed, dev, n_samples=36 * 160 * 2 - 160)
        tok = torch.randint(3, 50, (3, 5), device=dev)
        tok[:, 0] = 1
        batch.tokens_bos = (tok, torch.ones(3, device=dev))
        brain.evaluate_batch(batch, Stage.VALID)
        with torch.no_grad():
            feats = brain.features(*batch.sig)
            recon, _ = brain.modules.ConvAE(feats)
            e_r, _ = brain.asr_brain.get_predictions(recon, batch.sig[1], tok, eval=True)
            e_o, _ = brain.asr_brain.get_predictions(feats, batch.sig[1], tok, eval=True)
            want.append(torch.nn.functional.cosine_similarity(e_r.reshape(3, -1).float(), e_o.reshape(3, -1).float(),
                                                              dim=-1, eps=1e-8))
    want = torch.cat(want)
    got = torch.stack(list(brain.utility_similarity_aggregator.scores)) if isinstance(
        brain.utility_similarity_aggregator.scores, list) else brain.utility_similarity_aggregator.scores
    assert got.shape == want.shape and torch.allclose(got, want, atol=2e-5), (got, want)
    brain.train_stats = {"loss": 0.0}
    brain.on_stage_end(Stage.VALID, 0.5, 1)
    assert abs(brain.valid_stats["Utility_Retention"] - float(want.mean())) < 2e-5
    assert 0.0 < brain.valid_stats["Utility_Retention"] < 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("d,rows", [(768, 8064), (256, 37), (1024, 5)])
def test_add_layernorm_kernels_equal_torch(d, rows):
    """sa_add_layernorm_fwd / sa_layernorm_bwd (csrc/sa_asr.hip) against F.layer_norm of the bf16 sum:
    forward within one bf16 rounding, input gradient (the same for both addends) within bf16 noise."""
    import torch.nn.functional as F
    torch.manual_seed(d + rows)
    dev = torch.device("cuda:0")
    x = torch.randn(rows, d, device=dev).bfloat16().requires_grad_()
    r = (0.5 * torch.randn(rows, d, device=dev)).bfloat16().requires_grad_()
    w = (1.0 + 0.1 * torch.randn(d, device=dev)).bfloat16()
    b = (0.1 * torch.randn(d, device=dev)).bfloat16()
    dy = torch.randn(rows, d, device=dev).bfloat16()
    y = A._AddLayerNorm.apply(x, r, w, b, 1e-6)
    y.backward(dy)
    gx, gr = x.grad.clone(), r.grad.clone()
    assert torch.equal(gx, gr)
    xs = (x.detach() + r.detach()).float().requires_grad_()           # the bf16 sum, then fp32 arithmetic
    yr = F.layer_norm(xs, (d,), w.float(), b.float(), 1e-6)
    yr.backward(dy.float())

    def rel(a, bb):
        return float(((a.float() - bb.float()) ** 2).sum() / (bb.float() ** 2).sum())
    assert rel(y, yr) < 2e-5 and rel(gx, xs.grad) < 2e-5              # bf16 rounding of the outputs: ~2^-9 rms
    # no residual, no gradient: plain LayerNorm, nothing saved
    with torch.no_grad():
        y0 = A._AddLayerNorm.apply(x.detach(), None, w, b, 1e-6)
    assert rel(y0, F.layer_norm(x.detach().float(), (d,), w.float(), b.float(), 1e-6)) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 9, 7, 1), (3, 16, 10, 8), (2, 3, 3, 16), (1, 40, 20, 128)])
def test_reflect_pad_kernels_equal_torch(shape):
    """sa_reflect_pad_fwd / _bwd against F.pad(mode="reflect") and its autograd adjoint"""
    import torch.nn.functional as F
    torch.manual_seed(sum(shape))
    dev = torch.device("cuda:0")
    x = torch.randn(*shape, device=dev).bfloat16().requires_grad_()
    y = A._reflect_pad1(x)
    xr = x.detach().float().requires_grad_()
    yr = F.pad(xr.permute(0, 3, 1, 2), (1, 1, 1, 1), mode="reflect").permute(0, 2, 3, 1)
    assert torch.equal(y.float(), yr)
    dy = torch.randn_like(y)
    y.backward(dy)
    yr.backward(dy.float())
    assert float((x.grad.float() - xr.grad).abs().max()) <= 0.02 * float(xr.grad.abs().max())   # bf16 sum of <= 4 terms


@pytest.mark.gpu
def test_fused_attention_equals_the_explicit_form():
    """asr._Attention: torch's fused attention on strided head views ("sdpa", default) against the explicit
    GEMM / softmax form on head-major copies, self- and cross-attention with padding masks, forward and
    input gradient (bf16)."""
    torch.manual_seed(3)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    att = A._Attention(768, 8, g).to(dev).bfloat16()
    B, Tq, Tk = 3, 40, 56
    x = torch.randn(B, Tq, 768, device=dev).bfloat16()
    mem = torch.randn(B, Tk, 768, device=dev).bfloat16()
    pad = torch.zeros(B, Tk, dtype=torch.bool, device=dev)
    pad[1, 30:] = True
    bias = A.TransformerASR._key_bias(pad)
    self_bias = torch.full((Tq, Tq), float("-inf"), device=dev).triu(1)[None, None]
    out = {}
    for impl in ("gemm", "sdpa"):
        att.impl = impl
        xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
        o1 = att(xa, bias=self_bias)
        o2 = att(xb, kv=mem, bias=bias)
        (o1.float().pow(2).sum() + o2.float().pow(2).sum()).backward()
        out[impl] = (o1, o2, xa.grad, xb.grad)
    A._Attention.impl = "sdpa"

    def rel(a, b):
        return float(((a.float() - b.float()) ** 2).sum() / (b.float() ** 2).sum())
    for a, b in zip(out["sdpa"], out["gemm"]):
        assert rel(a, b) < 1e-3, rel(a, b)                            # two bf16 pipelines: ~1e-5 .. 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("fc", [(40, 128), (20, 512)])
def test_layernorm_leakyrelu_kernels_equal_torch(fc):
    """sa_ln_leaky_fwd / _bwd (the front end's LayerNorm over (frequency, channel) + LeakyReLU) against
    F.layer_norm + F.leaky_relu in fp32 on the same bf16 input"""
    import torch.nn.functional as F
    torch.manual_seed(fc[0])
    dev = torch.device("cuda:0")
    x = torch.randn(3, 11, *fc, device=dev).bfloat16().requires_grad_()
    w = (1.0 + 0.1 * torch.randn(*fc, device=dev)).bfloat16()
    b = (0.2 * torch.randn(*fc, device=dev)).bfloat16()
    dy = torch.randn(3, 11, *fc, device=dev).bfloat16()
    y = A._LNLeaky.apply(x, w, b, 1e-5, 0.01)
    y.backward(dy)
    xr = x.detach().float().requires_grad_()
    yr = F.leaky_relu(F.layer_norm(xr, fc, w.float(), b.float(), 1e-5), 0.01)
    yr.backward(dy.float())

    def rel(a, bb):
        return float(((a.float() - bb.float()) ** 2).sum() / (bb.float() ** 2).sum())
    assert rel(y, yr) < 2e-5 and rel(x.grad, xr.grad) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(3, 72), (2, 37), (1, 1008)])
def test_front_end_block0_kernel_equals_the_library_path(B, T):
    """sa_asr_block0_fwd / _bwd (reflect-padded Conv2d 1 -> 128, 3 x 3, stride 2 + LayerNorm + LeakyReLU in one
    pass each way, the convolution recomputed in the backward) against F.pad / F.conv2d / F.layer_norm /
    F.leaky_relu in fp32 on the same bf16 operands; even and odd T (the reflection at both ends)."""
    import torch.nn.functional as F
    torch.manual_seed(B * 100 + T)
    dev = torch.device("cuda:0")
    cnn = A.ConvolutionFrontEnd().to(dev).bfloat16()
    with torch.no_grad():
        cnn.b[0].copy_(0.1 * torch.randn(128))
        cnn.ln_w[0].copy_(1.0 + 0.1 * torch.randn(40, 128))
        cnn.ln_b[0].copy_(0.1 * torch.randn(40, 128))
    x = torch.randn(B, T, 80, device=dev).bfloat16().requires_grad_()
    y = A._Block0.apply(x, cnn.w[0], cnn.b[0], cnn.ln_w[0], cnn.ln_b[0], 1e-5, 0.01)
    dy = torch.randn_like(y)
    y.backward(dy)
    xr = x.detach().float().requires_grad_()
    h = F.pad(xr.unsqueeze(1), (1, 1, 1, 1), mode="reflect")
    z = F.conv2d(h, cnn.w[0].float(), cnn.b[0].float(), stride=2).permute(0, 2, 3, 1)
    yr = F.leaky_relu(F.layer_norm(z, (40, 128), cnn.ln_w[0].float(), cnn.ln_b[0].float(), 1e-5), 0.01)
    yr.backward(dy.float())
    assert y.shape == yr.shape

    def rel(a, b):
        return float(((a.float() - b.float()) ** 2).sum() / (b.float() ** 2).sum())
    assert rel(y, yr) < 5e-5, rel(y, yr)                  # bf16 roundings of z and y
    assert rel(x.grad, xr.grad) < 2e-4, rel(x.grad, xr.grad)   # + of d z and d x
    # the whole front end takes the kernel for its first block and gives the same features as without it
    feats = torch.randn(B, T, 80, device=dev).bfloat16()
    with torch.no_grad():
        out = cnn(feats)
        k0 = cnn.w[0]
        cnn.w[0] = torch.nn.Parameter(k0.float(), requires_grad=False)      # (dtype mismatch: library path)
        ref = cnn(feats)
        cnn.w[0] = k0
    assert rel(out, ref) < 1e-3
