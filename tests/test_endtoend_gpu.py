"""SURVEY 8(f)-1 / BASELINE config 4 (first half): models/EndToEnd.py:ConvReconstruction with the
frozen x-vector gender classifier inside the training graph and the adversarial-sign loss
(speechbrain_convae_train.py:111-121), through the C ABI, against
 (1) golden vectors generated from the reference's own ConvReconstruction class
     (tests/golden/endtoend_S.npz, oracle/gen_golden.py: its EncoderClassifier.from_hparams -- absolute
     paths on the authors' machine -- returns the oracle x-vector with seeded weights), and
 (2) the CPU oracle run live (oracle/endtoend.py), full tensors."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_mse(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float(((a - b) ** 2).sum() / (b ** 2).sum().clamp_min(1e-30))


def oracle_classifier(seed=1230):
    from oracle import endtoend as OE
    clf = OE.OracleEncoderClassifier()
    clf.load_state_dict(OE.numpy_params(clf, seed))
    return clf.eval()


def hip_classifier(oclf, pooling_noise=None):
    from speech_anonymization_amd import xvector as HX
    enc = HX.EncoderClassifier(HX.Xvector(pooling_noise=pooling_noise), HX.Classifier(input_shape=[None, None, 128]))
    enc.embedding_model.load_state_dict(oclf.embedding_model.state_dict())
    enc.classifier.load_state_dict(oclf.classifier.state_dict())
    return enc.to(DEV).eval()


@pytest.mark.parametrize("cfg", [(80, 512, 5, 1), (512, 512, 3, 2), (512, 512, 3, 3), (512, 512, 1, 1), (512, 1500, 1, 1)],
                         ids=["80-512k5", "512-512k3d2", "512-512k3d3", "512-512k1", "512-1500k1"])
def test_tdnn_block_input_gradient(cfg):
    """one frozen TDNN block (speechbrain Conv1d 'same' reflect padding -> LeakyReLU -> BatchNorm(eval)):
    sa_tdnn_bwd_input + sa_tdnn_fold against autograd through the oracle's layers (fp64)."""
    from oracle import xvector as OX
    from speech_anonymization_amd import xvector as HX
    cin, cout, k, d = cfg
    B, T = 2, 150
    g = torch.Generator().manual_seed(cin + cout + k + d)
    oc, ob = OX.Conv1d(cin, cout, k, d), OX.BatchNorm1d(cout)
    with torch.no_grad():
        ob.norm.running_mean.normal_(0, 0.3, generator=g); ob.norm.running_var.uniform_(0.5, 1.5, generator=g)
        ob.norm.weight.uniform_(0.8, 1.2, generator=g); ob.norm.bias.normal_(0, 0.1, generator=g)
    oc.eval(); ob.eval()
    x = torch.randn(B, T, cin, generator=g)
    gy = torch.randn(B, T, cout, generator=g)
    import copy
    hc, hb = HX._Conv(cin, cout, k, d), HX._BN(cout)
    hc.load_state_dict(oc.state_dict()); hb.load_state_dict(ob.state_dict())
    hc.to(DEV); hb.to(DEV).eval()
    xd, gyd = x.to(DEV), gy.to(DEV)
    yh, mask = HX._tdnn(xd, hc, hb, want_mask=True)
    dx = HX._tdnn_bwd(gyd, mask, hc, hb)
    torch.cuda.synchronize()
    ocd, obd = copy.deepcopy(oc).double(), copy.deepcopy(ob).double()
    xo = x.double().requires_grad_(True)
    z = ocd(xo)
    y = obd(torch.nn.functional.leaky_relu(z, 0.01))
    s_bn = (obd.norm.weight / torch.sqrt(obd.norm.running_var + obd.norm.eps)).detach()
    assert rel_mse(yh, y) < 1e-9
    # the LeakyReLU branch the forward took: equal to sign(z) of the fp64 oracle except where z is
    # within rounding distance of 0 (the derivative jumps there: a handful of the 10^5 elements may
    # legitimately land on the other side, in any fp32 implementation)
    m = mask.cpu().bool()
    zz = z.detach()
    assert bool((m == (zz > 0))[zz.abs() > 1e-4].all())
    assert int((m != (zz > 0)).sum()) <= 20
    # the arithmetic, given that branch: d z = d y * s * (mask ? 1 : slope), then the adjoint of the
    # reflect-padded dilated convolution (fp64 autograd through the oracle's Conv1d)
    gz = gy.double() * s_bn.double() * torch.where(m, 1.0, 0.01)
    (gx,) = torch.autograd.grad(z, xo, gz)
    assert dx.shape == x.shape
    e = rel_mse(dx, gx)
    eb = rel_mse(dx[:, :8], gx[:, :8]) + rel_mse(dx[:, -8:], gx[:, -8:])      # the reflected ends
    assert e < 1e-9 and eb < 1e-9, (e, eb)


def test_time_pool_and_head_backward():
    """sa_time_pool_bwd (length-masked mean / unbiased std) and sa_leaky_affine_bwd vs autograd"""
    import ctypes as C
    from oracle.convae import StatisticsPooling
    from speech_anonymization_amd import _lib as L
    lib = L.load()
    B, T, Cc = 3, 77, 96
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, T, Cc, generator=g)
    lens = torch.tensor([1.0, 0.6, 0.35])
    gp = torch.randn(B, 2 * Cc, generator=g)
    xo = x.double().requires_grad_(True)
    pooled = StatisticsPooling()(xo, lengths=lens).squeeze(1)
    (gx,) = torch.autograd.grad(pooled, xo, gp.double())
    xd, gd = x.to(DEV), gp.to(DEV)
    out = torch.empty(B, 2 * Cc, device=DEV)
    ld = lens.to(DEV)
    L.check(lib.sa_time_pool(L.ptr(xd), L.ptr(ld), None, B, T, Cc, C.c_float(1e-5), L.ptr(out), L.stream()), "pool")
    dx = torch.empty_like(xd)
    L.check(lib.sa_time_pool_bwd(L.ptr(xd), L.ptr(ld), L.ptr(gd), L.ptr(out), B, T, Cc, C.c_float(1e-5),
                                 L.ptr(dx), L.stream()), "pool_bwd")
    torch.cuda.synchronize()
    assert rel_mse(out, pooled) < 1e-10 and rel_mse(dx, gx) < 1e-9, rel_mse(dx, gx)
    v = torch.randn(5, 64, generator=g); s = torch.rand(64, generator=g) + 0.5; dy = torch.randn(5, 64, generator=g)
    vo = v.clone().requires_grad_(True)
    (gv,) = torch.autograd.grad(torch.nn.functional.leaky_relu(vo, 0.01) * s, vo, dy)
    dv = torch.empty(5, 64, device=DEV)
    dyd, vd, sd = dy.to(DEV), v.to(DEV), s.to(DEV)          # (held: the kernel takes raw pointers)
    L.check(lib.sa_leaky_affine_bwd(L.ptr(dyd), L.ptr(vd), L.ptr(sd), C.c_float(0.01), 5, 64,
                                    L.ptr(dv), L.stream()), "leaky_bwd")
    torch.cuda.synchronize()
    assert rel_mse(dv, gv) < 1e-12


def test_xvector_input_gradient_matches_oracle():
    """d log_probs / d feats through five frozen TDNN blocks (reflect-padded dilated convolutions,
    LeakyReLU, eval-mode BatchNorm), length-masked statistics pooling and the classifier head."""
    from oracle.features import synthetic_feats
    oclf = oracle_classifier()
    enc = hip_classifier(oclf)
    B, T = 3, 90
    feats = synthetic_feats(B, T, seed=6)
    lens = torch.tensor([1.0, 0.77, 0.5])
    w = torch.randn(B, 2, generator=torch.Generator().manual_seed(3))
    fo = feats.double().requires_grad_(True)              # fp64 oracle: the gradient passes 5 layers
    oclf.double()                                         # of 512-1500-channel sums
    o_logp, _, o_idx = oclf(fo, lens)
    (o_logp * w.double()).sum().backward()
    oclf.float()
    fh = feats.to(DEV).requires_grad_(True)
    logp, score, idx = enc(fh, lens)
    (logp * w.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    assert rel_mse(logp, o_logp) < 1e-9 and torch.equal(idx.cpu(), o_idx)
    assert fh.grad.shape == feats.shape
    # The per-block arithmetic is exact to 1e-9 given the LeakyReLU branches (test above).  End to
    # end the 2.3 M pre-activations include a few within rounding distance of 0 whose branch -- and
    # with it a factor 100 in one channel's gradient -- differs between ANY two fp32/fp64 evaluations
    # (the derivative is discontinuous there): north_star's 1e-4 plus a direction check.
    e = rel_mse(fh.grad, fo.grad)
    cs = float((fh.grad.cpu().double().flatten() @ fo.grad.flatten()) / (fh.grad.cpu().double().norm() * fo.grad.norm()))
    assert e < 1e-4 and cs > 0.9999, (e, cs)
    # frames beyond an utterance's length get no gradient from the pooling, only through the
    # receptive fields of the convolutions near the boundary; the classifier stays frozen
    assert all(p.grad is None for p in enc.parameters())


def _losses_and_backward(recon, logp, target, gender, w):
    """recon_w*L1 - sex_w*NLL + util_w*0 - conf_w*MSE(logp, -0.6931) with the HIP loss kernels"""
    from speech_anonymization_amd import ops
    loss_r, g_r = ops.recon_loss(recon.detach().contiguous(), target.contiguous(), "l1")
    out, dn, dc = ops.cls_losses(logp.detach().contiguous(), gender)
    torch.autograd.backward([recon, logp], [w[0] * g_r.view_as(recon), -w[1] * dn - w[3] * dc])
    torch.cuda.synchronize()
    return w[0] * float(loss_r) - w[1] * float(out[0]) - w[3] * float(out[1])


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_convreconstruction_against_reference_golden_vectors(golden_dir, precision):
    from oracle import endtoend as OE
    from speech_anonymization_amd.endtoend import ConvReconstruction
    z = np.load(os.path.join(golden_dir, "endtoend_S.npz"))
    oclf = oracle_classifier()
    enc_params = {k: v for k, v in OE.numpy_params(OE.ConvReconstruction(oclf), 8886).items()
                  if k.startswith("encoder.")}
    m = ConvReconstruction(hip_classifier(oclf), precision=precision)
    m.load_state_dict(enc_params, strict=False)
    m.to(DEV).train()
    feats, target = torch.from_numpy(z["feats"]).to(DEV), torch.from_numpy(z["target"]).to(DEV)
    gender = torch.from_numpy(z["gender"]).to(DEV)
    recon, logp = m(feats)
    loss = _losses_and_backward(recon, logp, target, gender, [float(v) for v in z["weights"]])
    assert rel_mse(recon, torch.from_numpy(z["recon"])) < 1e-9
    assert rel_mse(logp, torch.from_numpy(z["logp"])) < 1e-8
    assert abs(loss - float(z["loss"])) < 3e-5
    tol = 2e-5 if precision == "f32" else 1e-4
    for k, p in m.named_parameters():
        if not k.startswith("encoder."):
            assert p.grad is None, k                        # frozen classifier
            continue
        g = p.grad.reshape(-1)
        if g.numel() <= 2 or k in ("encoder.0.bias", "encoder.3.bias", "encoder.6.bias", "encoder.9.bias"):
            continue                                        # conv biases in front of InstanceNorm: null gradients
        step = max(1, g.numel() // 2048)
        e = rel_mse(g[::step][:2048], torch.from_numpy(z["grad_sub/" + k]))
        assert e < tol, (k, e)


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_convreconstruction_against_oracle_full_tensors(precision):
    from oracle import endtoend as OE, losses as OL
    from oracle.features import synthetic_feats
    from speech_anonymization_amd.endtoend import ConvReconstruction
    B, T = 4, 72
    feats = synthetic_feats(B, T, seed=21)
    target = feats + 0.1 * torch.randn(B, T, 80, generator=torch.Generator().manual_seed(2))
    gender = torch.arange(B) % 2
    w = [0.3, 0.6, 0.0, 0.2]
    oclf = oracle_classifier()
    om = OE.ConvReconstruction(oclf)
    enc_params = {k: v for k, v in OE.numpy_params(om, 99).items() if k.startswith("encoder.")}
    om.load_state_dict(enc_params, strict=False)
    om.train(); oclf.eval()
    fo = feats.clone().requires_grad_(True)
    o_recon, o_logp = om(fo)
    o_loss = (w[0] * OL.recon_loss(o_recon, target, "l1") - w[1] * OL.sex_loss(o_logp, gender)
              - w[3] * OL.confusion_loss(o_logp))
    o_loss.backward()
    m = ConvReconstruction(hip_classifier(oclf), precision=precision)
    m.load_state_dict(enc_params, strict=False)
    m.to(DEV).train()
    fh = feats.to(DEV).requires_grad_(True)
    recon, logp = m(fh)
    loss = _losses_and_backward(recon, logp, target.to(DEV), gender.to(DEV), w)
    assert rel_mse(recon, o_recon) < 1e-9 and rel_mse(logp, o_logp) < 1e-8
    assert abs(loss - float(o_loss)) < 3e-5 * max(1.0, abs(float(o_loss)))
    tol = 2e-5 if precision == "f32" else 1e-4
    og = {k: p.grad for k, p in om.named_parameters()}
    for k, p in m.named_parameters():
        if not k.startswith("encoder."):
            continue
        if k in ("encoder.0.bias", "encoder.3.bias", "encoder.6.bias", "encoder.9.bias"):
            scale = float(og[k.replace(".bias", ".weight")].abs().max())
            assert float(p.grad.abs().max()) < 1e-3 * scale, k
            continue
        assert rel_mse(p.grad, og[k]) < tol, (k, rel_mse(p.grad, og[k]))
    assert rel_mse(fh.grad, fo.grad) < tol


def test_endtoend_model_through_the_brain_hooks():
    """model_type endtoend with ConvReconstruction through SexAnonymizationTraining (Fbank x2,
    normalise, forward, adversarial-sign loss, backward, clip, Adam, Noam): two steps run, the
    frozen classifier's parameters do not move, the loss is the signed combination."""
    from tests import smoke_step
    from speech_anonymization_amd.brain import Batch
    from speech_anonymization_amd.endtoend import ConvReconstruction
    dev = torch.device(DEV)
    br = smoke_step.build("bf16x3", dev)
    oclf = oracle_classifier()
    model = ConvReconstruction(hip_classifier(oclf)).to(dev)
    br.modules["ConvAE"] = model
    br.optimizer = None
    br.init_optimizers()
    hp = br.hparams
    hp.model_type, hp.recon_loss_weight, hp.sex_loss_weight, hp.confusion_loss_weight = "endtoend", 0.5, 0.4, 0.1
    before = {k: v.clone() for k, v in model.sex_classifier.state_dict().items()}
    enc0 = model.encoder[3].weight.detach().clone()
    batch = Batch(smoke_step.make_wave(4, 11360), torch.tensor([1.0, 0.83, 0.61, 1.0]), torch.arange(4) % 2)
    for _ in range(2):
        br.step += 1
        loss = br.fit_batch(batch)
    torch.cuda.synchronize()
    ll = br.last_losses
    assert torch.isfinite(loss)
    assert float(loss) < 0.5 * float(ll["recon"])          # recon_w*recon MINUS the classifier terms
    for k, v in model.sex_classifier.state_dict().items():
        assert torch.equal(v, before[k]), k
    assert not torch.equal(model.encoder[3].weight.detach(), enc0)
