"""Per-kernel parity: every C-ABI entry point against the CPU oracle ops (plain PyTorch CPU
fp32 / the oracle package) on seeded inputs.  fp32 path: near bit-level; bf16 path: the
north_star tolerance (rel-MSE <= 1e-4), measured against a reference fed the same
bf16-rounded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]
DTX = DT + ["bf16x3"]            # + the benchmarked precision: fp32 storage, split-bf16 MFMA operands


def dev():
    return torch.device("cuda:0")


def rel_mse(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float(((a - b) ** 2).sum() / (b ** 2).sum().clamp_min(1e-30))


def tol(dt):
    # bf16x3: ~16 mantissa bits per operand (hi + lo) -> products good to ~2^-16 each
    return 1e-10 if dt == torch.float32 else (2e-9 if dt == "bf16x3" else 2e-5)


def sdt(dt):
    """storage dtype of a precision"""
    return torch.float32 if dt == "bf16x3" else dt


def kcode(dt):
    """(kernel code of conv_gemm, kernel code of wgrad) or (None, None) = from the storage dtype"""
    from speech_anonymization_amd import _lib as L
    return (L.BF16X3, L.BF16X1F) if dt == "bf16x3" else (None, None)


def rnd(dt, *shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g) * scale
    return x.to(sdt(dt)).float()     # values exactly representable in the storage dtype


def cl(x_bcl, dt):
    """[B,C,L] cpu fp32 -> channels-last [B,L,C] device tensor of dtype dt."""
    return x_bcl.permute(0, 2, 1).contiguous().to(dev(), sdt(dt))


def uncl(y_blc):
    return y_blc.float().cpu().permute(0, 2, 1).contiguous()


# (name, cin, cout, K, stride, dil, pad, transposed)
LAYERS = [
    ("enc2", 32, 64, 5, 2, 1, 2, False), ("enc5", 64, 64, 5, 1, 1, 2, False),
    ("enc8", 64, 128, 5, 2, 1, 2, False), ("enc11", 128, 128, 5, 1, 1, 2, False),
    ("tdnn0", 128, 128, 5, 1, 1, 0, False), ("tdnn3", 128, 128, 3, 1, 2, 0, False),
    ("tdnn6", 128, 128, 3, 1, 3, 0, False),
    ("dec1", 128, 64, 5, 2, 1, 2, True), ("dec5", 64, 32, 5, 2, 1, 2, True),
]


def ref_fwd(x, w, b, stride, dil, pad, transposed):
    if transposed:
        return F.conv_transpose1d(x, w, b, stride=stride, padding=pad, output_padding=1)
    return F.conv1d(x, w, b, stride=stride, padding=pad, dilation=dil)


def _conv_case(layer, dt, B, Lin):
    from speech_anonymization_amd import ops
    name, cin, cout, K, stride, dil, pad, tr = layer
    cg, cw = kcode(dt)
    x = rnd(dt, B, cin, Lin, seed=1)
    wshape = (cin, cout, K) if tr else (cout, cin, K)
    w = rnd(dt, *wshape, seed=2, scale=(cin * K) ** -0.5)
    bias = rnd(torch.float32, cout, seed=3, scale=0.1)
    x.requires_grad_(True)
    w.requires_grad_(True)
    y = ref_fwd(x, w, bias, stride, dil, pad, tr)
    Lout = y.shape[2]
    gy = rnd(dt, B, cout, Lout, seed=4)
    gx, gw = torch.autograd.grad(y, (x, w), gy)

    xd, wd, bd = cl(x.detach(), dt), w.detach().to(dev()), bias.to(dev())
    # ---- forward ----
    if tr:
        wp = ops.pack_weights(wd, "convT_fwd", sdt(dt), cg)
        yd, st = ops.conv_gemm(xd, wp, bd, cin, cout, 1, 2, ops.UP2, Lout, want_stats=True, code=cg)
    else:
        wp = ops.pack_weights(wd, "conv_fwd", sdt(dt), cg)
        yd, st = ops.conv_gemm(xd, wp, bd, cin, cout, stride, 1, ops.taps_conv(K, dil, pad), Lout,
                               want_stats=True, code=cg)
    torch.cuda.synchronize()
    assert rel_mse(uncl(yd), y.detach()) < tol(dt), name
    # epilogue statistics = sum / sumsq of the stored values per (b, c)
    s = ops.sum_partials(st, B).view(B, cout, 2).cpu()
    yy = yd.double().cpu()
    assert torch.allclose(s[..., 0], yy.sum(dim=1), rtol=1e-4, atol=1e-3 * (Lout / 300) ** 0.5)
    assert torch.allclose(s[..., 1], (yy * yy).sum(dim=1), rtol=1e-4, atol=1e-3 * (Lout / 300) ** 0.5)
    # ---- dgrad ----
    gyd = cl(gy, dt)
    if tr:
        wpd = ops.pack_weights(wd, "convT_dgrad", sdt(dt), cg)
        gxd = ops.conv_gemm(gyd, wpd, None, cout, cin, 2, 1, ops.taps_convT_dgrad(), Lin, code=cg)
    elif stride == 2:
        wpd = ops.pack_weights(wd, "conv_dgrad", sdt(dt), cg)
        gxd = ops.conv_gemm(gyd, wpd, None, cout, cin, 1, 2, ops.UP2, Lin, code=cg)
    else:
        wpd = ops.pack_weights(wd, "conv_dgrad", sdt(dt), cg)
        gxd = ops.conv_gemm(gyd, wpd, None, cout, cin, 1, 1, ops.taps_conv_dgrad_s1(K, dil, pad), Lin,
                            code=cg)
    torch.cuda.synchronize()
    assert rel_mse(uncl(gxd), gx) < tol(dt), name + " dgrad"
    # ---- wgrad (bf16x3 models run it with single-rounded bf16 operands, SA_BF16X1F: a weight
    # gradient averages ~B*L independent roundings of relative size 2^-9) ----
    gwd = torch.zeros(wshape, device=dev())
    if tr:
        taps = [(1, 0), (1, 1), (0, 0), (0, 1), (-1, 0)]
        ops.wgrad(xd, gyd, cin, cout, 1, 2, taps, Lin, gwd, (cout * K, K, 1), code=cw)
    else:
        taps = [(k * dil - pad, 0) for k in range(K)]
        ops.wgrad(xd, gyd, cin, cout, stride, 1, taps, Lout, gwd, (K, cin * K, 1), code=cw)
    torch.cuda.synchronize()
    assert rel_mse(gwd, gw) < (2e-5 if dt == "bf16x3" else tol(dt)), name + " wgrad"


@pytest.mark.parametrize("dt", DTX, ids=["f32", "bf16", "bf16x3"])
@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_conv_fwd_dgrad_wgrad(layer, dt):
    _conv_case(layer, dt, 2, 300)


@pytest.mark.parametrize("dt", ["bf16x3", torch.float32], ids=["bf16x3", "f32"])
@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_conv_fwd_dgrad_wgrad_at_training_length(layer, dt):
    """every template at the row counts of the benchmark shape (T = 1008: 80 640 / 40 320 /
    20 160 rows per utterance -> 158-630 tiles, multi-chunk split-K weight gradients, two-level
    slab sums) against F.conv1d / F.conv_transpose1d on the CPU."""
    name, cin, cout, K, stride, dil, pad, tr = layer
    Lin = {32: 80640, 64: 40320, 128: 20160}[cin]
    _conv_case(layer, dt, 2, Lin)


@pytest.mark.parametrize("dt", DTX, ids=["f32", "bf16", "bf16x3"])
def test_conv_prologue_relu(dt):
    """IN affine + swish + BN affine in the prologue, zero padding applied AFTER the transform,
    ReLU epilogue; ragged length (not a multiple of the 128-row tile)."""
    from speech_anonymization_amd import ops
    B, Cc, Lin, K = 3, 128, 517, 5
    cg, _ = kcode(dt)
    x = rnd(dt, B, Cc, Lin, seed=5)
    w = rnd(dt, Cc, Cc, K, seed=6, scale=(Cc * K) ** -0.5)
    s1, t1 = 1 + 0.2 * rnd(torch.float32, B, Cc, seed=7), 0.3 * rnd(torch.float32, B, Cc, seed=8)
    s2, t2 = 1 + 0.2 * rnd(torch.float32, Cc, seed=9), 0.3 * rnd(torch.float32, Cc, seed=10)
    z = x * s1[:, :, None] + t1[:, :, None]
    a = (z * torch.sigmoid(z)) * s2[None, :, None] + t2[None, :, None]
    a = a.to(sdt(dt)).float()                              # the kernel stages the operand in dt
    y = F.relu(F.conv1d(a, w, None, padding=2))
    wp = ops.pack_weights(w.to(dev()), "conv_fwd", sdt(dt), cg)
    yd = ops.conv_gemm(cl(x, dt), wp, None, Cc, Cc, 1, 1, ops.taps_conv(K, 1, 2), Lin,
                       s1=s1.to(dev()), t1=t1.to(dev()), s2=s2.to(dev()), t2=t2.to(dev()),
                       swish=True, relu=True, code=cg)
    torch.cuda.synchronize()
    assert rel_mse(uncl(yd), y) < {torch.float32: 1e-9, "bf16x3": 1e-8}.get(dt, 1e-4)


@pytest.mark.parametrize("per_c,relu_mask", [(False, False), (True, True)], ids=["instnorm", "batchnorm_relu"])
@pytest.mark.parametrize("layer", [LAYERS[1], LAYERS[3], LAYERS[6], LAYERS[7]],
                         ids=["enc5", "enc11", "tdnn6", "dec1"])
def test_conv_norm_backward_prologue(layer, per_c, relu_mask):
    """The normalisation-backward prologue of the data-gradient launches (SaConvArgs.nb_*, the
    PRO2 template, bf16x3): d y = c1*dz + c2*y + c3 [* (y > 0)] formed while the rows are staged,
    then the data-gradient convolution over d y; by-products: bf16(d y) (the weight gradient's
    operand) and the column sums of d y (the bias gradient).  Reference: the same arithmetic in
    torch on the CPU (what sa_ew_apply + a plain data gradient compute)."""
    from speech_anonymization_amd import _lib as L, ops
    name, cin, cout, K, stride, dil, pad, tr = layer
    B, Lx = 3, 517                      # Lx = rows of the layer's INPUT x (the dgrad's output)
    w = rnd(torch.float32, *((cin, cout, K) if tr else (cout, cin, K)), seed=2, scale=(cin * K) ** -0.5)
    x = rnd(torch.float32, B, cin, Lx, seed=1).requires_grad_(True)
    yref = ref_fwd(x, w, None, stride, dil, pad, tr)
    Ly = yref.shape[2]
    dz = rnd(torch.float32, B, cout, Ly, seed=3)
    ystored = rnd(torch.float32, B, cout, Ly, seed=4)          # forward tensor of the layer above
    shp = (cout,) if per_c else (B, cout)
    c1, c2, c3 = (1 + 0.2 * rnd(torch.float32, *shp, seed=5), 0.1 * rnd(torch.float32, *shp, seed=6),
                  0.05 * rnd(torch.float32, *shp, seed=7))
    bc = (lambda c: c[None, :, None]) if per_c else (lambda c: c[:, :, None])
    dy = bc(c1) * dz + bc(c2) * ystored + bc(c3)
    if relu_mask:
        dy = dy * (ystored > 0)
    (gx,) = torch.autograd.grad(yref, x, dy)
    d = dev()
    if tr:
        wp = ops.pack_weights(w.to(d), "convT_dgrad", torch.float32, L.BF16X3)
        args = (cout, cin, 2, 1, ops.taps_convT_dgrad(), Lx)
    else:
        wp = ops.pack_weights(w.to(d), "conv_dgrad", torch.float32, L.BF16X3)
        args = (cout, cin, 1, 1, ops.taps_conv_dgrad_s1(K, dil, pad), Lx)
    dzd, yd = cl(dz, torch.float32), cl(ystored, torch.float32)
    dyc = torch.full(dzd.shape, float("nan"), dtype=torch.bfloat16, device=d)
    out = ops.conv_gemm(dzd, wp, None, *args, code=L.BF16X3, a_out=dyc,
                        nb=dict(x=yd, c1=c1.to(d).contiguous(), c2=c2.to(d).contiguous(),
                                c3=c3.to(d).contiguous(), per_c=per_c, relu_mask=relu_mask,
                                want_colsum=True))
    gxd, cs = out
    torch.cuda.synchronize()
    assert rel_mse(uncl(gxd), gx) < 2e-9
    dy_cl = dy.permute(0, 2, 1)
    assert not torch.isnan(dyc.float()).any()
    assert rel_mse(dyc.float().cpu(), dy_cl.bfloat16().float()) < 1e-6       # one bf16 rounding (ties aside)
    assert torch.allclose(cs.sum(1).cpu().double(), dy_cl.double().sum(1), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("dt", DT)
def test_small_channel_kernels(dt):
    from speech_anonymization_amd import ops
    B, Ln = 2, 1300
    x = rnd(torch.float32, B, Ln, seed=11)
    w0 = rnd(torch.float32, 32, 1, 15, seed=12, scale=0.25)
    b0 = rnd(torch.float32, 32, seed=13, scale=0.1)
    # encoder.0 forward
    y0 = F.conv1d(x[:, None], w0, b0, padding=7)
    y0d, st = ops.conv1toC(x.to(dev()), w0.to(dev()), b0.to(dev()), dt, want_stats=True)
    torch.cuda.synchronize()
    assert rel_mse(uncl(y0d), y0) < tol(dt)
    assert torch.allclose(st.sum(1)[..., 0].cpu(), y0d.float().cpu().sum(1), rtol=1e-4, atol=1e-3)
    # encoder.0 wgrad: dW[c][k] = sum x[l+k-7] * dy[l][c]
    gy = rnd(dt, B, 32, Ln, seed=14)
    w0r = w0.clone().requires_grad_(True)
    (gw,) = torch.autograd.grad(F.conv1d(x[:, None], w0r, None, padding=7), w0r, gy)
    gwd = torch.zeros(32, 1, 15, device=dev())
    ops.wgrad1C(x.to(dev()), cl(gy, dt), gwd)
    torch.cuda.synchronize()
    assert rel_mse(gwd, gw) < tol(dt)
    # decoder.8 forward with IN+swish prologue
    v = rnd(dt, B, 32, Ln, seed=15)
    s1, t1 = 1 + 0.2 * rnd(torch.float32, B, 32, seed=16), 0.3 * rnd(torch.float32, B, 32, seed=17)
    w8 = rnd(torch.float32, 1, 32, 15, seed=18, scale=0.05)
    b8 = rnd(torch.float32, 1, seed=19, scale=0.1)
    z = v * s1[:, :, None] + t1[:, :, None]
    a = (z * torch.sigmoid(z)).requires_grad_(True)
    w8r = w8.clone().requires_grad_(True)
    y8 = F.conv1d(a, w8r, b8, padding=7)
    y8d = ops.convCto1(cl(v, dt), w8.to(dev()), b8.to(dev()), s1.to(dev()), t1.to(dev()), True)
    torch.cuda.synchronize()
    assert rel_mse(y8d.cpu(), y8.detach()[:, 0]) < 1e-9
    # decoder.8 dgrad (conv1toC, flipped taps) and wgrad (wgrad1C, flipped, with prologue)
    g8 = rnd(torch.float32, B, 1, Ln, seed=20)
    ga, gw8 = torch.autograd.grad(y8, (a, w8r), g8)
    gad = ops.conv1toC(g8[:, 0].contiguous().to(dev()), w8.to(dev()), None, dt, flip=True)
    gw8d = torch.zeros(1, 32, 15, device=dev())
    ops.wgrad1C(g8[:, 0].contiguous().to(dev()), cl(v, dt), gw8d, flip=True, s1=s1.to(dev()),
                t1=t1.to(dev()), swish=True)
    torch.cuda.synchronize()
    assert rel_mse(uncl(gad), ga) < tol(dt)
    assert rel_mse(gw8d, gw8) < 1e-9
    # the same dgrad with the fused [InstanceNorm -> swish] backward epilogue == dgrad + sa_ew_stats
    mu, rs = 0.1 * rnd(torch.float32, B, 32, seed=21).to(dev()), (1 + 0.1 * rnd(torch.float32, B, 32, seed=22)).abs().to(dev())
    vd = cl(v, dt)
    gz, st = ops.conv1toC(g8[:, 0].contiguous().to(dev()), w8.to(dev()), None, dt, flip=True, want_stats=True,
                          ep=dict(x=vd, s1=s1.to(dev()), t1=t1.to(dev()), mean=mu, rstd=rs))
    gz_ref = torch.empty_like(gad)
    st_ref = ops.ew("stats", gad, vd, 32, out=gz_ref, s1=s1.to(dev()), t1=t1.to(dev()), mean=mu, rstd=rs,
                    actbwd=True)
    torch.cuda.synchronize()
    assert rel_mse(gz.float(), gz_ref.float()) < tol(dt)
    assert rel_mse(ops.sum_partials(st, B).cpu(), ops.sum_partials(st_ref, B).cpu()) < (1e-10 if dt == torch.float32 else 1e-4)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("Cc", [32, 64, 128])
def test_instance_norm_swish_backward(dt, Cc):
    """two-phase IN + x*sigmoid(x) backward (sa_ew_stats -> sa_fin_norm_bwd -> sa_ew_apply)
    against autograd through nn.InstanceNorm1d(affine) + the reference's GLU."""
    from speech_anonymization_amd import ops
    B, Ln = 2, 700
    y = rnd(dt, B, Cc, Ln, seed=21).requires_grad_(True)
    gamma = (1 + 0.1 * rnd(torch.float32, Cc, seed=22)).requires_grad_(True)
    beta = (0.1 * rnd(torch.float32, Cc, seed=23)).requires_grad_(True)
    z = F.instance_norm(y, weight=gamma, bias=beta, eps=1e-5)
    a = z * torch.sigmoid(z)
    ga = rnd(dt, B, Cc, Ln, seed=24)
    gy, gg, gb = torch.autograd.grad(a, (y, gamma, beta), ga)

    yd = cl(y.detach(), dt)
    # forward statistics through the same finaliser the conv epilogue feeds
    yy = yd.float()
    part = torch.stack([yy.sum(1), (yy * yy).sum(1)], dim=-1).reshape(B, 1, Cc, 2).contiguous()
    sums = ops.sum_partials(part, B)
    mean, rstd, scale, shift = ops.fin_in_fwd(sums, B, Cc, Ln, gamma.detach().to(dev()),
                                              beta.detach().to(dev()))
    gad = cl(ga, dt)
    dz = torch.empty_like(gad)
    st = ops.ew("stats", gad, yd, Cc, out=dz, s1=scale, t1=shift, mean=mean, rstd=rstd, actbwd=True)
    dgam, dbet = torch.zeros(Cc, device=dev()), torch.zeros(Cc, device=dev())
    bs = ops.sum_partials(st, B)
    c1, c2, c3 = ops.fin_norm_bwd(bs, bs, B * Cc, Cc, Ln, gamma.detach().to(dev()), mean, rstd,
                                  dgamma=dgam, dbeta=dbet)
    dy = torch.empty_like(gad)
    st2 = ops.ew("apply", dz, yd, Cc, out=dy, c1=c1, c2=c2, c3=c3)
    torch.cuda.synchronize()
    t = 1e-9 if dt == torch.float32 else 1e-4
    assert rel_mse(uncl(dy), gy) < t
    assert rel_mse(dgam, gg) < t and rel_mse(dbet, gb) < t
    db = torch.zeros(Cc, device=dev())
    ops.fin_bias(ops.sum_partials(st2, B), B, Cc, db)
    torch.cuda.synchronize()
    assert torch.allclose(db.cpu(), dy.float().cpu().sum((0, 1)), atol=1e-2)


@pytest.mark.parametrize("dt", DT)
def test_relu_batchnorm_backward(dt):
    """TDNN block: stored r = relu(conv); BatchNorm1d (train) statistics over (B, L)."""
    from speech_anonymization_amd import ops
    B, Cc, Ln = 3, 128, 411
    y = rnd(dt, B, Cc, Ln, seed=31).requires_grad_(True)
    gamma = (1 + 0.1 * rnd(torch.float32, Cc, seed=32)).requires_grad_(True)
    beta = (0.1 * rnd(torch.float32, Cc, seed=33)).requires_grad_(True)
    r = F.relu(y)
    rm, rv = torch.zeros(Cc), torch.ones(Cc)
    o = F.batch_norm(r, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    go = rnd(dt, B, Cc, Ln, seed=34)
    gy, gg, gb = torch.autograd.grad(o, (y, gamma, beta), go)

    rd = cl(r.detach(), dt)
    rr = rd.float()
    part = torch.stack([rr.sum(1), (rr * rr).sum(1)], dim=-1).reshape(B, 1, Cc, 2).contiguous()
    sums = ops.sum_partials(part, 1)
    rmd, rvd = torch.zeros(Cc, device=dev()), torch.ones(Cc, device=dev())
    mean, rstd, scale, shift = ops.fin_bn_fwd(sums, Cc, B * Ln, gamma.detach().to(dev()),
                                              beta.detach().to(dev()), rmd, rvd)
    torch.cuda.synchronize()
    assert torch.allclose(rmd.cpu(), rm, atol=1e-5) and torch.allclose(rvd.cpu(), rv, atol=1e-5)
    ref_o = o.detach()
    assert rel_mse((rr * scale + shift).cpu().permute(0, 2, 1), ref_o) < 1e-9
    god = cl(go, dt)
    st = ops.ew("stats", god, rd, Cc, mean=mean, rstd=rstd, per_c=True)
    bs = ops.sum_partials(st, 1)
    dgam, dbet = torch.zeros(Cc, device=dev()), torch.zeros(Cc, device=dev())
    c1, c2, c3 = ops.fin_norm_bwd(bs, bs, Cc, Cc, B * Ln, gamma.detach().to(dev()), mean, rstd,
                                  dgamma=dgam, dbeta=dbet)
    dy = torch.empty_like(god)
    ops.ew("apply", god, rd, Cc, out=dy, c1=c1, c2=c2, c3=c3, relu_mask=True, per_c=True,
           want_stats=False)
    torch.cuda.synchronize()
    t = 1e-9 if dt == torch.float32 else 1e-4
    assert rel_mse(uncl(dy), gy) < t
    assert rel_mse(dgam, gg) < t and rel_mse(dbet, gb) < t


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("Ln", [20146, 333])
def test_statistics_pooling_reshape_quirk(dt, Ln):
    """reshape (not transpose) of [B,128,L] to [B,L,128] before StatisticsPooling
    (models/ConvAutoEncoder.py:61-66) -- forward and backward."""
    from oracle.convae import StatisticsPooling
    from speech_anonymization_amd import ops
    B, Cc = 2, 128
    r = rnd(dt, B, Cc, Ln, seed=41).abs()
    sc, sh = 1 + 0.1 * rnd(torch.float32, Cc, seed=42), 0.1 * rnd(torch.float32, Cc, seed=43)
    xbn = (r * sc[None, :, None] + sh[None, :, None]).requires_grad_(True)
    pooled = StatisticsPooling()(xbn.reshape(B, Ln, Cc)).squeeze(1)
    gp = rnd(torch.float32, B, 256, seed=44)
    (gx,) = torch.autograd.grad(pooled, xbn, gp)
    rd = cl(r, dt)
    pd, mean, sd = ops.pool_fwd(rd, sc.to(dev()), sh.to(dev()))
    g = ops.pool_bwd(rd, sc.to(dev()), sh.to(dev()), gp.to(dev()), mean, sd)
    torch.cuda.synchronize()
    assert rel_mse(pd, pooled.detach()) < 1e-9
    assert rel_mse(uncl(g), gx) < (1e-9 if dt == torch.float32 else 2e-5)
    # fused BatchNorm-backward statistics of the written gradient == a separate sa_ew_stats pass
    bm, br = 0.1 * rnd(torch.float32, Cc, seed=45).to(dev()), (1 + 0.1 * rnd(torch.float32, Cc, seed=46)).abs().to(dev())
    g2, st = ops.pool_bwd(rd, sc.to(dev()), sh.to(dev()), gp.to(dev()), mean, sd, bn=(bm, br))
    st_ref = ops.ew("stats", g, rd, Cc, mean=bm, rstd=br, per_c=True)
    assert torch.equal(g2, g)
    a, b_ = ops.sum_partials(st, 1).cpu(), ops.sum_partials(st_ref, 1).cpu()
    assert rel_mse(a, b_) < (1e-10 if dt == torch.float32 else 1e-6)
    # noise term of speechbrain's pooling: mean += eps*((1-9)*g+9)
    noise = torch.rand(B, 128)
    pn, _, _ = ops.pool_fwd(rd, sc.to(dev()), sh.to(dev()), noise=noise.to(dev()))
    ref = StatisticsPooling(noise=noise)(xbn.detach().reshape(B, Ln, Cc)).squeeze(1)
    torch.cuda.synchronize()
    assert rel_mse(pn, ref) < 1e-9


def test_fc_head_and_losses():
    from speech_anonymization_amd import ops
    B = 10
    torch.manual_seed(0)
    lin1, bn1 = torch.nn.Linear(256, 128), torch.nn.BatchNorm1d(128)
    lin2, bn2 = torch.nn.Linear(128, 64), torch.nn.BatchNorm1d(64)
    lin3 = torch.nn.Linear(64, 2)
    p = torch.randn(B, 256, requires_grad=True)
    h1 = F.relu(lin1(p)); n1 = bn1(h1)
    h2 = F.relu(lin2(n1)); n2 = bn2(h2)
    logits = lin3(n2)
    logp = F.log_softmax(logits, 1)
    gender = torch.arange(B) % 2
    nll = F.nll_loss(logp, gender)
    conf = F.mse_loss(logp, torch.ones_like(logp) * -0.6931)
    loss = 0.9 * nll + 0.3 * conf
    params = [lin1.weight, lin1.bias, bn1.weight, bn1.bias, lin2.weight, lin2.bias, bn2.weight,
              bn2.bias, lin3.weight, lin3.bias, p]
    grads = torch.autograd.grad(loss, params)

    d = dev()
    g = lambda t: t.detach().to(d).contiguous()
    H1 = ops.dense(g(p), g(lin1.weight), g(lin1.bias), 128, 256, relu=True)
    m1, r1, s1, t1 = ops.fin_bn_fwd(ops.colsums(H1), 128, B, g(bn1.weight), g(bn1.bias))
    H2 = ops.dense(H1, g(lin2.weight), g(lin2.bias), 64, 128, ps=s1, pt=t1, relu=True)
    m2, r2, s2, t2 = ops.fin_bn_fwd(ops.colsums(H2), 64, B, g(bn2.weight), g(bn2.bias))
    LG = ops.dense(H2, g(lin3.weight), g(lin3.bias), 2, 64, ps=s2, pt=t2)
    LP = ops.log_softmax(LG)
    out, dn, dc = ops.cls_losses(LP, gender.to(d))
    torch.cuda.synchronize()
    assert rel_mse(LP, logp.detach()) < 1e-9
    assert abs(float(out[0]) - float(nll)) < 1e-5 and abs(float(out[1]) - float(conf)) < 1e-5
    dLP = 0.9 * dn + 0.3 * dc
    dLG = ops.log_softmax_bwd(dLP, LP)
    dW3 = ops.dense_wgrad(dLG, H2, torch.empty(2, 64, device=d), ps=s2, pt=t2)
    dN2 = ops.dense(dLG, g(lin3.weight), None, 64, 2, transpose_w=True)
    S2 = ops.colsums(dN2, H2, m2, r2)
    dH2 = ops.bn2d_bwd(dN2, H2, S2, B, g(bn2.weight), m2, r2, True)
    dW2 = ops.dense_wgrad(dH2, H1, torch.empty(64, 128, device=d), ps=s1, pt=t1)
    dN1 = ops.dense(dH2, g(lin2.weight), None, 128, 64, transpose_w=True)
    S1 = ops.colsums(dN1, H1, m1, r1)
    dH1 = ops.bn2d_bwd(dN1, H1, S1, B, g(bn1.weight), m1, r1, True)
    dW1 = ops.dense_wgrad(dH1, g(p), torch.empty(128, 256, device=d))
    dP = ops.dense(dH1, g(lin1.weight), None, 256, 128, transpose_w=True)
    torch.cuda.synchronize()
    got = [dW1, dH1.sum(0), S1[:, 1], S1[:, 0], dW2, dH2.sum(0), S2[:, 1], S2[:, 0], dW3,
           dLG.sum(0), dP]
    for a, b in zip(got, grads):
        assert rel_mse(a, b) < 1e-8


@pytest.mark.parametrize("kind", ["l1", "mse"])
def test_recon_loss(kind):
    from oracle import losses
    from speech_anonymization_amd import ops
    a = rnd(torch.float32, 3, 72, 80, seed=51).requires_grad_(True)
    b = rnd(torch.float32, 3, 72, 80, seed=52)
    ref = losses.recon_loss(a, b, kind)
    (ga,) = torch.autograd.grad(ref, a)
    loss, grad = ops.recon_loss(a.detach().to(dev()), b.to(dev()), kind)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref)) < 1e-6 * max(1.0, abs(float(ref)))
    assert rel_mse(grad, ga) < 1e-10


def test_cosine_and_mi_golden(golden_dir):
    """the reference's own CosineSimilarityLoss / ClusterMI / GroupSamplingMI outputs
    (tests/golden/losses.npz, generated from /root/reference by oracle/gen_golden.py)."""
    import os
    from speech_anonymization_amd import ops
    z = np.load(os.path.join(golden_dir, "losses.npz"))
    x1, x2 = torch.from_numpy(z["x1"]), torch.from_numpy(z["x2"])
    loss, dx1 = ops.cosine_loss(x1.to(dev()), x2.to(dev()), want_grad=True)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(z["cos_loss"])) < 1e-5
    x1r = x1.clone().requires_grad_(True)
    from oracle.losses import cosine_similarity_loss
    (g,) = torch.autograd.grad(cosine_similarity_loss(x1r, x2), x1r)
    assert rel_mse(dx1, g) < 1e-8
    X, y = torch.from_numpy(z["X"]).to(dev()), torch.from_numpy(z["y"]).to(dev())
    mi = ops.cluster_mi(X, y)
    idx = torch.from_numpy(z["idx_sets"]).long().to(dev())
    mis = ops.cluster_mi(X, y, idx)
    torch.cuda.synchronize()
    assert abs(float(mi[0]) - float(z["mi"])) < 1e-4
    assert np.allclose(mis.cpu().numpy(), z["mi_list"], atol=1e-4)
    # second reference-generated case: N = 32 (the bench's batch size) with duplicated rows, i.e.
    # exact ties in `d <= anchor` (utils/ClusterMI.py:115) -- neighbour counting is index work
    z = np.load(os.path.join(golden_dir, "losses_n32.npz"))
    X, y = torch.from_numpy(z["X"]).to(dev()), torch.from_numpy(z["y"]).to(dev())
    mi = ops.cluster_mi(X, y)
    mis = ops.cluster_mi(X, y, torch.from_numpy(z["idx_sets"]).long().to(dev()))
    torch.cuda.synchronize()
    assert abs(float(mi[0]) - float(z["mi"])) < 1e-4
    assert np.allclose(mis.cpu().numpy(), z["mi_list"], atol=1e-4)


@pytest.mark.parametrize("mode", ["utterance", "batch"])
def test_fbank_and_normalization(mode):
    from oracle import features as OF
    from speech_anonymization_amd import Fbank, InputNormalization
    B, N = 3, 11360
    wav = OF.synthetic_wave(B, N, seed=8886)
    wav[2, 7000:] = 0.0                                          # zero-padded tail
    lens = torch.tensor([1.0, 0.83, 0.61])
    ofb, onorm = OF.Fbank(top_db_mode=mode), OF.InputNormalization(update_until_epoch=4)
    fb, norm = Fbank(top_db_mode=mode).to(dev()), InputNormalization(update_until_epoch=4).to(dev())
    f_ref = ofb(wav)
    f = fb(wav.to(dev()))
    torch.cuda.synchronize()
    assert f.raw.shape == (B, 72, 80)
    assert rel_mse(f.clamped(), f_ref) < 1e-8
    for epoch in (1, 1, 5):                                       # first call, running update, frozen
        r_ref = OF.pad_to_multiple(onorm(f_ref, lens, epoch=epoch), 36)
        r = norm(f, lens, epoch=epoch, pad_multiple=36)
        torch.cuda.synchronize()
        assert r.shape == r_ref.shape
        assert rel_mse(r, r_ref) < 1e-8
    sd = norm.state_dict()
    assert sd["count"] == onorm.count
    assert torch.allclose(sd["glob_mean"].cpu(), onorm.glob_mean, atol=1e-3)
    assert torch.allclose(sd["glob_std"].cpu(), onorm.glob_std, atol=1e-3)


def test_xvector_classifier_forward():
    """x-vector gender classifier forward (eval) vs the oracle restatement, with the
    reference checkpoint's parameter naming."""
    import json, os
    from oracle import xvector as OX
    from speech_anonymization_amd import xvector as HX
    torch.manual_seed(3)
    ox, oc = OX.Xvector().eval(), OX.Classifier().eval()
    for m in list(ox.modules()) + list(oc.modules()):
        if isinstance(m, torch.nn.BatchNorm1d):                 # non-trivial running statistics
            m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.8, 1.2); m.bias.data.normal_(0, 0.1)
    hx, hc = HX.Xvector(pooling_noise=None), HX.Classifier(input_shape=[None, None, 128])
    assert list(hx.state_dict().keys()) == list(ox.state_dict().keys())
    assert list(hc.state_dict().keys()) == list(oc.state_dict().keys())
    pins = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_pins.json")))
    assert {k: list(v.shape) for k, v in hc.state_dict().items()} == pins["classifier_ckpt"]["shapes"]
    hx.load_state_dict(ox.state_dict()); hc.load_state_dict(oc.state_dict())
    hx.to(dev()); hc.to(dev())
    B, T = 3, 90
    feats = rnd(torch.float32, B, T, 80, seed=61)
    lens = torch.tensor([1.0, 0.77, 0.5])
    with torch.no_grad():
        e_ref = ox(feats, lens)
        p_ref = oc(e_ref)
    e = hx(feats.to(dev()), lens)
    p = hc(e)
    torch.cuda.synchronize()
    assert e.shape == (B, 1, 128) and p.shape == (B, 1, 2)
    assert rel_mse(e, e_ref) < 1e-9 and rel_mse(p, p_ref) < 1e-9
    enc = HX.EncoderClassifier(hx, hc)
    out, score, index = enc.classify_batch_feats(feats.to(dev()), lens)
    assert out.shape == (B, 2) and index.shape == (B,)


def test_pack_weights_multi_matches_single():
    """one-launch refresh of several images == sa_pack_weights image by image (all codes/kinds)."""
    from speech_anonymization_amd import _lib as L, ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    items = []
    for i, (shape, kind, code, dt) in enumerate([
            ((128, 64, 5), "conv_fwd", L.BF16X3, torch.float32),
            ((128, 128, 3), "conv_dgrad", L.BF16X3, torch.float32),
            ((128, 64, 5), "convT_fwd", L.BF16X1F, torch.float32),
            ((64, 32, 5), "convT_dgrad", L.F32, torch.float32),
            ((64, 64, 5), "conv_fwd", L.BF16, torch.float32)]):
        w = torch.randn(*shape, generator=g).to(dev)
        items.append((i, w, kind, code))
    pk = ops.PackedWeights(items, torch.float32)
    for img, _ in pk.images.values():
        img.zero_()
    pk.refresh()
    for tag, w, kind, code in items:
        img, c = pk.images[tag]
        one = ops.pack_weights(w, kind, torch.bfloat16 if code == L.BF16 else torch.float32, code)
        assert c == code and img.numel() == one.numel()
        assert torch.equal(img.view(torch.uint8), one.view(torch.uint8))


@pytest.mark.parametrize("cfg", [(128, 128, 1, 1, 5, 1, 2), (64, 128, 2, 1, 5, 1, 2), (128, 128, 1, 1, 3, 2, 0),
                                 (128, 64, 1, 2, 5, 1, 2), (32, 64, 2, 1, 5, 1, 2)])
def test_wgrad_from_cached_operand_is_bit_identical(cfg):
    """sa_conv_gemm's a_out (bf16 transformed input) fed to sa_wgrad(x_pre) == sa_wgrad recomputing
    the transform from the fp32 rows: same products, same order -> same bits."""
    from speech_anonymization_amd import _lib as L, ops
    cin, cout, sa, u, K, dil, pad = cfg
    d = dev()
    g = torch.Generator(device="cpu").manual_seed(11)
    B = 3
    if u == 2:
        Lin, Lout = 203, 406
        w = torch.randn(cin, cout, 5, generator=g).to(d) * 0.1
        wp, phases = ops.pack_weights(w, "convT_fwd", torch.float32, L.BF16X3), ops.UP2
        taps, Mrows, dst, strides = [(1, 0), (1, 1), (0, 0), (0, 1), (-1, 0)], Lin, torch.empty(cin, cout, 5, device=d), (cout * 5, 5, 1)
    else:
        Lin = 406 if sa == 1 else 407
        Lout = (Lin + 2 * pad - dil * (K - 1) - 1) // sa + 1
        w = torch.randn(cout, cin, K, generator=g).to(d) * 0.1
        wp, phases = ops.pack_weights(w, "conv_fwd", torch.float32, L.BF16X3), ops.taps_conv(K, dil, pad)
        taps, Mrows, dst, strides = [(k * dil - pad, 0) for k in range(K)], Lout, torch.empty(cout, cin, K, device=d), (K, cin * K, 1)
    x = torch.randn(B, Lin, cin, generator=g).to(d)
    s1 = (torch.rand(B, cin, generator=g) + 0.5).to(d)
    t1 = (torch.randn(B, cin, generator=g) * 0.1).to(d)
    s2 = (torch.rand(cin, generator=g) + 0.5).to(d)
    t2 = (torch.randn(cin, generator=g) * 0.1).to(d)
    a_out = torch.full((B, Lin, cin), float("nan"), dtype=torch.bfloat16, device=d)
    ops.conv_gemm(x, wp, None, cin, cout, sa, u, phases, Lout, s1=s1, t1=t1, swish=True, s2=s2, t2=t2,
                  code=L.BF16X3, a_out=a_out)
    assert not torch.isnan(a_out.float()).any()          # every input row was written by its owner tile
    ref = torch.nn.functional.silu(x * s1[:, None, :] + t1[:, None, :]) * s2 + t2
    assert rel_mse(a_out.float(), ref) < 1e-5
    dy = torch.randn(B, Lout, cout, generator=g).to(d)
    d1, d2 = torch.empty_like(dst), torch.empty_like(dst)
    ops.wgrad(x, dy, cin, cout, sa, u, taps, Mrows, d1, strides, s1=s1, t1=t1, swish=True, s2=s2, t2=t2,
              code=L.BF16X1F)
    ops.wgrad(a_out, dy, cin, cout, sa, u, taps, Mrows, d2, strides, code=L.BF16X1F, x_pre=True)
    assert torch.equal(d1, d2)


@pytest.mark.parametrize("layer", [LAYERS[0], LAYERS[3], LAYERS[6], LAYERS[7], LAYERS[8]],
                         ids=["enc2", "enc11", "tdnn6", "dec1", "dec5"])
@pytest.mark.parametrize("dt", ["bf16x3", torch.float32], ids=["bf16x3", "f32"])
def test_pingpong_conv_kernel(layer, dt):
    """the opt-in two-groups-in-anti-phase kernel (sa_conv_pp.hip, ops.conv_impl(pingpong=True)):
    forward + statistics slabs, data gradient, at a ragged length and at a training length"""
    from speech_anonymization_amd import ops
    ops.conv_impl(pingpong=True)
    try:
        _conv_case(layer, dt, 3, 517)
        _conv_case(layer, dt, 2, {32: 80640, 64: 40320, 128: 20160}[layer[1]])
    finally:
        ops.conv_impl(pingpong=False)


def test_pingpong_conv_kernel_in_the_model():
    """whole model forward + backward on the ping-pong kernel == on the default kernel (same
    products, another summation order of the statistics slabs)"""
    from oracle.convae import numpy_params
    from oracle.features import synthetic_feats
    from speech_anonymization_amd import ops
    from speech_anonymization_amd.convae import ConvAutoencoder
    feats = synthetic_feats(4, 144, seed=3).to(dev())
    gender = (torch.arange(4) % 2).to(dev())
    out = []
    for pp in (False, True):
        ops.conv_impl(pingpong=pp)
        try:
            m = ConvAutoencoder(precision="bf16x3", pooling_noise=None)
            m.load_state_dict(numpy_params(8886))
            m.to(dev()).train()
            recon, logp = m(feats)
            _, g_r = ops.recon_loss(recon.detach().contiguous(), feats.contiguous(), "l1")
            _, dn, _ = ops.cls_losses(logp.detach(), gender)
            torch.autograd.backward([recon, logp], [0.1 * g_r.view_as(recon), 0.9 * dn])
            torch.cuda.synchronize()
            out.append((recon.detach().clone(), logp.detach().clone(),
                        {k: p.grad.clone() for k, p in m.named_parameters()}))
        finally:
            ops.conv_impl(pingpong=False)
    from tests.test_convae_gpu import NULL_BIAS
    assert rel_mse(out[1][0], out[0][0]) < 1e-10 and rel_mse(out[1][1], out[0][1]) < 1e-8
    for k, g in out[0][2].items():
        if k not in NULL_BIAS:
            assert rel_mse(out[1][2][k], g) < 1e-5, k       # classifier branch amplifies the slab-order noise


@pytest.mark.parametrize("Ln,pad", [(20160, 2), (20001, 2), (20164, 0)], ids=["train", "ragged", "valid"])
@pytest.mark.parametrize("mode", ["plain", "swish_stats_cache", "relu_stats", "pro_stats", "two_affine"])
def test_weight_stationary_conv_kernel(mode, Ln, pad):
    """sa_conv_ws.hip (persistent, weights in registers, rows by LDS-DMA, epilogue / transform in the
    MFMA loop's issue gaps) serves the large 128->128 bf16x3 forward launches: same output BITS as the
    one-tile kernel (same operand split, same accumulation order), statistics equal up to the order
    of the in-tile sums, and both against the fp32 torch convolution.  `ragged`: a partial last tile;
    `valid`: no padding, Lout = Lin - 4 (the trailing input rows belong to the last tile)."""
    from speech_anonymization_amd import _lib as L, ops
    d, B = dev(), 6
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, Ln, 128, generator=g).to(d)
    w = (torch.randn(128, 128, 5, generator=g) * 0.05).to(d)
    bias = torch.randn(128, generator=g).to(d)
    s1 = (torch.rand(B, 128, generator=g) + 0.5).to(d)
    t1 = (torch.randn(B, 128, generator=g) * 0.1).to(d)
    wp = ops.pack_weights(w, "conv_fwd", torch.float32, L.BF16X3)
    Lout = Ln + 2 * pad - 4
    kw = dict(code=L.BF16X3)
    s2 = (torch.rand(128, generator=g) + 0.5).to(d)
    t2 = (torch.randn(128, generator=g) * 0.1).to(d)
    cached = mode in ("swish_stats_cache", "pro_stats", "two_affine")
    if mode == "swish_stats_cache":
        kw.update(s1=s1, t1=t1, swish=True, want_stats=True)
    elif mode == "relu_stats":
        kw.update(relu=True, want_stats=True)
    elif mode == "pro_stats":                    # decoder.0: also the statistics of its own transformed input
        kw.update(s1=s1, t1=t1, swish=True, want_pro_stats=True)
    elif mode == "two_affine":                   # the classifier's first TDNN layer: BatchNorm behind the activation, ReLU out
        kw.update(s1=s1, t1=t1, swish=True, s2=s2, t2=t2, relu=True, want_stats=True)

    def run(ws):
        ops.conv_impl(ws=ws)
        a_out = torch.full((B, Ln, 128), float("nan"), dtype=torch.bfloat16, device=d) if cached else None
        r = ops.conv_gemm(x, wp, bias if mode != "plain" else None, 128, 128, 1, 1, ops.taps_conv(5, 1, pad), Lout,
                          a_out=a_out, **kw)
        torch.cuda.synchronize()
        return (r if isinstance(r, tuple) else (r,)) + ((a_out,) if a_out is not None else ())

    try:
        ref, got = run(False), run(True)
        import ctypes as C
        a = L.SaConvArgs()                                           # (the comparison is not one-tile against one-tile)
        a.B, a.Lin, a.Lout = B, Ln, Lout
        a.taps = L.make_taps(ops.taps_conv(5, 1, pad))
        assert L.load().sa_conv_gemm_route(L.BF16X3, 128, 128, 1, 1, C.byref(a)) == 2
    finally:
        ops.conv_impl()
    assert torch.equal(ref[0], got[0])                               # y: bit-equal
    if mode != "plain":                                              # statistics / pro_stats slabs
        assert ref[1].shape == got[1].shape and torch.allclose(ref[1], got[1], rtol=2e-6, atol=1e-3)
    if cached:
        assert not torch.isnan(got[2].float()).any() and torch.equal(ref[2], got[2])
    xin = x
    if cached:
        xin = torch.nn.functional.silu(x * s1[:, None, :] + t1[:, None, :])
        if mode == "two_affine":
            xin = xin * s2 + t2
    yref = F.conv1d(xin.permute(0, 2, 1), w, bias if mode != "plain" else None, padding=pad).permute(0, 2, 1)
    if mode in ("relu_stats", "two_affine"):
        yref = yref.relu()
    if mode == "pro_stats":                                          # (sum, sumsq) of the transformed input per utterance, channel
        ps = got[1].double().sum(dim=1)
        assert torch.allclose(ps[..., 0], xin.double().sum(dim=1), rtol=1e-5, atol=1e-2)
        assert torch.allclose(ps[..., 1], (xin.double() ** 2).sum(dim=1), rtol=1e-5, atol=1e-2)
    assert rel_mse(got[0], yref) < 2e-9


@pytest.mark.parametrize("dil", [2, 3])
@pytest.mark.parametrize("affine", [True, False], ids=["bn_affine", "plain"])
def test_weight_stationary_conv_kernel_dilated(dil, affine):
    """the 3-tap dilated TDNN layers (models/ConvAutoEncoder.py:37-43: k3 d2, k3 d3, no padding) on
    the weight-stationary kernel: 144 MFMAs per tile, the per-channel BatchNorm affine of the layer
    below as prologue, ReLU + statistics + operand cache; output bits == the one-tile kernel"""
    from speech_anonymization_amd import _lib as L, ops
    d, B, Ln = dev(), 6, 20156
    g = torch.Generator().manual_seed(13 + dil)
    x = torch.randn(B, Ln, 128, generator=g).to(d)
    w = (torch.randn(128, 128, 3, generator=g) * 0.05).to(d)
    bias = torch.randn(128, generator=g).to(d)
    s2 = (torch.rand(128, generator=g) + 0.5).to(d)
    t2 = (torch.randn(128, generator=g) * 0.1).to(d)
    wp = ops.pack_weights(w, "conv_fwd", torch.float32, L.BF16X3)
    Lout = Ln - 2 * dil
    kw = dict(s2=s2, t2=t2) if affine else {}

    def run(ws):
        ops.conv_impl(ws=ws)
        a_out = torch.full((B, Ln, 128), float("nan"), dtype=torch.bfloat16, device=d)
        y, st = ops.conv_gemm(x, wp, bias, 128, 128, 1, 1, ops.taps_conv(3, dil, 0), Lout, relu=True, want_stats=True,
                              code=L.BF16X3, a_out=a_out, **kw)
        torch.cuda.synchronize()
        return y, st, a_out

    try:
        ref, got = run(False), run(True)
        import ctypes as C
        a = L.SaConvArgs()
        a.B, a.Lin, a.Lout = B, Ln, Lout
        a.taps = L.make_taps(ops.taps_conv(3, dil, 0))
        assert L.load().sa_conv_gemm_route(L.BF16X3, 128, 128, 1, 1, C.byref(a)) == 2
    finally:
        ops.conv_impl()
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[2], got[2])
    assert not torch.isnan(got[2].float()).any()
    assert torch.allclose(ref[1], got[1], rtol=2e-6, atol=1e-3)
    xin = x * s2 + t2 if affine else x
    yref = F.conv1d(xin.permute(0, 2, 1), w, bias, dilation=dil).permute(0, 2, 1).relu()
    assert rel_mse(got[0], yref) < 2e-9


@pytest.mark.parametrize("Ln", [40320, 40001], ids=["train", "ragged"])
@pytest.mark.parametrize("mode", ["plain", "swish_stats_cache", "swish_cache"])
def test_weight_stationary_conv_kernel_64ch(mode, Ln):
    """the 64 -> 64 forward layers (encoder.5, decoder.4: models/ConvAutoEncoder.py:146-148,167-169) on
    the weight-stationary kernel: two column blocks x two row halves of a 128-row tile per workgroup,
    the statistics of the two halves added after the tile barrier; output bits == the one-tile kernel"""
    from speech_anonymization_amd import _lib as L, ops
    d, B = dev(), 6
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, Ln, 64, generator=g).to(d)
    w = (torch.randn(64, 64, 5, generator=g) * 0.07).to(d)
    bias = torch.randn(64, generator=g).to(d)
    s1 = (torch.rand(B, 64, generator=g) + 0.5).to(d)
    t1 = (torch.randn(B, 64, generator=g) * 0.1).to(d)
    wp = ops.pack_weights(w, "conv_fwd", torch.float32, L.BF16X3)
    kw = dict(code=L.BF16X3)
    if mode != "plain":
        kw.update(s1=s1, t1=t1, swish=True, want_stats=mode == "swish_stats_cache")

    def run(ws):
        ops.conv_impl(ws=ws)
        a_out = torch.full((B, Ln, 64), float("nan"), dtype=torch.bfloat16, device=d) if mode != "plain" else None
        r = ops.conv_gemm(x, wp, bias, 64, 64, 1, 1, ops.taps_conv(5, 1, 2), Ln, a_out=a_out, **kw)
        torch.cuda.synchronize()
        return (r if isinstance(r, tuple) else (r,)) + ((a_out,) if a_out is not None else ())

    try:
        ref, got = run(False), run(True)
        import ctypes as C
        a = L.SaConvArgs()
        a.B, a.Lin, a.Lout = B, Ln, Ln
        a.taps = L.make_taps(ops.taps_conv(5, 1, 2))
        assert L.load().sa_conv_gemm_route(L.BF16X3, 64, 64, 1, 1, C.byref(a)) == 2
    finally:
        ops.conv_impl()
    assert torch.equal(ref[0], got[0])
    if mode == "swish_stats_cache":
        assert ref[1].shape == got[1].shape and torch.allclose(ref[1], got[1], rtol=2e-6, atol=1e-3)
    if mode != "plain":
        assert not torch.isnan(got[-1].float()).any() and torch.equal(ref[-1], got[-1])
    xin = torch.nn.functional.silu(x * s1[:, None, :] + t1[:, None, :]) if mode != "plain" else x
    yref = F.conv1d(xin.permute(0, 2, 1), w, bias, padding=2).permute(0, 2, 1)
    assert rel_mse(got[0], yref) < 2e-9


@pytest.mark.parametrize("Ln", [40320, 40003], ids=["train", "ragged"])
@pytest.mark.parametrize("mode", ["plain", "swish_stats_cache"])
def test_weight_stationary_conv_kernel_stride2(mode, Ln):
    """encoder.8 (64 -> 128, stride 2: models/ConvAutoEncoder.py:149) on the weight-stationary kernel:
    two staged input rows per output row (strided A-fragment rows), input side on 64 channels, output
    side on 128; output bits == the one-tile kernel"""
    from speech_anonymization_amd import _lib as L, ops
    d, B = dev(), 6
    g = torch.Generator().manual_seed(19)
    x = torch.randn(B, Ln, 64, generator=g).to(d)
    w = (torch.randn(128, 64, 5, generator=g) * 0.07).to(d)
    bias = torch.randn(128, generator=g).to(d)
    s1 = (torch.rand(B, 64, generator=g) + 0.5).to(d)
    t1 = (torch.randn(B, 64, generator=g) * 0.1).to(d)
    wp = ops.pack_weights(w, "conv_fwd", torch.float32, L.BF16X3)
    Lout = (Ln + 4 - 5) // 2 + 1
    kw = dict(code=L.BF16X3)
    if mode != "plain":
        kw.update(s1=s1, t1=t1, swish=True, want_stats=True)

    def run(ws):
        ops.conv_impl(ws=ws)
        a_out = torch.full((B, Ln, 64), float("nan"), dtype=torch.bfloat16, device=d) if mode != "plain" else None
        r = ops.conv_gemm(x, wp, bias, 64, 128, 2, 1, ops.taps_conv(5, 1, 2), Lout, a_out=a_out, **kw)
        torch.cuda.synchronize()
        return (r if isinstance(r, tuple) else (r,)) + ((a_out,) if a_out is not None else ())

    try:
        ref, got = run(False), run(True)
        import ctypes as C
        a = L.SaConvArgs()
        a.B, a.Lin, a.Lout = B, Ln, Lout
        a.taps = L.make_taps(ops.taps_conv(5, 1, 2))
        assert L.load().sa_conv_gemm_route(L.BF16X3, 64, 128, 2, 1, C.byref(a)) == 2
    finally:
        ops.conv_impl()
    assert torch.equal(ref[0], got[0])
    if mode != "plain":
        assert torch.allclose(ref[1], got[1], rtol=2e-6, atol=1e-3)
        assert not torch.isnan(got[2].float()).any() and torch.equal(ref[2], got[2])
    xin = torch.nn.functional.silu(x * s1[:, None, :] + t1[:, None, :]) if mode != "plain" else x
    yref = F.conv1d(xin.permute(0, 2, 1), w, bias, stride=2, padding=2).permute(0, 2, 1)
    assert rel_mse(got[0], yref) < 2e-9


@pytest.mark.parametrize("cin,cout,Lin", [(128, 64, 20160), (128, 64, 20003), (64, 32, 40320), (64, 32, 40000), (64, 32, 40003)],
                         ids=["128-64-train", "128-64-ragged", "64-32-train", "64-32-odd-slabs", "64-32-ragged"])
@pytest.mark.parametrize("mode", ["plain", "stats_cache"])
def test_weight_stationary_conv_kernel_transposed(mode, cin, cout, Lin):
    """decoder.1 / decoder.5 (ConvTranspose1d 128 -> 64 and 64 -> 32, k5 s2 p2 op1:
    models/ConvAutoEncoder.py:161-163,170-172) on the weight-stationary kernel: wave pairs of a column
    block are the two output phases (3 and 2 taps, the missing tap runs zero fragments), their
    statistics added behind the tile barrier; 64 -> 32: two row halves per tile = two statistics slabs
    (an odd slab count per utterance leaves the last tile with one); output bits == the one-tile kernel"""
    from speech_anonymization_amd import _lib as L, ops
    d, B = dev(), 6
    g = torch.Generator().manual_seed(23)
    x = torch.randn(B, Lin, cin, generator=g).to(d)
    w = (torch.randn(cin, cout, 5, generator=g) * 0.05).to(d)
    bias = torch.randn(cout, generator=g).to(d)
    wp = ops.pack_weights(w, "convT_fwd", torch.float32, L.BF16X3)
    Lout = 2 * Lin
    kw = dict(code=L.BF16X3)
    if mode != "plain":
        kw.update(want_stats=True)

    def run(ws):
        ops.conv_impl(ws=ws)
        a_out = torch.full((B, Lin, cin), float("nan"), dtype=torch.bfloat16, device=d) if mode != "plain" else None
        r = ops.conv_gemm(x, wp, bias, cin, cout, 1, 2, ops.UP2, Lout, a_out=a_out, **kw)
        torch.cuda.synchronize()
        return (r if isinstance(r, tuple) else (r,)) + ((a_out,) if a_out is not None else ())

    try:
        ref, got = run(False), run(True)
        import ctypes as C
        a = L.SaConvArgs()
        a.B, a.Lin, a.Lout = B, Lin, Lout
        a.taps = L.make_taps(ops.UP2)
        assert L.load().sa_conv_gemm_route(L.BF16X3, cin, cout, 1, 2, C.byref(a)) == 2
    finally:
        ops.conv_impl()
    assert torch.equal(ref[0], got[0])
    if mode != "plain":
        assert ref[1].shape == got[1].shape and torch.allclose(ref[1], got[1], rtol=2e-6, atol=1e-3)
        assert not torch.isnan(got[2].float()).any() and torch.equal(ref[2], got[2])
    yref = F.conv_transpose1d(x.permute(0, 2, 1), w, bias, stride=2, padding=2, output_padding=1).permute(0, 2, 1)
    assert rel_mse(got[0], yref) < 2e-9


def test_weight_stationary_conv_kernel_random_lengths():
    """seeded sweep over utterance counts and lengths (tile ranges that start / end anywhere inside an
    utterance, partial last tiles, single-tile tails) for every layer form the weight-stationary kernel
    serves: output, statistics slabs and operand cache against the one-tile kernel"""
    import ctypes as C
    import random
    from speech_anonymization_amd import _lib as L, ops
    d = dev()
    rng = random.Random(20260401)
    # (cin, cout, sa, u, taps-or-phases, weight kind, weight shape, base length range)
    forms = [
        (128, 128, 1, 1, lambda: ops.taps_conv(5, 1, 2), "conv_fwd", (128, 128, 5), (17000, 21000), True),
        (128, 128, 1, 1, lambda: ops.taps_conv(3, 3, 0), "conv_fwd", (128, 128, 3), (17000, 21000), False),
        (64, 64, 1, 1, lambda: ops.taps_conv(5, 1, 2), "conv_fwd", (64, 64, 5), (34000, 41000), True),
        (64, 128, 2, 1, lambda: ops.taps_conv(5, 1, 2), "conv_fwd", (128, 64, 5), (34000, 41000), True),
        (128, 64, 1, 2, lambda: ops.UP2, "convT_fwd", (128, 64, 5), (17000, 21000), False),
    ]
    try:
        for cin, cout, sa, u, mk, kind, wshape, (lo, hi), swish in forms:
            for _ in range(2):
                B, Lin = rng.randint(6, 9), rng.randint(lo, hi)
                taps = mk()
                offs = [o for ph in taps for (o, _) in ph]
                Lout = 2 * Lin if u == 2 else (Lin + (-2 * min(offs) if min(offs) < 0 else 0) - (max(offs) - min(offs)) - 1) // sa + 1
                if u == 1 and min(offs) == 0:
                    Lout = (Lin - max(offs) - 1) // sa + 1
                g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
                x = torch.randn(B, Lin, cin, generator=g).to(d)
                w = (torch.randn(*wshape, generator=g) * 0.06).to(d)
                bias = torch.randn(cout, generator=g).to(d)
                s1 = (torch.rand(B, cin, generator=g) + 0.5).to(d)
                t1 = (torch.randn(B, cin, generator=g) * 0.1).to(d)
                wp = ops.pack_weights(w, kind, torch.float32, L.BF16X3)
                kw = dict(s1=s1, t1=t1, swish=True) if swish else {}
                outs = []
                for ws in (False, True):
                    ops.conv_impl(ws=ws)
                    a_out = torch.full((B, Lin, cin), float("nan"), dtype=torch.bfloat16, device=d)
                    y, st = ops.conv_gemm(x, wp, bias, cin, cout, sa, u, taps, Lout, want_stats=True, code=L.BF16X3,
                                          a_out=a_out, **kw)
                    torch.cuda.synchronize()
                    outs.append((y, st, a_out))
                a = L.SaConvArgs()
                a.B, a.Lin, a.Lout = B, Lin, Lout
                a.taps = L.make_taps(taps)
                case = (cin, cout, sa, u, B, Lin, Lout)
                assert L.load().sa_conv_gemm_route(L.BF16X3, cin, cout, sa, u, C.byref(a)) == 2, case
                assert torch.equal(outs[0][0], outs[1][0]), case
                assert torch.equal(outs[0][2], outs[1][2]) and not torch.isnan(outs[1][2].float()).any(), case
                assert torch.allclose(outs[0][1], outs[1][1], rtol=2e-6, atol=1e-3), case
    finally:
        ops.conv_impl()


@pytest.mark.parametrize("B", [10, 32, 64, 3])
def test_fused_fc_head_equals_the_separate_launches(B):
    """sa_head_fwd / sa_head_bwd (the FC head of the sex classifier, models/ConvAutoEncoder.py:47-55,68, as
    one forward and one backward launch) against the chain of separate launches they replace (3 dense, column
    sums, BatchNorm finalisers, log-softmax; ~25 launches backward) and against torch autograd on the CPU:
    outputs, BatchNorm running statistics, every parameter gradient and d pooled."""
    from speech_anonymization_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(B)
    cls = torch.nn.Sequential(torch.nn.Linear(256, 128), torch.nn.ReLU(), torch.nn.BatchNorm1d(128),
                              torch.nn.Linear(128, 64), torch.nn.ReLU(), torch.nn.BatchNorm1d(64),
                              torch.nn.Linear(64, 2))
    with torch.no_grad():
        for p_ in cls.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.3 if p_.dim() == 1 else p_.shape[1] ** -0.5))
        cls[2].weight.add_(1.0); cls[5].weight.add_(1.0)
    pooled = torch.randn(B, 256, generator=g)
    dlogp = torch.randn(B, 2, generator=g) / B
    # CPU reference (train-mode BatchNorm)
    ref = __import__("copy").deepcopy(cls).train()
    xr = pooled.clone().requires_grad_(True)
    lp = torch.log_softmax(ref(xr), dim=1)
    lp.backward(dlogp)
    P = {k: v.detach().to(d).contiguous() for k, v in cls.state_dict().items() if v.dtype.is_floating_point and "running" not in k}
    bn1, bn2 = __import__("copy").deepcopy(cls[2]).to(d), __import__("copy").deepcopy(cls[5]).to(d)
    pd, dl = pooled.to(d), dlogp.to(d)
    H1, f1, H2, f2, logp = ops.head_fwd(pd, P, bn1, bn2)
    grads = {k: torch.full_like(v, float("nan")) for k, v in P.items()}
    dpooled = ops.head_bwd(dl, logp, pd, H1, f1, H2, f2, P, grads)
    torch.cuda.synchronize()
    assert rel_mse(logp, lp) < (1e-10 if B >= 10 else 2e-9)      # (BatchNorm over 3 rows amplifies the rounding)
    assert rel_mse(bn1.running_mean, ref[2].running_mean) < 1e-10 and rel_mse(bn1.running_var, ref[2].running_var) < 1e-10
    assert rel_mse(bn2.running_mean, ref[5].running_mean) < 1e-10 and rel_mse(bn2.running_var, ref[5].running_var) < 1e-10
    assert rel_mse(dpooled, xr.grad) < (1e-9 if B >= 10 else 1e-7)
    for k, v in ref.named_parameters():
        assert rel_mse(grads[k], v.grad) < (1e-9 if B >= 10 else 1e-7), k
    # the separate launches (what the model runs under SyncBatchNorm, in eval mode and for B > 64)
    sH1 = ops.dense(pd, P["0.weight"], P["0.bias"], 128, 256, relu=True)
    b1s = __import__("copy").deepcopy(cls[2]).to(d)
    sf1 = ops.fin_bn_fwd(ops.colsums(sH1), 128, B, P["2.weight"], P["2.bias"], b1s.running_mean, b1s.running_var)
    sH2 = ops.dense(sH1, P["3.weight"], P["3.bias"], 64, 128, ps=sf1[2], pt=sf1[3], relu=True)
    b2s = __import__("copy").deepcopy(cls[5]).to(d)
    sf2 = ops.fin_bn_fwd(ops.colsums(sH2), 64, B, P["5.weight"], P["5.bias"], b2s.running_mean, b2s.running_var)
    slogp = ops.log_softmax(ops.dense(sH2, P["6.weight"], P["6.bias"], 2, 64, ps=sf2[2], pt=sf2[3]))
    torch.cuda.synchronize()
    assert rel_mse(logp, slogp) < (1e-10 if B >= 10 else 2e-9) and rel_mse(H1, sH1) < 1e-11 and rel_mse(H2, sH2) < 1e-9
    for i in range(4):
        assert rel_mse(f1[i], sf1[i]) < 1e-10 and rel_mse(f2[i], sf2[i]) < (1e-10 if B >= 10 else 2e-9)


WSD_CASES = [
    # name, K, dil, pad, Lin, prologue (None | "in" | "bn"), epilogue variant, second gradient
    ("enc11", 5, 1, 2, 20160, "in", 1, False),      # InstanceNorm prologue, (acc) * swish'(z)
    ("tdnn0", 5, 1, 0, 20156, "bn", 3, False),      # BatchNorm + ReLU-mask prologue, xhat from swish(z)
    ("dec0", 5, 1, 2, 20160, None, 1, True),        # plain rows, pending BatchNorm apply as addend
    ("tdnn3", 3, 2, 0, 20152, "bn", 2, False),      # 3 taps dilation 2, partial last tile
    ("tdnn6", 3, 3, 0, 20146, "bn", 2, False),      # 3 taps dilation 3
    ("enc11_ragged", 5, 1, 2, 20001, "in", 1, False),
    ("tdnn3_short", 3, 2, 0, 5003, "bn", 2, False), # many utterances: tile ranges cross utterance ends
]


@pytest.mark.parametrize("case", WSD_CASES, ids=[c[0] for c in WSD_CASES])
def test_weight_stationary_fused_data_gradient_kernel(case):
    """sa_conv_wsd.hip (persistent, weights in registers, rows by LDS-DMA, prologue / epilogue in the
    MFMA loop's issue gaps, values loaded across the loop's back edge in reserved registers) serves
    the five 128 -> 128 fused data gradients of a step: same y and bf16 d y BITS as the one-tile
    kernel (sa_conv_gemm_kernel<bf16x3_t,128,128,1,1,64,PRO2>: same operand split, accumulation order
    and epilogue operations), column sums bit-equal, statistics equal up to the order of the in-tile
    sums; run twice: a stale in-flight load would show as a run-to-run difference."""
    import ctypes as C
    from speech_anonymization_amd import _lib as L, ops
    name, K, dil, pad, Lin, nbk, epk, g2 = case
    d = dev()
    B = 24 if "short" in name else 6
    g = torch.Generator().manual_seed(3)
    Lout = Lin + dil * (K - 1) - 2 * pad
    x = torch.randn(B, Lin, 128, generator=g).to(d)
    w = (torch.randn(128, 128, K, generator=g) * 0.05).to(d)
    wd = ops.pack_weights(w, "conv_dgrad", torch.float32, L.BF16X3)
    taps = ops.taps_conv_dgrad_s1(K, dil, pad)
    kw = dict(code=L.BF16X3, want_stats=True)
    if nbk:
        per_c = nbk == "bn"
        shp = (128,) if per_c else (B, 128)
        c = [(torch.rand(*shp, generator=g) + 0.5).to(d), (torch.randn(*shp, generator=g) * 0.1).to(d),
             (torch.randn(*shp, generator=g) * 0.05).to(d)]
        y2 = torch.randn(B, Lin, 128, generator=g).to(d)
        kw["nb"] = dict(x=y2, c1=c[0], c2=c[1], c3=c[2], per_c=per_c, relu_mask=per_c, want_colsum=True)
    xe = torch.randn(B, Lout, 128, generator=g).to(d)
    s1 = (torch.rand(B, 128, generator=g) + 0.5).to(d)
    t1 = (torch.randn(B, 128, generator=g) * 0.1).to(d)
    if epk == 1:
        mean, rstd = (torch.randn(B, 128, generator=g) * 0.1).to(d), (torch.rand(B, 128, generator=g) + 0.5).to(d)
        kw["ep"] = dict(mode=1, x=xe, s1=s1, t1=t1, mean=mean, rstd=rstd)
        if g2:
            kw["ep"]["g2"] = torch.randn(B, Lout, 128, generator=g).to(d)
            kw["ep"]["g2k"] = [(torch.rand(128, generator=g) + 0.5).to(d), (torch.randn(128, generator=g) * 0.1).to(d),
                               (torch.randn(128, generator=g) * 0.05).to(d)]
    else:
        mean, rstd = (torch.randn(128, generator=g) * 0.1).to(d), (torch.rand(128, generator=g) + 0.5).to(d)
        kw["ep"] = dict(mode=2, x=xe, mean=mean, rstd=rstd, per_c=True)
        if epk == 3:
            kw["ep"].update(s1=s1, t1=t1, xp_is_act=True)

    def run(ws):
        ops.conv_impl(ws=ws)
        ao = torch.full((B, Lin, 128), float("nan"), device=d, dtype=torch.bfloat16) if nbk else None
        out = ops.conv_gemm(x, wd, None, 128, 128, 1, 1, taps, Lout, a_out=ao, **kw)
        torch.cuda.synchronize()
        return tuple(out) + ((ao,) if ao is not None else ())

    try:
        ref, got, again = run(False), run(True), run(True)
        a = L.SaConvArgs()
        a.B, a.Lin, a.Lout = B, Lin, Lout
        a.taps = L.make_taps(taps)
        a.stats, a.ep_x, a.ep_mean, a.ep_rstd = ops._f(ref[1]), ops._f(xe), ops._f(mean), ops._f(rstd)
        a.ep_mode, a.ep_bstride, a.ep_xp_is_act = (1 if epk == 1 else 2), (128 if epk == 1 else 0), int(epk == 3)
        if epk in (1, 3):
            a.ep_s1, a.ep_t1 = ops._f(s1), ops._f(t1)
        if g2:
            a.ep_g2 = ops._f(kw["ep"]["g2"])
            a.ep_g2k1, a.ep_g2k2, a.ep_g2k3 = (ops._f(t) for t in kw["ep"]["g2k"])
        if nbk:
            a.nb_x, a.nb_c1, a.nb_c2, a.nb_c3 = ops._f(y2), ops._f(c[0]), ops._f(c[1]), ops._f(c[2])
            a.nb_relu_mask = int(nbk == "bn")
        assert L.load().sa_conv_gemm_route(L.BF16X3, 128, 128, 1, 1, C.byref(a)) == 3
    finally:
        ops.conv_impl()
    assert torch.equal(ref[0], got[0]) and torch.equal(got[0], again[0])                 # y
    assert torch.allclose(ref[1], got[1], rtol=2e-5, atol=2e-3) and torch.equal(got[1], again[1])   # statistics slabs
    if nbk:
        assert torch.equal(ref[2], got[2])                                               # column sums of d y
        assert not torch.isnan(got[3].float()).any() and torch.equal(ref[3], got[3])     # bf16 d y
    # anchor: the one-tile kernel against fp32 torch (plain rows, mode 1: dx = conv_transpose(dy) * swish'(z) ...)
    if name == "dec0":
        dyr = F.conv_transpose1d(x.permute(0, 2, 1), w, padding=pad).permute(0, 2, 1)
        z = xe * s1[:, None, :] + t1[:, None, :]
        sg = torch.sigmoid(z)
        k1, k2, k3 = kw["ep"]["g2k"]
        g2v = k1 * kw["ep"]["g2"] + k2 * (z * sg) + k3
        want = (dyr + g2v) * (sg * (1 + z * (1 - sg)))
        assert rel_mse(got[0], want) < 2e-9


def test_weight_stationary_conv_routing():
    """what goes to the weight-stationary kernel: bf16x3 128->128 stride-1 5-tap launches with at
    least 1536 tiles (six per CU) and no fused backward epilogue / normalisation-backward prologue;
    everything else to the one-tile kernel"""
    import ctypes as C
    from speech_anonymization_amd import _lib as L, ops
    lib = L.load()
    ops.conv_impl()
    a = L.SaConvArgs()
    a.B, a.Lin, a.Lout = 32, 20160, 20160
    a.taps = L.make_taps(ops.taps_conv(5, 1, 2))
    route = lambda code=L.BF16X3, cin=128, cout=128: lib.sa_conv_gemm_route(code, cin, cout, 1, 1, C.byref(a))
    assert route() == 2
    assert route(L.F32) == 0 and route(L.BF16X3, 64, 128) == 0 and route(L.BF16X3, 32, 64) == 0   # (64 -> 128 is a stride-2 layer)
    a.B = 4
    assert route() == 0                                              # 1260 tiles: under six per CU
    a.B = 32
    a.ep_mode = 1
    assert route() == 0
    a.ep_mode = 0
    a.taps = L.make_taps(ops.taps_conv(3, 2, 0))                      # 3 taps over 4 rows: covered
    assert route() == 2
    a.taps = L.make_taps(ops.taps_conv(3, 4, 0))                      # over 8 rows: not
    assert route() == 0
    ops.conv_impl(ws=False)
    a.taps = L.make_taps(ops.taps_conv(5, 1, 2))
    try:
        assert route() == 0
    finally:
        ops.conv_impl()


def _fp8_e4m3(t):
    """OCP e4m3 round trip (saturating), the quantisation of the SA_FP8 operand path"""
    return t.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_conv_forward_fp8_operands(layer):
    """SA_FP8 (BASELINE config 5): bf16 storage, e4m3 MFMA operands.  Against F.conv1d on operands
    quantised the same way on the CPU (activations after the prologue transform, weights times their
    per-tensor power-of-two scale): the products are then exact in fp32, so the kernel must match to
    the rounding of its bf16 output -- which also pins the e4m3 fragment layout of
    v_mfma_f32_32x32x16_fp8_fp8."""
    from speech_anonymization_amd import _lib as L, ops
    name, cin, cout, K, stride, dil, pad, tr = layer
    B, Lin = 2, 333
    x = rnd(torch.bfloat16, B, cin, Lin, seed=1)
    wshape = (cin, cout, K) if tr else (cout, cin, K)
    w = rnd(torch.float32, *wshape, seed=2, scale=(cin * K) ** -0.5)
    bias = rnd(torch.float32, cout, seed=3, scale=0.1)
    s1, t1 = 1 + 0.2 * rnd(torch.float32, B, cin, seed=7), 0.3 * rnd(torch.float32, B, cin, seed=8)
    z = x * s1[:, :, None] + t1[:, :, None]
    a = _fp8_e4m3(z * torch.sigmoid(z))
    scale = 2.0 ** torch.floor(torch.log2(448.0 / w.abs().max()))
    wq = _fp8_e4m3(w * scale) / scale
    y = ref_fwd(a, wq, bias, stride, dil, pad, tr)
    Lout = y.shape[2]
    xd, wd = cl(x, torch.bfloat16), w.to(dev())
    kind = "convT_fwd" if tr else "conv_fwd"
    wp = ops.pack_weights(wd, kind, torch.bfloat16, L.FP8)
    args = (cin, cout, 1, 2, ops.UP2, Lout) if tr else (cin, cout, stride, 1, ops.taps_conv(K, dil, pad), Lout)
    yd, st = ops.conv_gemm(xd, wp, bias.to(dev()), *args, s1=s1.to(dev()), t1=t1.to(dev()), swish=True,
                           want_stats=True, code=L.FP8)
    torch.cuda.synchronize()
    assert abs(float(wp[-4:].view(torch.float32)) - float(scale)) == 0.0       # the scale behind the image
    assert rel_mse(uncl(yd), y) < 1e-5, name
    s = ops.sum_partials(st, B).view(B, cout, 2).cpu()
    yy = yd.double().cpu()
    assert torch.allclose(s[..., 0], yy.sum(dim=1), rtol=1e-3, atol=1e-2)


def test_model_fp8_precision_report():
    """precision="fp8" end to end (forward convolutions on e4m3 operands, gradients on the bf16
    kernels): what it costs against the fp32 oracle, next to the bf16 mode on the same input.  The
    numbers are printed and held to loose bounds: e4m3 has 3 mantissa bits, this mode is a
    throughput / memory point (BASELINE config 5), not a parity mode."""
    from oracle.convae import numpy_params
    from oracle.features import synthetic_feats
    from tests.test_convae_gpu import run_oracle, run_hip, hip_model, cosine
    B, T = 4, 144
    feats = synthetic_feats(B, T, seed=31)
    target = feats + 0.1 * torch.randn(B, T, 80, generator=torch.Generator().manual_seed(1))
    gender = torch.arange(B) % 2
    params = numpy_params(8886)
    o_recon, o_logp, o_loss, o_grads, _ = run_oracle(params, feats, target, gender, "l1")
    rows = {}
    for prec in ("bf16", "fp8"):
        m = hip_model(prec, params)
        recon, logp, loss, grads = run_hip(m, feats, target, gender, "l1")
        rows[prec] = dict(recon=rel_mse(recon.float(), o_recon), loss=abs(loss - o_loss),
                          dec=rel_mse(grads["decoder.4.weight"], o_grads["decoder.4.weight"]),
                          dec_cos=cosine(grads["decoder.4.weight"], o_grads["decoder.4.weight"]),
                          enc_cos=cosine(grads["encoder.5.weight"], o_grads["encoder.5.weight"]))
        print(prec, {k: f"{v:.3e}" for k, v in rows[prec].items()})
    assert rows["fp8"]["recon"] < 5e-2 and rows["fp8"]["dec_cos"] > 0.9
    assert rows["bf16"]["recon"] < 1e-3


@pytest.mark.parametrize("B,nslab,Cc", [(3, 7, 64), (32, 315, 128), (5, 700, 32)])
def test_reduce_finalize_equals_the_separate_launches(B, nslab, Cc):
    """sa_reduce_finalize (slab sums + finaliser in one launch, last-arriving workgroup for the sums
    over utterances) against sa_sum_partials -> sa_fin_*: same fp64 sums in the same order, so the
    results are BIT-identical -- every mode, twice (the self-resetting tickets)."""
    from speech_anonymization_amd import _lib as L, ops
    d = dev()
    g = torch.Generator().manual_seed(B * 1000 + nslab)
    part = torch.randn(B, nslab, Cc, 2, generator=g).to(d)
    part[..., 1] = part[..., 1].abs() * 3 + 1.0            # a plausible sum of squares
    gamma = (1 + 0.1 * torch.randn(Cc, generator=g)).to(d)
    beta = (0.1 * torch.randn(Cc, generator=g)).to(d)
    n_in, n_bn = 4096.0, 4096.0 * B

    def same(a, b):
        return torch.allclose(a, b, rtol=3e-7, atol=1e-12)
    for rep in range(2):
        # InstanceNorm forward
        rows = ops.sum_partials(part, B)
        ref = ops.fin_in_fwd(rows, B, Cc, int(n_in), gamma, beta)
        got = ops.reduce_finalize(L.FIN_IN_FWD, part, B, Cc, count=n_in, gamma=gamma, beta=beta)
        for a, b in zip(got, ref):
            assert same(a.reshape(-1), b.reshape(-1))
        mean, rstd = ref[0].contiguous(), ref[1].contiguous()
        # InstanceNorm backward (+ d gamma / d beta over utterances)
        dg0, db0, dg1, db1 = (torch.empty(Cc, device=d) for _ in range(4))
        ref = ops.fin_norm_bwd(rows, rows, B * Cc, Cc, n_in, gamma, mean, rstd, dgamma=dg0, dbeta=db0)
        got = ops.reduce_finalize(L.FIN_IN_BWD, part, B, Cc, count=n_in, gamma=gamma, mean=mean, rstd=rstd,
                                  dgamma=dg1, dbeta=db1)
        for a, b in zip(got, ref):
            assert same(a.reshape(-1), b.reshape(-1))
        assert torch.equal(dg0, dg1) and torch.equal(db0, db1)
        # BatchNorm forward (per-utterance rows, then over utterances) incl. running statistics
        rm0, rv0 = torch.zeros(Cc, device=d), torch.ones(Cc, device=d)
        rm1, rv1 = torch.zeros(Cc, device=d), torch.ones(Cc, device=d)
        ref = ops.fin_bn_fwd(rows, Cc, n_bn, gamma, beta, rm0, rv0)
        got = ops.reduce_finalize(L.FIN_BN_FWD, part, B, Cc, count=n_bn, gamma=gamma, beta=beta,
                                  run_mean=rm1, run_var=rv1)
        for a, b in zip(got, ref):
            assert same(a, b)
        assert same(rm0, rm1) and same(rv0, rv1)
        bm, br = ref[0].contiguous(), ref[1].contiguous()
        # BatchNorm backward with the GradReverse sign
        ref = ops.fin_norm_bwd(rows, rows, Cc, Cc, n_bn, gamma, bm, br, sign=-1.0, dgamma=dg0, dbeta=db0)
        got = ops.reduce_finalize(L.FIN_BN_BWD, part, B, Cc, count=n_bn, gamma=gamma, mean=bm, rstd=br,
                                  sign=-1.0, dgamma=dg1, dbeta=db1)
        for a, b in zip(got, ref):
            assert same(a, b)
        assert torch.equal(dg0, dg1) and torch.equal(db0, db1)
        # bias gradients: two-component slabs and single-component column sums
        b0, b1 = torch.empty(Cc, device=d), torch.empty(Cc, device=d)
        ops.fin_bias(rows, B, Cc, b0)
        ops.reduce_finalize(L.FIN_BIAS, part, B, Cc, db=b1)
        assert torch.equal(b0, b1)
        cs = part[..., 0].contiguous()
        ops.fin_bias(ops.sum_partials(cs.view(B, nslab, Cc, 1), B), B, Cc, b0, ncomp=1)
        ops.reduce_finalize(L.FIN_BIAS, cs, B, Cc, ncomp=1, db=b1)
        assert torch.equal(b0, b1)
    torch.cuda.synchronize()
    assert all(int(tk.abs().sum()) == 0 for tk in ops._tickets.values())   # every ticket went back to zero


@pytest.mark.parametrize("shape", [(5, 128, 128), (3, 128, 128), (5, 64, 64), (5, 32, 64), (3, 64, 32)],
                         ids=lambda s: "x".join(map(str, s)))
def test_wgrad_reduce_row_widths_and_tails(shape):
    """sa_wgrad_reduce alone: every row width of the reducer (4 / 2 / 1 outputs per lane, chosen from
    ntaps*cin*cout), slab counts that end inside, at and just behind the unrolled eight-load group,
    the PyTorch-layout scatter (Conv1d [co][ci][k] and ConvTranspose1d [ci][co][k] strides) and
    accumulate -- against a float64 sum on the host"""
    import ctypes as C
    from speech_anonymization_amd import _lib as L
    lib = L.load()
    nt, cin, cout = shape
    per = nt * cin * cout
    g = torch.Generator().manual_seed(per)
    for nslab in (1, 3, 31, 32, 33, 67, 260):
        slabs = torch.randn(nslab, nt, cin, cout, generator=g)
        want = slabs.double().sum(0)                                   # [t][ci][co]
        for layout in ("conv", "convT"):
            if layout == "conv":                                       # dst [co][ci][k]
                dst0 = torch.randn(cout, cin, nt, generator=g)
                sk, sn, st = nt, cin * nt, 1
                ref = want.permute(2, 1, 0)
            else:                                                      # dst [ci][co][k]
                dst0 = torch.randn(cin, cout, nt, generator=g)
                sk, sn, st = cout * nt, nt, 1
                ref = want.permute(1, 2, 0)
            for acc in (0, 1):
                dst = dst0.clone().to(dev())
                sl = slabs.to(dev()).contiguous()
                rc = lib.sa_wgrad_reduce(L.ptr(sl), L.ptr(dst), nslab, nt, cin, cout, sk, sn, st, acc, L.stream())
                assert rc == 0
                exp = (ref + dst0.double() if acc else ref).float()
                torch.testing.assert_close(dst.cpu(), exp, rtol=2e-6, atol=2e-6 * max(1.0, nslab ** 0.5))


def test_colsums_writes_fp32_gradient_views():
    """sa_colsums out0 / out1: the fp64 column sums rounded to fp32, written into (possibly offset)
    views of a flat bucket -- what the FC-head bias / BatchNorm-affine gradients use instead of a
    copy launch; bit-equal to converting the fp64 sums"""
    from speech_anonymization_amd import ops
    g = torch.Generator().manual_seed(11)
    M, N = 32, 128
    X, H = torch.randn(M, N, generator=g).to(dev()), torch.randn(M, N, generator=g).to(dev())
    hm, hr = torch.randn(N, generator=g).to(dev()), (torch.rand(N, generator=g) + 0.5).to(dev())
    flat = torch.full((3 * N + 5,), 7.0, device=dev())
    o0, o1 = flat[5:5 + N], flat[5 + 2 * N:5 + 3 * N]
    s = ops.colsums(X, H, hm, hr, out0=o0, out1=o1)
    assert torch.equal(o0, s[:, 0].float()) and torch.equal(o1, s[:, 1].float())
    assert bool((flat[:5] == 7).all()) and bool((flat[5 + N:5 + 2 * N] == 7).all())
    ref0 = X.double().sum(0)
    ref1 = (X.double() * ((H.double() - hm.double()) * hr.double())).sum(0)
    torch.testing.assert_close(s[:, 0], ref0, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(s[:, 1], ref1, rtol=1e-5, atol=1e-5)
    s2 = ops.colsums(X)                                   # no outputs: sums only (sum, sum of squares)
    torch.testing.assert_close(s2[:, 1], (X.double() ** 2).sum(0), rtol=1e-6, atol=1e-6)


def test_batched_bias_gradients_equal_the_per_layer_launches():
    """sa_bias_multi (the bias gradients of a backward stage in two launches) against sa_sum_partials +
    sa_fin_bias per layer: same lanes, same order -> the same BITS; more records than one launch pair takes
    (SA_BIAS_MAX = 8), mixed slab layouts ([.., C, 2] statistics slabs and [.., C] column sums), ragged sizes."""
    from speech_anonymization_amd import ops
    torch.manual_seed(11)
    d = dev()
    shapes = [(32, 315, 128, 2), (32, 315, 128, 1), (10, 630, 64, 2), (3, 1260, 32, 1), (1, 7, 64, 2), (6, 17, 128, 1),
              (32, 315, 64, 1), (4, 40, 32, 2), (5, 9, 128, 2), (2, 3, 64, 1)]
    items, ref = [], []
    for nb, ns, cc, ncomp in shapes:
        part = torch.randn(nb, ns, cc, ncomp, device=d) * (1.0 + torch.rand(1, device=d) * 100)
        db = torch.full((cc,), float("nan"), device=d)
        items.append((part, nb, cc, ncomp, db))
        r = torch.empty(cc, device=d)
        ops.fin_bias(ops.sum_partials(part, nb, n=cc * ncomp), nb, cc, r, ncomp=ncomp)
        ref.append(r)
    ops.bias_multi(items)
    torch.cuda.synchronize()
    for (part, nb, cc, ncomp, db), r in zip(items, ref):
        assert torch.equal(db, r), (nb, cc, ncomp)
        exact = part[..., 0].double().sum(dim=(0, 1))
        assert float((db.double() - exact).abs().max()) <= 1e-6 * float(exact.abs().max() + 1.0)


def test_batched_wgrad_reducers_equal_the_single_launches():
    """sa_wgrad_reduce_multi (the split-K reducers of a backward stage in one launch) against one
    sa_wgrad_reduce per layer: the same kernel body per record -> the same BITS; the three row widths
    (VEC 4 / 2 / 1 by layer size), PyTorch and transposed destination strides, more records than one launch."""
    from speech_anonymization_amd import ops, _lib as L
    import ctypes as C
    torch.manual_seed(5)
    d = dev()
    lib = L.load()
    layers = [(5, 128, 128, 300), (3, 128, 128, 256), (5, 64, 64, 512), (5, 64, 128, 200), (5, 32, 64, 77), (5, 64, 32, 130),
              (5, 128, 64, 256), (15, 1, 32, 64), (3, 128, 128, 9), (5, 64, 64, 33)]
    items, ref = [], []
    for nt, cin, cout, nslab in layers:
        slabs = torch.randn(nslab, nt, cin, cout, device=d)
        transposed = (cin + cout + nt) % 2 == 0
        shape = (cin, cout, nt) if transposed else (cout, cin, nt)
        sk, sn, st = (cout * nt, nt, 1) if transposed else (nt, cin * nt, 1)
        dst = torch.full(shape, float("nan"), device=d)
        r = torch.empty(shape, device=d)
        L.check(lib.sa_wgrad_reduce(L.ptr(slabs), L.ptr(r), nslab, nt, cin, cout, sk, sn, st, 0, L.stream()), "sa_wgrad_reduce")
        items.append((slabs, dst, nslab, nt, cin, cout, sk, sn, st, 0))
        ref.append(r)
    ops.wgrad_reduce_multi(items)
    torch.cuda.synchronize()
    for it, r in zip(items, ref):
        assert torch.equal(it[1], r), it[2:6]


@pytest.mark.parametrize("scale", [0.01, 40.0])
def test_fused_gradient_clip_equals_torch(scale):
    """sa_clip_grads (clip_grad_norm_ on flat gradient buffers, two launches) against torch.nn.utils.clip_grad_norm_
    on the same values as 56-style separate tensors: below the threshold nothing changes (bit-equal), above it the
    scaled gradients agree to fp32 rounding of the coefficient."""
    from speech_anonymization_amd import ops
    torch.manual_seed(3)
    d = dev()
    sizes = [890_000 // 4, 620_000 // 4, 155_003]
    flats = [torch.randn(n, device=d) * scale * 1e-2 for n in sizes]
    ref_params = []
    for f in flats:
        off = 0
        for n in (7, 128 * 128 * 5, 64, f.numel()):
            n = min(n, f.numel() - off)
            if n <= 0:
                break
            p = torch.nn.Parameter(torch.zeros(n, device=d))
            p.grad = f[off:off + n].clone()
            ref_params.append(p)
            off += n
    tn_ref = torch.nn.utils.clip_grad_norm_(ref_params, 5.0)
    before = [f.clone() for f in flats]
    tn = ops.clip_flats(flats, 5.0)
    torch.cuda.synchronize()
    assert abs(float(tn) - float(tn_ref)) <= 2e-6 * float(tn_ref)
    got = torch.cat(flats)
    want = torch.cat([p.grad for p in ref_params])
    if float(tn_ref) <= 5.0:
        assert torch.equal(got, torch.cat(before))
    assert float((got - want).abs().max()) <= 3e-7 * float(want.abs().max())
