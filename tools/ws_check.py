"""Weight-stationary conv kernel against the one-tile kernel on the same inputs (bit-equal outputs
expected: same operand split, same accumulation order).  "nb" / "dgrad" run on the one-tile kernel in
both arms unless the library was built with -DSA_WS_PRO2.  python tools/ws_check.py [B] [L]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from speech_anonymization_amd import _lib as L, ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6          # >= 1536 tiles, or both arms run the one-tile kernel
Ln = int(sys.argv[2]) if len(sys.argv) > 2 else 20160
g = torch.Generator().manual_seed(3)
x = torch.randn(B, Ln, 128, generator=g).to(dev)
y2 = torch.randn(B, Ln, 128, generator=g).to(dev)
w = (torch.randn(128, 128, 5, generator=g) * 0.05).to(dev)
bias = torch.randn(128, generator=g).to(dev)
s1 = (torch.rand(B, 128, generator=g) + 0.5).to(dev)
t1 = (torch.randn(B, 128, generator=g) * 0.1).to(dev)
c = [(torch.rand(B, 128, generator=g) + 0.5).to(dev) for _ in range(3)]
code = L.BF16X3
wf = ops.pack_weights(w, "conv_fwd", torch.float32, code)
wd = ops.pack_weights(w, "conv_dgrad", torch.float32, code)


def run(kind):
    ao = torch.zeros(B, Ln, 128, device=dev, dtype=torch.bfloat16)
    if kind == "plain":
        return (ops.conv_gemm(x, wf, bias, 128, 128, 1, 1, ops.taps_conv(5, 1, 2), Ln, code=code),)
    if kind == "fwd":
        y, st = ops.conv_gemm(x, wf, bias, 128, 128, 1, 1, ops.taps_conv(5, 1, 2), Ln, s1=s1, t1=t1, swish=True,
                              want_stats=True, code=code, a_out=ao)
        return y, st, ao
    if kind == "nb":
        y, st, cs = ops.conv_gemm(x, wd, None, 128, 128, 1, 1, ops.taps_conv_dgrad_s1(5, 1, 2), Ln, want_stats=True,
                                  code=code, a_out=ao,
                                  nb=dict(x=y2, c1=c[0], c2=c[1], c3=c[2], per_c=False, relu_mask=False, want_colsum=True))
        return y, st, ao, cs
    if kind == "dgrad":
        y, st, cs = ops.conv_gemm(x, wd, None, 128, 128, 1, 1, ops.taps_conv_dgrad_s1(5, 1, 2), Ln, want_stats=True,
                                  code=code, a_out=ao,
                                  nb=dict(x=y2, c1=c[0], c2=c[1], c3=c[2], per_c=False, relu_mask=True, want_colsum=True),
                                  ep=dict(mode=1, x=y2, s1=s1, t1=t1, mean=t1, rstd=s1))
        return y, st, ao, cs


ok = True
for kind in ("plain", "fwd", "nb", "dgrad"):
    ops.conv_impl(ws=False)
    ref = run(kind)
    ops.conv_impl(ws=True)
    got = run(kind)
    torch.cuda.synchronize()
    for i, (r, o) in enumerate(zip(ref, got)):
        r, o = r.float(), o.float()
        d = (r - o).abs().max().item()
        rel = d / max(r.abs().max().item(), 1e-30)
        exact = torch.equal(r, o)
        print(f"{kind:6s} out{i} shape {tuple(r.shape)} max|d| {d:.3e} rel {rel:.2e} {'bit-equal' if exact else ''}", flush=True)
        if rel > 1e-5:
            ok = False
        if not exact and r.dim() == 3 and r.shape[1] == Ln and os.environ.get("WS_WHERE"):
            if True:
                bad = ((r - o).abs() > 0).nonzero()
                rows = sorted(set((int(b_), int(l_)) for b_, l_, _ in bad.tolist()))
                cols = sorted(set(int(c_) for _, _, c_ in bad.tolist()))
                print("   bad rows (b, l):", rows[:40], "... total", len(rows))
                print("   bad cols:", cols[:40], "... total", len(cols))
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
