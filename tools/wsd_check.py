"""Fused data-gradient kernel (sa_conv_wsd.hip) against the one-tile kernel on the same inputs: y and the
bf16 d y cache must be bit-equal (same operand split, same accumulation order, same epilogue
operations); statistics / column sums equal up to the order of the in-tile sums.  Also times both.
  python tools/wsd_check.py [B] [--time]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import ctypes as C
import torch
from speech_anonymization_amd import _lib as L, ops

dev = torch.device("cuda:0")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if args else 6
TIME = "--time" in sys.argv
code = L.BF16X3


def case(name, K, dil, pad, Lin, nb, ep, g2=False, seed=3):
    """dgrad of Conv1d(128,128,K,dilation=dil,padding=pad): input rows Lin, output rows Lout"""
    g = torch.Generator().manual_seed(seed)
    Lout = Lin + dil * (K - 1) - 2 * pad
    x = torch.randn(B, Lin, 128, generator=g).to(dev)
    w = (torch.randn(128, 128, K, generator=g) * 0.05).to(dev)
    wd = ops.pack_weights(w, "conv_dgrad", torch.float32, code)
    taps = ops.taps_conv_dgrad_s1(K, dil, pad)
    kw = dict(code=code, want_stats=True)
    if nb:
        per_c = nb == "bn"
        shp = (128,) if per_c else (B, 128)
        c = [(torch.rand(*shp, generator=g) + 0.5).to(dev), (torch.randn(*shp, generator=g) * 0.1).to(dev),
             (torch.randn(*shp, generator=g) * 0.05).to(dev)]
        y2 = torch.randn(B, Lin, 128, generator=g).to(dev)
        kw["nb"] = dict(x=y2, c1=c[0], c2=c[1], c3=c[2], per_c=per_c, relu_mask=per_c, want_colsum=True)
    xe = torch.randn(B, Lout, 128, generator=g).to(dev)
    s1 = (torch.rand(B, 128, generator=g) + 0.5).to(dev)
    t1 = (torch.randn(B, 128, generator=g) * 0.1).to(dev)
    if ep == 1:
        mean, rstd = (torch.randn(B, 128, generator=g) * 0.1).to(dev), (torch.rand(B, 128, generator=g) + 0.5).to(dev)
        kw["ep"] = dict(mode=1, x=xe, s1=s1, t1=t1, mean=mean, rstd=rstd)
        if g2:
            kw["ep"]["g2"] = torch.randn(B, Lout, 128, generator=g).to(dev)
            kw["ep"]["g2k"] = [(torch.rand(128, generator=g) + 0.5).to(dev), (torch.randn(128, generator=g) * 0.1).to(dev),
                               (torch.randn(128, generator=g) * 0.05).to(dev)]
    else:
        mean, rstd = (torch.randn(128, generator=g) * 0.1).to(dev), (torch.rand(128, generator=g) + 0.5).to(dev)
        kw["ep"] = dict(mode=2, x=xe, mean=mean, rstd=rstd, per_c=True)
        if ep == 3:
            kw["ep"].update(s1=s1, t1=t1, xp_is_act=True)

    def run():
        ao = torch.full((B, Lin, 128), float("nan"), device=dev, dtype=torch.bfloat16) if nb else None
        out = ops.conv_gemm(x, wd, None, 128, 128, 1, 1, taps, Lout, a_out=ao, **kw)
        return tuple(out) + ((ao,) if ao is not None else ())

    a = L.SaConvArgs()
    return name, run, (Lin, Lout, taps, kw)


CASES = [
    ("enc11 (k5 p2, IN prologue, mode 1)", 5, 1, 2, 20160, "in", 1, False),
    ("tdnn0 (k5 p0, BN prologue, mode 2 + act)", 5, 1, 0, 20156, "bn", 3, False),
    ("dec0  (k5 p2, plain rows, mode 1 + g2k)", 5, 1, 2, 20160, None, 1, True),
    ("tdnn3 (k3 d2, BN prologue, mode 2)", 3, 2, 0, 20152, "bn", 2, False),
    ("tdnn6 (k3 d3, BN prologue, mode 2)", 3, 3, 0, 20146, "bn", 2, False),
    ("enc11 ragged (partial last tile)", 5, 1, 2, 20001, "in", 1, False),
    ("tdnn3 short utterances", 3, 2, 0, 5003, "bn", 2, False),
]
ok = True
for name, K, dil, pad, Lin, nb, ep, g2 in CASES:
    if "short" in name:
        B_save, B = B, 24
    nm, run, _ = case(name, K, dil, pad, Lin, nb, ep, g2)
    ops.conv_impl(ws=False)
    ref = run()
    ops.conv_impl(ws=True)
    got = run()
    torch.cuda.synchronize()
    names = ["y", "stats"] + (["colsum", "a_out"] if nb else [])
    line = []
    for nme, r, o in zip(names, ref, got):
        r, o = r.float(), o.float()
        if torch.isnan(o).any():
            line.append(f"{nme}: NaN!")
            ok = False
            continue
        d = (r - o).abs().max().item()
        rel = d / max(r.abs().max().item(), 1e-30)
        exact = torch.equal(r, o)
        line.append(f"{nme}: {'bit-equal' if exact else f'rel {rel:.1e}'}")
        if nme in ("y", "a_out") and not exact:
            ok = False
            if os.environ.get("WS_WHERE"):
                bad = ((r - o).abs() > 0).nonzero()
                rows = sorted(set((int(b_), int(l_)) for b_, l_, _ in bad[:200000].tolist()))
                cols = sorted(set(int(c_) for _, _, c_ in bad[:200000].tolist()))
                print("   bad rows (b, l):", rows[:24], "... total", len(rows), " frac", bad.shape[0] / r.numel())
                print("   bad cols:", cols[:24], "... total", len(cols))
        if nme in ("stats", "colsum") and rel > 2e-5:
            ok = False
    msg = ""
    if TIME:
        ts = []
        for ws in (False, True):
            ops.conv_impl(ws=ws)
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1000)
        msg = f"  [one-tile {ts[0]:.0f} us, wsd {ts[1]:.0f} us]"
    print(f"{name:44s} B={B:2d} " + "  ".join(line) + msg, flush=True)
    if "short" in name:
        B = B_save
ops.conv_impl()
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
