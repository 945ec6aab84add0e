"""Kernel micro-benchmarks at the shape-M sizes (interleaved rounds, HIP-event timing).
python tools/kbench.py [conv|wgrad|ew|all] [precisions, comma separated]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_anonymization_amd import ops

dev = torch.device("cuda:0")
if os.environ.get("KB_TILE_ROWS"):                     # conv tile rows knob (64 | 128)
    from speech_anonymization_amd import _lib as _L
    assert _L.load().sa_conv_gemm_set_tile_rows(int(os.environ["KB_TILE_ROWS"])) == 0
if os.environ.get("KB_WG_TARGET"):                     # "big,small" workgroup-count targets of ops.wgrad
    _b, _s = os.environ["KB_WG_TARGET"].split(",")
    ops.WGRAD_TARGET_WGS.update({True: int(_b), False: int(_s)})
what = (sys.argv[1] if len(sys.argv) > 1 else "all") if __name__ == "__main__" else "none"
precs = (sys.argv[2] if len(sys.argv) > 2 else "bf16x3,bf16,f32").split(",")
B, L4 = int(__import__("os").environ.get("KB_B", "10")), 20160


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3          # us


CONV = [  # name, cin, cout, sa, u, phases, Lin, Lout
    ("c128 k5", 128, 128, 1, 1, ops.taps_conv(5, 1, 2), L4, L4),
    ("c64 k5", 64, 64, 1, 1, ops.taps_conv(5, 1, 2), 2 * L4, 2 * L4),
    ("c64->128 s2", 64, 128, 2, 1, ops.taps_conv(5, 1, 2), 2 * L4, L4),
    ("c32->64 s2", 32, 64, 2, 1, ops.taps_conv(5, 1, 2), 4 * L4, 2 * L4),
    ("cT128->64", 128, 64, 1, 2, ops.UP2, L4, 2 * L4),
    ("cT64->32", 64, 32, 1, 2, ops.UP2, 2 * L4, 4 * L4),
]
for prec in precs:
    dt, code = ops.PRECISIONS[prec]
    if what in ("conv", "all"):
        for name, cin, cout, sa, u, ph, Lin, Lout in CONV:
            x = torch.randn(B, Lin, cin, device=dev).to(dt)
            w = torch.randn(cout, cin, 5, device=dev) * 0.05
            wp = ops.pack_weights(w, "conv_fwd", dt, code)
            s1 = torch.rand(B, cin, device=dev) + 0.5
            t1 = torch.randn(B, cin, device=dev) * 0.1
            bias = torch.randn(cout, device=dev)
            y = torch.empty(B, Lout, cout, device=dev, dtype=dt)
            f = lambda: ops.conv_gemm(x, wp, bias, cin, cout, sa, u, ph, Lout, s1=s1, t1=t1, swish=True,
                                      want_stats=True, out=y, code=code)
            us = timeit(f)
            ntap = sum(len(p) for p in ph)
            flops = 2 * B * (Lout // u) * ntap * cin * cout
            byts = (x.numel() + y.numel()) * x.element_size()
            print(f"{prec:7s} conv  {name:12s} {us:8.1f} us  {flops/us/1e6:7.1f} TF  {byts/us/1e3:7.1f} GB/s")
    if what in ("wgrad", "all"):
        for name, cin, cout, sa, u, ph, Lin, Lout in CONV:
            x = torch.randn(B, Lin, cin, device=dev).to(dt)
            dy = torch.randn(B, Lout, cout, device=dev).to(dt)
            s1 = torch.rand(B, cin, device=dev) + 0.5
            t1 = torch.randn(B, cin, device=dev) * 0.1
            if u == 2:
                taps, Mrows, dst, strides = [(1, 0), (1, 1), (0, 0), (0, 1), (-1, 0)], Lin, torch.empty(cin, cout, 5, device=dev), (cout * 5, 5, 1)
            else:
                taps, Mrows, dst, strides = [(k - 2, 0) for k in range(5)], Lout, torch.empty(cout, cin, 5, device=dev), (5, cin * 5, 1)
            if os.environ.get("KB_WG_RAW"):                 # no prologue transform (VALU share probe)
                f = lambda: ops.wgrad(x, dy, cin, cout, sa, u, taps, Mrows, dst, strides,
                                      code=ops.WGRAD_CODE[prec])
            else:
                f = lambda: ops.wgrad(x, dy, cin, cout, sa, u, taps, Mrows, dst, strides, s1=s1, t1=t1,
                                      swish=True, code=ops.WGRAD_CODE[prec])
            us = timeit(f)
            flops = 2 * B * Mrows * 5 * cin * cout
            byts = (x.numel() + dy.numel()) * x.element_size()
            print(f"{prec:7s} wgrad {name:12s} {us:8.1f} us  {flops/us/1e6:7.1f} TF  {byts/us/1e3:7.1f} GB/s (incl. reduce)")
    if what in ("ew", "all") and prec != "bf16x3":
        for Cc, Ln in ((128, L4), (64, 2 * L4), (32, 4 * L4)):
            g = torch.randn(B, Ln, Cc, device=dev).to(dt)
            x = torch.randn(B, Ln, Cc, device=dev).to(dt)
            o = torch.empty_like(g)
            v = torch.rand(B, Cc, device=dev) + 0.5
            f1 = lambda: ops.ew("stats", g, x, Cc, out=o, s1=v, t1=v, mean=v, rstd=v, actbwd=True)
            f2 = lambda: ops.ew("apply", g, x, Cc, out=o, c1=v, c2=v, c3=v)
            for nm, f in (("stats", f1), ("apply", f2)):
                us = timeit(f)
                byts = 3 * g.numel() * g.element_size()
                print(f"{prec:7s} ew {nm} C{Cc:<4d} {us:8.1f} us  {byts/us/1e3:7.1f} GB/s")
