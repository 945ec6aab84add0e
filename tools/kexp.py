"""one-off experiments: conv c128 variants"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_anonymization_amd import ops
from tools.kbench import timeit
dev = torch.device("cuda:0")
B, L4 = int(__import__("os").environ.get("KB_B", "10")), 20160
from speech_anonymization_amd import _lib
for tm in (64,):
  _lib.load().sa_conv_gemm_set_tile_rows(tm)
  print("tile rows", tm)
  for prec in ("bf16", "bf16x3"):
      dt, code = ops.PRECISIONS[prec]
      cin = cout = 128
      x = torch.randn(B, L4, cin, device=dev).to(dt)
      w = torch.randn(cout, cin, 5, device=dev) * 0.05
      wp = ops.pack_weights(w, "conv_fwd", dt, code)
      s1 = torch.rand(B, cin, device=dev) + 0.5
      t1 = torch.randn(B, cin, device=dev) * 0.1
      y = torch.empty(B, L4, cout, device=dev, dtype=dt)
      ph = ops.taps_conv(5, 1, 2)
      for name, kw in (("full", dict(s1=s1, t1=t1, swish=True, want_stats=True)),
                       ("no swish", dict(s1=s1, t1=t1, want_stats=True)),
                       ("no prologue", dict(want_stats=True)),
                       ("no prologue no stats", dict()),
                       ("no pro/stats, shared weight slice", dict(relu=2)),
                       ("3 taps", dict(taps3=True))):
          p = ops.taps_conv(3, 2, 2) if kw.pop("taps3", False) else ph
          us = timeit(lambda: ops.conv_gemm(x, wp, None, cin, cout, 1, 1, p, L4, out=y, code=code, **kw))
          print(f"{prec:7s} {name:22s} {us:8.1f} us")
