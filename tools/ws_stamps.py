"""Diagnostic build (-DSA_WS_STAMPS): where wave 0 of one workgroup of sa_conv_ws spends each tile
(s_memtime ticks = shader cycles; the printed ns per tick is the clock the chip held under the kernel).  python tools/ws_stamps.py [plain|fwd|nb]"""
import sys, os, ctypes as C, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
src = os.path.join(R, "speech-anonymization_amd", "csrc")
abl = int(os.environ.get("WS_ABL", "0"))
so = os.path.join(R, "build", "abl", f"libsa_ws_stamps_{abl}.so")
if not os.path.exists(so) or os.environ.get("WS_REBUILD"):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(f"cd {src} && /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -DSA_WS_STAMPS -DSA_ABL={abl} -shared -o {so} sa_conv_gemm.hip sa_conv_pp.hip sa_conv_ws.hip sa_wgrad.hip sa_small.hip sa_elementwise.hip sa_head.hip sa_fbank.hip sa_mi.hip", shell=True)
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.exit(0)
import numpy as np
import torch
from speech_anonymization_amd import _lib, ops
_lib.LIB_PATH = so
lib = _lib.load()
L = _lib
ops.conv_impl(ws=True)
dev = torch.device("cuda:0")
B, L4 = int(os.environ.get("KB_B", "32")), 20160
code = L.BF16X3
x = torch.randn(B, L4, 128, device=dev)
y2 = torch.randn(B, L4, 128, device=dev)
w = torch.randn(128, 128, 5, device=dev) * 0.05
wf = ops.pack_weights(w, "conv_fwd", torch.float32, code)
wd = ops.pack_weights(w, "conv_dgrad", torch.float32, code)
s1 = torch.rand(B, 128, device=dev) + 0.5
y = torch.empty(B, L4, 128, device=dev)
a_out = torch.empty(B, L4, 128, device=dev, dtype=torch.bfloat16)
c = [torch.rand(B, 128, device=dev) + 0.5 for _ in range(3)]
which = sys.argv[1] if len(sys.argv) > 1 else "plain"


def run():
    if which == "plain":
        ops.conv_gemm(x, wf, None, 128, 128, 1, 1, ops.taps_conv(5, 1, 2), L4, out=y, code=code)
    elif which == "fwd":
        ops.conv_gemm(x, wf, None, 128, 128, 1, 1, ops.taps_conv(5, 1, 2), L4, s1=s1, t1=s1, swish=True,
                      want_stats=True, out=y, code=code, a_out=a_out)
    else:
        ops.conv_gemm(x, wd, None, 128, 128, 1, 1, ops.taps_conv_dgrad_s1(5, 1, 2), L4, want_stats=True,
                      out=y, code=code, a_out=a_out,
                      nb=dict(x=y2, c1=c[0], c2=c[1], c3=c[2], per_c=False, relu_mask=False, want_colsum=True),
                      **({"ep": dict(mode=1, x=y2, s1=s1, t1=s1, mean=s1, rstd=s1)} if which == "dgrad" else {}))


for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
buf = (C.c_ulonglong * (64 * 8))()
lib.sa_ws_dbg_read(buf)
a = np.array(list(buf), dtype=np.float64).reshape(64, 8)
n = int((a[:, 0] > 0).sum())
span = a[n - 1, 0] - a[0, 0]
print(f"ABL={abl} {which}: {us:.1f} us per launch; workgroup 7: {n - 1} tiles in {span:.0f} ticks "
      f"=> {span / (n - 1):.0f} ticks per tile, {us * 1e3 / span:.3f} ns per tick if the workgroup spans the launch")
print("per tile (ticks): epilogue slots 0..33 | empty slots up to FT-1 | counted vmcnt wait | transform + DMA slots FT.. | wait states, accumulator copy, barrier, scalar set-up | total")
for it in range(min(n - 1, int(os.environ.get("WS_ROWS", "12")))):
    r = a[it]
    nxt = a[it + 1, 0]
    print(f"it={it:2d}  {r[3]-r[0]:7.0f} {r[4]-r[3]:7.0f} {r[5]-r[4]:7.0f} {r[1]-r[5]:7.0f} {nxt-r[1]:7.0f}   {nxt-r[0]:7.0f}")
