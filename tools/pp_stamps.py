"""Diagnostic build (-DSA_PP_STAMPS): where one workgroup of sa_conv_pp spends its half-steps
(shader cycles; wave 0 of each group).  python tools/pp_stamps.py [fwd|dgrad]"""
import sys, os, ctypes as C, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
src = os.path.join(R, "speech-anonymization_amd", "csrc")
abl = int(os.environ.get("PP_ABL", "0"))
so = os.path.join(R, "build", "abl", f"libsa_pp_stamps_{abl}.so")
if not os.path.exists(so) or os.environ.get("PP_REBUILD"):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(f"cd {src} && /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -DSA_PP_STAMPS -DSA_ABL={abl} -shared -o {so} sa_conv_gemm.hip sa_conv_pp.hip sa_wgrad.hip sa_small.hip sa_elementwise.hip sa_head.hip sa_fbank.hip sa_mi.hip", shell=True)
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.exit(0)
import numpy as np
import torch
from speech_anonymization_amd import _lib, ops
_lib.LIB_PATH = so
lib = _lib.load()
L = _lib
if os.environ.get("SA_CONV_IMPL"):
    impl = os.environ["SA_CONV_IMPL"]
    ops.conv_impl(pingpong=impl != "old", pp_rows=64 if impl == "pp64" else 0)
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
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"


def run():
    if which == "fwd":
        ops.conv_gemm(x, wf, None, 128, 128, 1, 1, ops.taps_conv(5, 1, 2), L4, s1=s1, t1=s1, swish=True,
                      want_stats=True, out=y, code=code, a_out=a_out)
    else:
        ops.conv_gemm(x, wd, None, 128, 128, 1, 1, ops.taps_conv_dgrad_s1(5, 1, 2), L4, want_stats=True,
                      out=y, code=code, a_out=a_out,
                      nb=dict(x=y2, c1=c[0], c2=c[1], c3=c[2], per_c=False, relu_mask=False, want_colsum=True),
                      ep=dict(mode=1, x=y2, s1=s1, t1=s1, mean=s1, rstd=s1))


for _ in range(5):
    run()
torch.cuda.synchronize()
buf = (C.c_ulonglong * (2 * 64 * 8))()
lib.sa_pp_dbg_read(buf)
a = np.array(list(buf), dtype=np.float64).reshape(2, 64, 8)
t0 = a[0, 0, 0]
print(which, "half-steps of workgroup 7 (cycles since its start): start | epi done | stage done | (loads landed) | mfma done | after barrier")
import time
torch.cuda.synchronize(); t_0 = time.perf_counter()
for _ in range(20):
    run()
torch.cuda.synchronize()
print(f"ABL={abl} {which}: {(time.perf_counter() - t_0) / 20 * 1e6:.1f} us per launch")
for h in range(int(os.environ.get("PP_ROWS", "8"))):
    for g in range(2):
        r = a[g, h]
        if r[0] == 0:
            continue
        f = lambda v: f"{v - t0:9.0f}" if v else "        -"
        print(f"h={h:2d} grp {g}: start {f(r[0])}  epi {f(r[1])}  loads {f(r[5])}  staged {f(r[2])}  mfma {f(r[3])}  barrier {f(r[4])}   half-step {r[4]-r[0]:7.0f}")
