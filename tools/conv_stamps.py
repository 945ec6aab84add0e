"""Diagnostic build (-DSA_CONV_STAMPS): phase durations of sa_conv_gemm workgroups in shader cycles."""
import sys, os, ctypes as C, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
src = os.path.join(R, "speech-anonymization_amd", "csrc")
so = "/tmp/libsa_stamps.so"
subprocess.check_call(f"cd {src} && /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -DSA_CONV_STAMPS -shared -o {so} sa_conv_gemm.hip sa_wgrad.hip sa_small.hip sa_elementwise.hip sa_head.hip sa_fbank.hip sa_mi.hip", shell=True)
import torch
from speech_anonymization_amd import _lib, ops
_lib.LIB_PATH = so
lib = _lib.load()
dev = torch.device("cuda:0")
B, L4 = int(os.environ.get("KB_B", "32")), 20160
for prec in ("bf16x3", "bf16"):
    dt, code = ops.PRECISIONS[prec]
    x = torch.randn(B, L4, 128, device=dev).to(dt)
    w = torch.randn(128, 128, 5, device=dev) * 0.05
    wp = ops.pack_weights(w, "conv_fwd", dt, code)
    s1 = torch.rand(B, 128, device=dev) + 0.5
    y = torch.empty(B, L4, 128, device=dev, dtype=dt)
    for _ in range(3):
        ops.conv_gemm(x, wp, None, 128, 128, 1, 1, ops.taps_conv(5, 1, 2), L4, s1=s1, t1=s1, swish=True, want_stats=True, out=y, code=code)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 512)()
    lib.sa_conv_dbg_read(buf)
    import numpy as np
    full = np.array(list(buf), dtype=np.float64).reshape(64, 8)
    full = full[full[:, 0] > 0]
    a = full[:, :7]
    d = np.diff(a, axis=1)
    names = ["row loads land", "xform + LDS write + barrier", "MFMA loop", "barrier (As free)", "acc->LDS + barrier", "store + stats"]
    print(prec, "workgroups sampled:", len(a))
    for n, v in zip(names, np.median(d, axis=0)):
        print(f"   {n:26s} {v:9.0f} cycles")
    print(f"   {'total':26s} {np.median(a[:,6]-a[:,0]):9.0f} cycles")
    # clock: run once more recording (memtime, memrealtime) deltas between two launches
    t0 = full[:, [6, 7]].copy()
    for _ in range(50):
        ops.conv_gemm(x, wp, None, 128, 128, 1, 1, ops.taps_conv(5, 1, 2), L4, s1=s1, t1=s1, swish=True, want_stats=True, out=y, code=code)
    torch.cuda.synchronize()
    lib.sa_conv_dbg_read(buf)
    f2 = np.array(list(buf), dtype=np.float64).reshape(64, 8)
    f2 = f2[f2[:, 0] > 0]
    n = min(len(t0), len(f2))
    dc, dr = f2[:n, 6] - t0[:n, 0], f2[:n, 7] - t0[:n, 1]
    print(f"   shader clock over 50 back-to-back launches: {np.median(dc / dr) * 100:.0f} MHz")
