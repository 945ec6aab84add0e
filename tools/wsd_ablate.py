"""Timing-only ablation of the fused data-gradient kernel (sa_conv_wsd.hip, -DSA_WSD_ABL=<mask>: what each
stream of the tile body costs at B = 32; WRONG results in every build but mask 0).
  python tools/wsd_ablate.py build          (CPU box: compiles the variants into build/abl/)
  python tools/wsd_ablate.py [case ...]     (GPU box: times them, HIP events, min of 3 rounds of 10 launches)"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
src = os.path.join(R, "speech-anonymization_amd", "csrc")
MASKS = [0, 1, 2, 3, 4, 8, 16, 32, 7]
NAMES = {0: "shipped", 1: "no epilogue stream", 2: "no transform stream", 3: "MFMA + A reads only", 4: "no MFMA",
         8: "no counted waits", 16: "no epilogue loads / waits", 32: "no epilogue stores", 7: "empty body"}


def so_of(m):
    return os.path.join(R, "build", "abl", f"libsa_wsd_abl_{m}.so")


if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(os.path.join(R, "build", "abl"), exist_ok=True)
    subprocess.check_call(["make", "-C", src, "-j8"], stdout=subprocess.DEVNULL)
    objs = [o for o in os.listdir(src) if o.endswith(".o") and o != "sa_conv_wsd.o"]
    procs = []
    for m in MASKS:
        o = os.path.join(R, "build", "abl", f"wsd_{m}.o")
        procs.append((m, o, subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "--offload-arch=gfx950", "-std=c++17",
                                               f"-DSA_WSD_ABL={m}", "-c", os.path.join(src, "sa_conv_wsd.hip"), "-o", o])))
        if len(procs) % 4 == 0:
            for _, _, p_ in procs[-4:]:
                p_.wait()
    for m, o, p_ in procs:
        assert p_.wait() == 0
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so_of(m), o] +
                              [os.path.join(src, x) for x in objs] + ["-ldl"])
    print("built", [so_of(m) for m in MASKS])
    sys.exit(0)

import torch
dev = torch.device("cuda:0")
B = int(os.environ.get("KB_B", "32"))
want = [a for a in sys.argv[1:]] or ["enc11", "tdnn0", "dec0", "tdnn3"]
CASES = {"enc11": (5, 1, 2, 20160, "in", 1, False), "tdnn0": (5, 1, 0, 20156, "bn", 3, False),
         "dec0": (5, 1, 2, 20160, None, 1, True), "tdnn3": (3, 2, 0, 20152, "bn", 2, False),
         "tdnn6": (3, 3, 0, 20146, "bn", 2, False)}
# one child process per library variant (a process binds one libsa_hip.so)
if os.environ.get("WSD_ABL_CHILD") is None:
    print(f"B = {B}; us per launch (min of 3 rounds x 10 launches)")
    print(f"{'build':28s}" + "".join(f"{c:>9s}" for c in want))
    for m in MASKS:
        env = dict(os.environ, WSD_ABL_CHILD=str(m))
        out = subprocess.run([sys.executable, __file__] + want, env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        print(f"{NAMES[m]:28s}" + (line[0][6:] if line else "  failed: " + out.stderr[-300:]), flush=True)
    sys.exit(0)
m = int(os.environ["WSD_ABL_CHILD"])
from speech_anonymization_amd import _lib
_lib.LIB_PATH = so_of(m)
from speech_anonymization_amd import _lib as L, ops
ops.conv_impl(ws=True)
res = []
for cname in want:
    K, dil, pad, Lin, nb, ep, g2 = CASES[cname]
    g = torch.Generator().manual_seed(3)
    Lout = Lin + dil * (K - 1) - 2 * pad
    x = torch.randn(B, Lin, 128, generator=g).to(dev)
    w = (torch.randn(128, 128, K, generator=g) * 0.05).to(dev)
    wd = ops.pack_weights(w, "conv_dgrad", torch.float32, L.BF16X3)
    taps = ops.taps_conv_dgrad_s1(K, dil, pad)
    kw = dict(code=L.BF16X3, want_stats=True)
    ao = None
    if nb:
        per_c = nb == "bn"
        shp = (128,) if per_c else (B, 128)
        c = [(torch.rand(*shp, generator=g) + 0.5).to(dev) for _ in range(3)]
        y2 = torch.randn(B, Lin, 128, generator=g).to(dev)
        kw["nb"] = dict(x=y2, c1=c[0], c2=c[1], c3=c[2], per_c=per_c, relu_mask=per_c, want_colsum=True)
        ao = torch.empty(B, Lin, 128, device=dev, dtype=torch.bfloat16)
    xe = torch.randn(B, Lout, 128, generator=g).to(dev)
    s1 = (torch.rand(B, 128, generator=g) + 0.5).to(dev)
    if ep == 1:
        kw["ep"] = dict(mode=1, x=xe, s1=s1, t1=s1, mean=s1, rstd=s1)
        if g2:
            kw["ep"]["g2"] = torch.randn(B, Lout, 128, generator=g).to(dev)
            kw["ep"]["g2k"] = [(torch.rand(128, generator=g) + 0.5).to(dev) for _ in range(3)]
    else:
        mr = (torch.rand(128, generator=g) + 0.5).to(dev)
        kw["ep"] = dict(mode=2, x=xe, mean=mr, rstd=mr, per_c=True)
        if ep == 3:
            kw["ep"].update(s1=s1, t1=s1, xp_is_act=True)
    y = torch.empty(B, Lout, 128, device=dev)

    def run():
        ops.conv_gemm(x, wd, None, 128, 128, 1, 1, taps, Lout, a_out=ao, out=y, **kw)
    for _ in range(3):
        run()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1000)
    res.append(best)
print("RESULT" + "".join(f"{r:9.1f}" for r in res))
