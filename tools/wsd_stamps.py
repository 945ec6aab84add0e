"""Diagnostic build (-DSA_WSD_STAMPS): where wave 0 of one workgroup of sa_conv_wsd spends each tile
(s_memtime ticks = shader cycles; s_memrealtime at 100 MHz gives the clock the chip held).
  python tools/wsd_stamps.py build | python tools/wsd_stamps.py [enc11|tdnn0|dec0|tdnn3]"""
import os, subprocess, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
src = os.path.join(R, "speech-anonymization_amd", "csrc")
so = os.path.join(R, "build", "abl", "libsa_wsd_stamps.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["make", "-C", src, "-j8"], stdout=subprocess.DEVNULL)
    o = os.path.join(R, "build", "abl", "wsd_stamps.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "--offload-arch=gfx950", "-std=c++17", "-DSA_WSD_STAMPS",
                           "-c", os.path.join(src, "sa_conv_wsd.hip"), "-o", o])
    objs = [os.path.join(src, x) for x in os.listdir(src) if x.endswith(".o") and x != "sa_conv_wsd.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, o] + objs + ["-ldl"])
    sys.exit(0)
import torch
from speech_anonymization_amd import _lib
_lib.LIB_PATH = so
from speech_anonymization_amd import _lib as L, ops
lib = L.load()
ops.conv_impl(ws=True)
if os.environ.get("KB_PQ"):                                   # cost of a plain iteration in quarter tiles (default 9)
    assert lib.sa_conv_wsd_set_bcost(int(os.environ["KB_PQ"])) == 0
dev = torch.device("cuda:0")
B = int(os.environ.get("KB_B", "32"))
CASES = {"enc11": (5, 1, 2, 20160, "in", 1, False), "tdnn0": (5, 1, 0, 20156, "bn", 3, False),
         "dec0": (5, 1, 2, 20160, None, 1, True), "tdnn3": (3, 2, 0, 20152, "bn", 2, False)}
for cname in (sys.argv[1:] or ["enc11", "tdnn3"]):
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
    run = lambda: ops.conv_gemm(x, wd, None, 128, 128, 1, 1, taps, Lout, a_out=ao, out=y, **kw)
    for _ in range(30):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (64 * 16))()
    lib.sa_wsd_dbg_read(buf)
    st = [[buf[i * 16 + j] for j in range(16)] for i in range(64)]
    print(f"== {cname}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch (stamped build)")
    wg = (C.c_ulonglong * 1024)()
    lib.sa_wsd_wg_read(wg)
    ent = [wg[2 * i] for i in range(512) if wg[2 * i]]
    ext = [wg[2 * i + 1] for i in range(512) if wg[2 * i]]
    if ent:
        t0w = min(ent)
        life = sorted((b - a) * 0.01 for a, b in zip(ent, ext))
        print(f"   {len(ent)} workgroups: entries spread over {(max(ent) - t0w) * 0.01:.1f} us, exits from {(min(ext) - t0w) * 0.01:.1f} to "
              f"{(max(ext) - t0w) * 0.01:.1f} us after the first entry; lifetimes min / median / max {life[0]:.1f} / {life[len(life) // 2]:.1f} / {life[-1]:.1f} us")
    if ent and os.environ.get("KB_XCD"):
        lt = [(wg[2 * i + 1] - wg[2 * i]) * 0.01 for i in range(256)]
        med = lambda v: sorted(v)[len(v) // 2]
        print("   median lifetime by workgroup index mod 8 (XCD):", " ".join(f"{med(lt[x::8]):.1f}" for x in range(8)))
        slow = sorted(range(256), key=lambda i: -lt[i])[:12]
        print("   slowest workgroups:", " ".join(f"{i}:{lt[i]:.0f}" for i in slow))
    pro = st[63]
    st[63] = [0] * 16
    if pro[0]:
        print(f"   workgroup 7: entry -> weights in registers {pro[1] - pro[0]} cycles, -> first tile staged {pro[2] - pro[1]}, "
              f"tile walk + tail {pro[3] - pro[2]}; entry..exit {(pro[9] - pro[8]) * 0.01:.1f} us of the launch's time")
    n = max(i for i in range(64) if st[i][0]) + 1
    rt0, rtn = st[0][5], st[n - 1][5]
    cyc = st[n - 1][0] - st[0][0]
    ghz = cyc / ((rtn - rt0) * 10.0) if rtn > rt0 else 0
    print(f"   {n} iterations, {cyc} cycles in {(rtn - rt0) * 0.01:.1f} us -> {ghz:.2f} GHz")
    print("   it  sec0  [E..SB]  sec1  [E..SB]  tail  barrier  total   (cycles; fast iterations have both sections stamped; top: barrier..loop top, A reads + flags, pointers, ..body)")
    for i in range(n):
        s0, s1_, s2, s3, s4 = st[i][0], st[i][1], st[i][2], st[i][3], st[i][4]
        nxt = st[i + 1][0] if i + 1 < n else s4
        if s1_ and s2:
            top = f"  top: {st[i][8] - st[i - 1][4] if i else 0:5d} {st[i][9] - st[i][8]:5d} {st[i][10] - st[i][9]:5d} {s0 - st[i][10]:5d}"
            print(f"   {i:2d} {s1_ - s0:6d} {st[i][6] - s0:7d} {s2 - s1_:6d} {st[i][7] - s1_:7d} {s3 - s2:6d} {s4 - s3:6d} {nxt - s0:7d}" + top)
        else:
            print(f"   {i:2d}   (plain iteration) body..tail {s3 - s0:7d} barrier {s4 - s3:6d} total {nxt - s0:7d}")
