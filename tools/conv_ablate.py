"""Timing of sa_conv_gemm<bf16x3,128,128,1,1> as the train step launches it (forward with the
operand cache and statistics; data gradient with the normalisation-backward prologue and the fused
backward epilogue) for one build of the library.

  python tools/conv_ablate.py                       # the in-tree library
  SA_HIP_LIB=build/abl/libsa_abl_4.so python tools/conv_ablate.py   # an ablation / experiment build
  python tools/conv_ablate.py --all build/abl       # every lib*.so in a directory + the in-tree one,
                                                    # one child process each (interleaved rounds)
Ablation builds (-DSA_ABL=mask, see sa_conv_gemm.hip) compute WRONG results: timing only."""
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)


def one():
    import torch
    from speech_anonymization_amd import _lib as L, ops
    dev = torch.device("cuda:0")
    if os.environ.get("SA_CONV_IMPL"):          # "old" | "old128" | "pp128" | "pp64" | "ws"
        impl = os.environ["SA_CONV_IMPL"]
        ops.conv_impl(pingpong=impl.startswith("pp"), pp_rows=64 if impl == "pp64" else 0,
                      tile_rows=128 if impl == "old128" else 0, ws=impl == "ws")
    else:
        ops.conv_impl(ws=False)
    B, L4 = int(os.environ.get("KB_B", "32")), 20160
    code = L.BF16X3
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(B, L4, 128, generator=g).to(dev)
    y2 = torch.randn(B, L4, 128, generator=g).to(dev)
    w = (torch.randn(128, 128, 5, generator=g) * 0.05).to(dev)
    wf = ops.pack_weights(w, "conv_fwd", torch.float32, code)
    wd = ops.pack_weights(w, "conv_dgrad", torch.float32, code)
    s1 = (torch.rand(B, 128, generator=g) + 0.5).to(dev)
    t1 = (torch.randn(B, 128, generator=g) * 0.1).to(dev)
    bias = torch.randn(128, generator=g).to(dev)
    y = torch.empty(B, L4, 128, device=dev)
    a_out = torch.empty(B, L4, 128, device=dev, dtype=torch.bfloat16)
    c = [(torch.rand(B, 128, generator=g) + 0.5).to(dev) for _ in range(3)]
    mean, rstd = t1, s1

    def fwd():
        ops.conv_gemm(x, wf, bias, 128, 128, 1, 1, ops.taps_conv(5, 1, 2), L4, s1=s1, t1=t1, swish=True,
                      want_stats=True, out=y, code=code, a_out=a_out)

    def fwd_plain():
        ops.conv_gemm(x, wf, bias, 128, 128, 1, 1, ops.taps_conv(5, 1, 2), L4, out=y, code=code)

    def dgrad():
        ops.conv_gemm(x, wd, None, 128, 128, 1, 1, ops.taps_conv_dgrad_s1(5, 1, 2), L4, want_stats=True,
                      out=y, code=code, a_out=a_out,
                      nb=dict(x=y2, c1=c[0], c2=c[1], c3=c[2], per_c=False, relu_mask=False, want_colsum=True),
                      ep=dict(mode=1, x=y2, s1=s1, t1=t1, mean=mean, rstd=rstd))

    def timeit(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    res = {}
    for rnd in range(3):
        for name, fn in (("fwd", fwd), ("plain", fwd_plain), ("dgrad", dgrad)):
            res.setdefault(name, []).append(timeit(fn))
    tag = os.path.basename(os.environ.get("SA_HIP_LIB", "in-tree")) + " " + os.environ.get("SA_CONV_IMPL", "")
    print(f"{tag:28s} " + "  ".join(f"{k} {min(v):7.1f} us (med {sorted(v)[1]:7.1f})" for k, v in res.items()),
          flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--all":
        libs = [None] + sorted(os.path.join(sys.argv[2], f) for f in os.listdir(sys.argv[2]) if f.endswith(".so"))
        for lib in libs:
            env = dict(os.environ)
            if lib:
                env["SA_HIP_LIB"] = os.path.abspath(lib)
            else:
                env.pop("SA_HIP_LIB", None)
            rc = subprocess.call([sys.executable, os.path.abspath(__file__)], env=env)
            if rc != 0:
                print(f"{lib}: exit {rc}", flush=True)
    else:
        one()
