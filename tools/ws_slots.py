"""Static instruction count of every filler slot of the compiled sa_conv_ws kernels (both the
interior and the masked edge variant of a slot are counted: an upper bound of what a tile executes).
  python tools/ws_slots.py"""
import os, re, subprocess, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "speech-anonymization_amd", "csrc", "sa_conv_ws.hip")
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "--offload-arch=gfx950", "-std=c++17",
                           "-save-temps=obj", "-c", src, "-o", os.path.join(d, "ws.o")], cwd=os.path.dirname(src))
    asm = open(os.path.join(d, "sa_conv_ws-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
for k in re.split(r"\n(?=_ZN12_GLOBAL__N_117sa_conv_ws_kernelILi\d)", asm)[1:]:
    name = k.split(":")[0][-34:]
    lines = k.split("s_endpgm")[0].split("\n")
    stm = [i for i, l in enumerate(lines) if "v_mfma_f32_32x32x16_bf16" in l]
    starts = [n for n, i in enumerate(stm) if lines[i].rstrip().endswith(", 0")][0::2] + [len(stm)]
    for seg in range(len(starts) - 1):
        body = stm[starts[seg]:starts[seg + 1]]
        gaps = []
        for a_, b_ in zip(body[:-1], body[1:]):
            n = 0
            for l in lines[a_ + 1:b_]:
                t = l.strip()
                if t and not t.startswith(";") and not t.startswith(".") and not t.endswith(":"):
                    n += 1
            gaps.append(n)
        print(f"{name} body {seg}: {len(body)} MFMAs, {sum(gaps)} instructions in the gaps between them ({8 * len(body)} issue "
              f"slots per tile; 7 per gap keep the matrix pipe fed), max {max(gaps)}, gaps over 7: {sum(g > 7 for g in gaps)}; "
              f"epilogue gaps avg {sum(gaps[0:64]) / 64:.1f}")
        print("  ", gaps)
