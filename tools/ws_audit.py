"""Audit of the compiled sa_conv_ws kernels (cdna_hip_programming.md 5.7 item 4: the MFMAs are inline
asm, so hipcc neither knows their latency nor pads hazards around them): no spills, no scratch, the
accumulators stay in ONE register block for the whole tile loop, and no compiler-generated
instruction touches that block between the first and the last MFMA statement of the loop.
  python tools/ws_audit.py        (exit 1 on a finding; runs on the CPU box, compiles only)"""
import os, re, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "speech-anonymization_amd", "csrc", "sa_conv_ws.hip")
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "--offload-arch=gfx950", "-std=c++17",
                           "-save-temps=obj", "-c", src, "-o", os.path.join(d, "ws.o")], cwd=os.path.dirname(src))
    asm = open(os.path.join(d, "sa_conv_ws-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
bad = 0
for key in ("vgpr_spill_count", "private_segment_fixed_size"):
    for v in re.findall(rf"\.{key}:\s+(\d+)", asm):
        if int(v):
            print(f"FINDING: .{key} {v}")
            bad += 1
kernels = re.split(r"\n(?=_ZN12_GLOBAL__N_117sa_conv_ws_kernelILi\d)", asm)[1:]
for k in kernels:
    name = k.split(":")[0]
    lines = k.split("s_endpgm")[0].split("\n")
    idx = [i for i, l in enumerate(lines) if "v_mfma_f32_32x32x16_bf16" in l]
    accs = sorted(set(re.search(r"bf16 (v\[\d+:\d+\])", lines[i]).group(1) for i in idx))
    print(f"{name[-22:]}: {len(idx)} MFMAs, accumulators {accs}")
    if len(accs) != 2:
        print("FINDING: accumulators migrate between register blocks")
        bad += 1
        continue
    regs = set()
    for a in accs:
        lo, hi = map(int, re.findall(r"\d+", a))
        regs |= set(range(lo, hi + 1))
    # one segment per unrolled tile body: it starts with the two C = 0 MFMAs (one per accumulator)
    starts = [n for n, i in enumerate(idx) if lines[i].rstrip().endswith(", 0")][0::2]
    if not starts or starts[0] != 0:
        print("FINDING: a tile body does not start with a C = 0 MFMA")
        bad += 1
    bounds = starts + [len(idx)]
    print(f"      tile bodies: {[bounds[n + 1] - bounds[n] for n in range(len(starts))]} MFMAs")
    for seg in range(len(starts)):
        first, last = idx[bounds[seg]], idx[bounds[seg + 1] - 1]
        in_asm = True                               # `first` sits inside an asm statement
        for i in range(first, last + 1):
            l = lines[i]
            if "#ASMSTART" in l:
                in_asm = True
            elif "#ASMEND" in l:
                in_asm = False
            elif not in_asm and not l.strip().startswith(";"):
                used = set()
                for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", l):
                    if m.group(3):
                        used.add(int(m.group(3)))
                    else:
                        used |= set(range(int(m.group(1)), int(m.group(2)) + 1))
                if used & regs:
                    print(f"FINDING: compiler instruction touches an accumulator inside the MFMA loop: {l.strip()}")
                    bad += 1
print("audit:", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)
