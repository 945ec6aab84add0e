"""Audit of the compiled sa_conv_wsd kernels (cdna_hip_programming.md 5.7: MFMAs, loads, stores and
waits of the tile body are inline asm, so hipcc neither knows their latency nor counts them):
  1. no scratch access between the first and the last MFMA of a kernel (a lane constant parked in
     scratch at kernel entry and fetched back in the tail is tolerated and reported);
  2. every section of a tile body accumulates into ONE register block, the two sections of a body into
     two different blocks (the overlapped and the plain copy of the body may use different pairs: the
     compiler then moves a live accumulator between them at the loop top, behind the tile barrier);
  3. while a section accumulates (from its C = 0 MFMA to its last MFMA) no compiler-generated
     instruction touches its block, and none READS it sooner than three MFMA statements after the
     last MFMA that wrote it (outside that window the block is an ordinary value: once its epilogue
     has read a register the compiler may reuse it);
  4. the reserved registers v240..v255 / a240..a255 (values loaded across the tile loop's back edge,
     hand-placed weights) appear in no compiler-generated instruction, and the kernel descriptor
     covers them (accum_offset 256, 256 accumulator registers);
  5. no compiler-inserted s_waitcnt vmcnt inside the overlapped body;
  6. no compiler-generated instruction reads or writes m0 (the LDS-DMA statements set it and leave it).
  python tools/wsd_audit.py        (exit 1 on a finding; compiles only, runs on the CPU box)"""
import os, re, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "speech-anonymization_amd", "csrc", "sa_conv_wsd.hip")
if len(sys.argv) > 1:
    asm = open(sys.argv[1]).read()
else:
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "--offload-arch=gfx950", "-std=c++17",
                               "-save-temps=obj", "-c", src, "-o", os.path.join(d, "wsd.o")], cwd=os.path.dirname(src))
        asm = open(os.path.join(d, "sa_conv_wsd-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
bad = 0


def finding(msg):
    global bad
    bad += 1
    print("FINDING:", msg)


def regs_of(text):
    used = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(3):
            used.add(int(m.group(3)))
        else:
            used |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    return used


spills = [int(v) for v in re.findall(r"\.vgpr_spill_count:\s+(\d+)", asm)]
print("vgpr spills per kernel:", spills)
for m in re.finditer(r"; NumAgprs: (\d+)\n; TotalNumVgprs: (\d+)\n(?:.*\n)*?; AccumOffset: (\d+)", asm):
    if m.group(3) != "256" or m.group(1) != "256":
        finding(f"descriptor does not cover the reserved registers: NumAgprs {m.group(1)}, AccumOffset {m.group(3)}")
kernels = re.split(r"\n(?=_ZN12_GLOBAL__N_118sa_conv_wsd_kernelILi\d)", asm)[1:]
for k in kernels:
    name = k.split(":")[0]
    lines = k.split("s_endpgm")[0].split("\n")
    idx = [i for i, l in enumerate(lines) if "v_mfma_f32_32x32x16_bf16" in l]
    blk = [re.search(r"bf16 (v\[\d+:\d+\])", lines[i]).group(1) for i in idx]
    starts = [n for n, i in enumerate(idx) if lines[i].rstrip().endswith(", 0")]
    nsec = len(starts)
    sect = starts[1] - starts[0]
    print(f"{name[-30:]}: {len(idx)} MFMAs, {nsec} sections of {sect}, blocks {sorted(set(blk))}")
    for i in range(idx[0], idx[-1] + 1):
        if lines[i].strip().startswith("scratch_"):
            finding(f"scratch access inside the tile loop: {lines[i].strip()}")
    if nsec % 2 or any(starts[n] != n * sect for n in range(nsec)):
        finding("irregular sections")
        continue
    for n in range(nsec):
        if len(set(blk[starts[n]:starts[n] + sect])) != 1:
            finding(f"section {n}: more than one accumulator block")
    for body in range(nsec // 2):
        if blk[starts[2 * body]] == blk[starts[2 * body + 1]]:
            finding(f"body {body}: both sections on one block")
    # walk every body (two sections): classify lines as asm / compiler
    for body in range(nsec // 2):
        accregs = [regs_of(blk[starts[2 * body]]), regs_of(blk[starts[2 * body + 1]])]
        first, last = idx[starts[2 * body]], idx[starts[2 * body] + 2 * sect - 1]
        in_asm, since = True, [10 ** 6, 10 ** 6]          # MFMA statements since the last write of block 0 / 1
        left = [0, 0]                                        # MFMAs the block's running section still has to issue
        fastbody = any("landed" in lines[i] for i in range(first, last + 1))
        for i in range(first, last + 1):
            l = lines[i]
            t = l.strip()
            if "#ASMSTART" in l:
                in_asm = True
                continue
            if "#ASMEND" in l:
                in_asm = False
                continue
            if not t or t.startswith((";", ".")):
                continue
            if in_asm:
                if "v_mfma" in t:
                    b = 0 if regs_of(re.search(r"bf16 (v\[\d+:\d+\])", t).group(1)) == accregs[0] else 1
                    since[b] = 0
                    since[1 - b] += 1
                    left[b] = sect - 1 if t.rstrip().endswith(", 0") else left[b] - 1
                continue
            # compiler-generated instruction
            ops = t.split(None, 1)
            used = regs_of(ops[1]) if len(ops) > 1 else set()
            dst = regs_of(ops[1].split(",")[0]) if len(ops) > 1 and not ops[0].startswith(("global_store", "ds_write", "s_", "buffer_store")) else set()
            for b in (0, 1):
                if used & accregs[b] and left[b] > 0:
                    finding(f"body {body}: compiler instruction touches accumulator block {b} while it accumulates: {t}")
                elif (used - dst) & accregs[b] and since[b] < 3:
                    finding(f"body {body}: accumulator block {b} read {since[b]} MFMAs behind its last write: {t}")
            if fastbody and ops[0] == "s_waitcnt" and "vmcnt" in t:
                finding(f"body {body}: compiler-inserted wait in the overlapped body: {t}")
            if ops[0].startswith("scratch_"):
                finding(f"body {body}: scratch access: {t}")
    # reserved registers: anywhere in the kernel, compiler code must not name them
    in_asm = False
    for l in lines:
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        elif not in_asm and l.startswith("\t") and not l.strip().startswith((";", ".")):
            ops = l.strip().split(None, 1)
            if len(ops) > 1:
                if re.search(r"\bm0\b", ops[1]):
                    finding(f"compiler instruction uses m0 (the LDS-DMA statements do not preserve it): {l.strip()}")
                if regs_of(ops[1]) & set(range(240, 256)):
                    finding(f"compiler instruction names a reserved vector register: {l.strip()}")
                for m in re.finditer(r"\ba\[(\d+):(\d+)\]|\ba(\d+)\b", ops[1]):
                    hi = int(m.group(3) or m.group(2))
                    if hi >= 240:
                        finding(f"compiler instruction names a reserved accumulator register: {l.strip()}")
print("audit:", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)
