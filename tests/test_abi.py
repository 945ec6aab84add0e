"""CPU checks of the drop-in boundary: libsa_hip.so builds, loads, and exports exactly the
entry points include/sa_hip.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "sa_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"^\s*int\s+(sa_\w+)\s*\(", src, flags=re.M)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from speech_anonymization_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    decl = _declared()
    assert len(decl) >= 38
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/sa_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == decl, "ctypes binding and header disagree"


def test_ctypes_structs_match_header_field_order():
    from speech_anonymization_amd import _lib
    src = open(os.path.join(ROOT, "include", "sa_hip.h")).read()
    for st in (_lib.SaConvArgs, _lib.SaWgradArgs, _lib.SaEwArgs, _lib.SaTaps):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (st.__name__, st.__name__), src, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                m = re.search(r"(\w+)\s*(\[[^\]]*\])*\s*$", part.strip())
                names.append(m.group(1))
        assert names == [f[0] for f in st._fields_], st.__name__


def test_comm_entry_points_before_init():
    """the data-parallel exchange entry points (sa_comm_init / sa_comm_destroy) without a
    communicator: no RCCL, no HIP call is reached -- the codes include/sa_hip.h documents"""
    import errno
    from speech_anonymization_amd import _lib
    lib = _lib.load()
    assert lib.sa_comm_world() == 0
    assert lib.sa_comm_allreduce(None, ctypes.c_longlong(4), _lib.F32, 1, None) == -errno.ENOTCONN
    assert lib.sa_comm_allreduce_inline(None, ctypes.c_longlong(4), _lib.F32, 0, None) == -errno.ENOTCONN
    assert lib.sa_comm_join(None) == -errno.ENOTCONN
    assert lib.sa_comm_init(0, 1, None, 0) == -errno.EINVAL
    assert lib.sa_comm_init(2, 2, b"x" * 128, 0) == -errno.EINVAL
    assert lib.sa_comm_destroy() == 0


def test_missing_library_fails_loudly(monkeypatch):
    from speech_anonymization_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsa_hip.so")
    try:
        _lib.load()
    except _lib.SaHipError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load() must raise when the HIP library is missing")


def test_weight_stationary_kernel_isa_audit():
    """sa_conv_ws.hip issues its MFMAs from inline asm, so hipcc neither knows their latency nor pads
    hazards around them (cdna_hip_programming.md 5.7): every build is audited on the ISA -- no spills,
    no scratch, the accumulators in ONE register block per tile body, no compiler instruction on that
    block between a body's first and last MFMA (tools/ws_audit.py; cross-compiles, no GPU needed)."""
    import shutil
    import subprocess
    import sys
    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not installed")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ws_audit.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "audit: clean" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_fused_data_gradient_kernel_isa_audit():
    """sa_conv_wsd.hip keeps loads in flight across hand-scheduled slots into registers the compiler must not
    know (v240..v255 / a240..a255): every build is audited on the ISA -- no scratch in the tile loop, the
    accumulator blocks untouched while they accumulate and read no sooner than three MFMAs after the last
    write, the reserved registers named by no compiler instruction and covered by the kernel descriptor, no
    compiler-inserted s_waitcnt vmcnt in the fast body (tools/wsd_audit.py; cross-compiles, no GPU needed)."""
    import shutil
    import subprocess
    import sys
    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not installed")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wsd_audit.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "audit: clean" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
