"""Thin torch-tensor wrappers over the C ABI (one function per entry point family).

Tensors are device memory handles only; all arithmetic happens in libsa_hip.so on the
current torch stream.  Activations are channels-last [B, L, C] (see include/sa_hip.h).
"""
import ctypes as C
import functools

import torch

from . import _lib as L

# ------------------------------------------------------------------------------------
# conv geometry: tap tables of the row-gather GEMM for each kind of layer
# ------------------------------------------------------------------------------------
UP2 = [[(1, 0), (0, 2), (-1, 4)], [(1, 1), (0, 3)]]      # k5 s2 p2 (op1): convT fwd / conv-s2 dgrad


def taps_conv(K, dil, pad):
    """Conv1d forward (any stride via SA): input row = m*SA + k*dil - pad."""
    return [[(k * dil - pad, k) for k in range(K)]]


def taps_conv_dgrad_s1(K, dil, pad):
    """stride-1 Conv1d dgrad: dx[i] = sum_k dy[i + pad - k*dil] W[k]^T."""
    return [[(pad - k * dil, k) for k in range(K)]]


def taps_convT_dgrad(K=5, pad=2):
    """ConvTranspose1d(stride 2) dgrad = stride-2 conv over dy (SA = 2)."""
    return [[(k - pad, k) for k in range(K)]]


def _f(t):
    return L.ptr(t)


class _Profile:
    """HIP-event timing of ONE kernel family during bench.py's timed region (the events are
    recorded on the stream the kernel is launched on = torch's current stream)."""

    def __init__(self):
        self.key, self.kinds = None, {}

    def enable(self, key):
        self.key, self.kinds = key, {}

    def start(self, key):
        if key != self.key:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, e0, nbytes, flops, alg_bytes=None, kind=""):
        """nbytes: what the launch is designed to move; alg_bytes: SURVEY 8(d)'s algorithmic
        figure (inputs + outputs once, + the stored tensor a fused backward epilogue re-reads);
        kind: which device kernel served the launch (one family can be served by several)."""
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        k = self.kinds.setdefault(kind, {"ev": [], "bytes": 0, "alg_bytes": 0, "flops": 0})
        k["ev"].append((e0, e1))
        k["bytes"] += nbytes
        k["alg_bytes"] += nbytes if alg_bytes is None else alg_bytes
        k["flops"] += flops

    def collect(self):
        """one record per device kernel of the family, largest total time first"""
        torch.cuda.synchronize()
        out = []
        for kind, k in self.kinds.items():
            out.append({"kernel": kind or self.key, "family": self.key, "launches": len(k["ev"]),
                        "ms": sum(a.elapsed_time(b) for a, b in k["ev"]), "bytes": k["bytes"],
                        "alg_bytes": k["alg_bytes"], "flops": k["flops"]})
        out.sort(key=lambda r: -r["ms"])
        self.key, self.kinds = None, {}
        return out


PROFILE = _Profile()


# precision modes of the MFMA kernels: name -> (activation storage dtype, kernel dtype code)
PRECISIONS = {"f32": (torch.float32, L.F32), "bf16": (torch.bfloat16, L.BF16),
              "bf16x3": (torch.float32, L.BF16X3), "bf16x1f": (torch.float32, L.BF16X1F),
              # fp8: bf16 storage, OCP e4m3 MFMA operands (weights + activations) in the forward
              # convolutions; gradients on the bf16 kernels (BASELINE config 5)
              "fp8": (torch.bfloat16, L.FP8)}
# kernel code of the weight-gradient GEMM per model precision (SA_BF16X1F: see sa_common.h)
WGRAD_CODE = {"f32": L.F32, "bf16": L.BF16, "bf16x3": L.BF16X1F, "bf16x1f": L.BF16X1F, "fp8": L.BF16}
# kernel code of the data-gradient convolutions per model precision
DGRAD_CODE = {"f32": L.F32, "bf16": L.BF16, "bf16x3": L.BF16X3, "bf16x1f": L.BF16X1F, "fp8": L.BF16}


def _pack_geometry(shape, kind):
    """(Kw, K, N, sk, sn) of sa_pack_weights for a parameter of `shape` used as `kind`."""
    if kind in ("conv_fwd", "conv_dgrad"):
        Cout, Cin, Kw = shape
        if kind == "conv_fwd":
            return Kw, Cin, Cout, Kw, Cin * Kw
        return Kw, Cout, Cin, Cin * Kw, Kw
    Cin, Cout, Kw = shape
    if kind == "convT_fwd":
        return Kw, Cin, Cout, Cout * Kw, Kw
    return Kw, Cout, Cin, Kw, Cout * Kw


def _image_buffer(code, dtype, n, device):
    if code == L.FP8:                       # e4m3 image + its per-tensor scale (a float) behind it
        return torch.empty(n + 4, dtype=torch.uint8, device=device)
    if code == L.BF16X3:
        return torch.empty(2 * n, dtype=torch.bfloat16, device=device)
    if code in (L.BF16X1F, L.BF16):
        return torch.empty(n, dtype=torch.bfloat16, device=device)
    return torch.empty(n, dtype=torch.float32, device=device)


def pack_weights(w, kind, dtype, code=None):
    """w: fp32 parameter in PyTorch layout.  kind: conv_fwd | conv_dgrad | convT_fwd |
    convT_dgrad.  Returns the fragment-major operand image (flat tensor; for code BF16X3 the
    hi image followed by the lo image, both bf16)."""
    lib = L.load()
    code = L.dt_code(dtype) if code is None else code
    Kw, K, N, sk, sn = _pack_geometry(w.shape, kind)
    out = _image_buffer(code, dtype, Kw * K * N, w.device)
    L.check(lib.sa_pack_weights(code, _f(w), _f(out), Kw, K, N, sk, sn, 1, L.stream()),
            "sa_pack_weights")
    return out


class PackedWeights:
    """Persistent operand images of a fixed list of (parameter, kind, code) and the device-side
    descriptor table that lets one sa_pack_weights_multi launch refresh all of them."""

    def __init__(self, items, dtype):
        """items: list of (tag, fp32 parameter tensor, kind, code)."""
        self.images, descs = {}, (L.SaPackDesc * len(items))()
        dev = items[0][1].device
        for d, (tag, w, kind, code) in zip(descs, items):
            Kw, K, N, sk, sn = _pack_geometry(w.shape, kind)
            img = _image_buffer(code, dtype, Kw * K * N, dev)
            self.images[tag] = (img, code)
            d.src, d.dst, d.dtype = w.data_ptr(), img.data_ptr(), code
            d.ntaps, d.K, d.N, d.sk, d.sn, d.st = Kw, K, N, sk, sn, 1
            d.scale = img.data_ptr() + Kw * K * N if code == L.FP8 else None
            self.fp8 = getattr(self, "fp8", False) or code == L.FP8
        raw = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8)
        self.table = raw.to(dev)
        self.n = len(items)
        self.key = tuple(w.data_ptr() for _, w, _, _ in items)

    def refresh(self):
        if getattr(self, "fp8", False):     # per-tensor scales of the e4m3 images first
            L.check(L.load().sa_pack_scales_multi(_f(self.table), self.n, L.stream()), "sa_pack_scales_multi")
        L.check(L.load().sa_pack_weights_multi(_f(self.table), self.n, 64, L.stream()),
                "sa_pack_weights_multi")


@functools.lru_cache(maxsize=None)
def _conv_geometry(code, cin, cout, u, Lout):
    """(row tiles, statistics slabs) per utterance of a launch (cached: kernel choice and tile
    policy are fixed per process; conv_impl() clears the cache when it switches them)"""
    nt, ns = C.c_int(0), C.c_int(0)
    L.check(L.load().sa_conv_gemm_geometry(code, cin, cout, u, Lout, C.byref(nt), C.byref(ns)),
            "sa_conv_gemm_geometry")
    return nt.value, ns.value


def conv_impl(pingpong=False, tile_rows=0, pp_rows=0, ws=True):
    """kernel choice for the f32 / bf16x3 policies (A/B timing, kernel tests).  Default: the
    weight-stationary kernel (sa_conv_ws.hip) for the large 128->128 / 64->64 bf16x3 launches it covers, the
    one-tile-per-workgroup kernel for everything else; ws=False: one-tile kernel only; pingpong=True:
    the two-groups-in-anti-phase kernel (sa_conv_pp.hip).  Tile-row knobs of the latter two (0 = policy)."""
    lib = L.load()
    L.check(lib.sa_conv_gemm_set_impl(1 if pingpong else (2 if ws else 0)), "sa_conv_gemm_set_impl")
    L.check(lib.sa_conv_gemm_set_tile_rows(int(tile_rows)), "sa_conv_gemm_set_tile_rows")
    L.check(lib.sa_conv_pp_set_tile_rows(int(pp_rows)), "sa_conv_pp_set_tile_rows")
    _conv_geometry.cache_clear()


def conv_gemm(x, wp, bias, cin, cout, sa, u, phases, Lout, s1=None, t1=None, s2=None, t2=None,
              swish=False, relu=False, want_stats=False, out=None, code=None, ep=None, a_out=None,
              nb=None, want_pro_stats=False):
    """x [B, Lin, cin] -> y [B, Lout, cout] (+ per-tile partial stats [B, ntiles, cout, 2]).
    ep: fused backward epilogue dict(mode=1|2, x=, g2=, s1=, t1=, mean=, rstd=, xp_is_act=,
    per_c=) -- see SaConvArgs.ep_* in include/sa_hip.h.  a_out: optional bf16 [B, Lin, cin] tensor
    that receives the transformed input rows (the A operand of wgrad(..., x_pre=True)).
    nb=dict(x=, c1=, c2=, c3=, per_c=, relu_mask=, want_colsum=): normalisation-backward prologue
    (SaConvArgs.nb_*); with want_colsum the per-tile column sums [B, ntiles, cin] are returned last.
    want_pro_stats: also return per-tile (sum, sumsq) of the transformed input rows [B, ntiles, cin, 2]."""
    lib = L.load()
    B, Lin, _ = x.shape
    assert x.shape[2] == cin
    y = out if out is not None else torch.empty(B, Lout, cout, dtype=x.dtype, device=x.device)
    kc = L.dt_code(x.dtype) if code is None else code
    nt, nslab = _conv_geometry(kc, cin, cout, u, Lout)
    stats = torch.empty(B, nslab, cout, 2, dtype=torch.float32, device=x.device) if want_stats else None
    a = L.SaConvArgs()
    a.x, a.wp, a.bias, a.y = _f(x), _f(wp), _f(bias), _f(y)
    a.s1, a.t1, a.s2, a.t2 = _f(s1), _f(t1), _f(s2), _f(t2)
    a.swish, a.relu, a.stats = int(swish), int(relu), _f(stats)
    a.B, a.Lin, a.Lout = B, Lin, Lout
    a.taps = L.make_taps(phases)
    if kc == L.FP8:                          # the scale sits behind the e4m3 image
        a.wscale = C.c_void_p(wp.data_ptr() + wp.numel() - 4)
    if a_out is not None:
        assert a_out.dtype == torch.bfloat16 and a_out.shape == x.shape
        a.a_out = _f(a_out)
    pro_stats = None
    if want_pro_stats:
        pro_stats = torch.empty(B, nt, cin, 2, dtype=torch.float32, device=x.device)
        a.pro_stats = _f(pro_stats)
    colsum = None
    if nb:
        assert nb["x"].shape == x.shape and nb["x"].dtype == x.dtype
        a.nb_x, a.nb_c1, a.nb_c2, a.nb_c3 = _f(nb["x"]), _f(nb["c1"]), _f(nb["c2"]), _f(nb["c3"])
        a.nb_bstride, a.nb_relu_mask = (0 if nb.get("per_c") else cin), int(bool(nb.get("relu_mask")))
        if nb.get("want_colsum"):
            colsum = torch.empty(B, nt, cin, dtype=torch.float32, device=x.device)
            a.nb_colsum = _f(colsum)
    if ep:
        a.ep_mode, a.ep_xp_is_act = int(ep["mode"]), int(bool(ep.get("xp_is_act")))
        a.ep_bstride = 0 if ep.get("per_c") else cout
        a.ep_x, a.ep_g2 = _f(ep["x"]), _f(ep.get("g2"))
        a.ep_s1, a.ep_t1 = _f(ep.get("s1")), _f(ep.get("t1"))
        a.ep_mean, a.ep_rstd = _f(ep.get("mean")), _f(ep.get("rstd"))
        if ep.get("g2k") is not None:
            a.ep_g2k1, a.ep_g2k2, a.ep_g2k3 = (_f(t) for t in ep["g2k"])
    e0 = PROFILE.start(f"conv_gemm({cin},{cout},{sa},{u})") if PROFILE.key else None
    L.check(lib.sa_conv_gemm(kc, cin, cout, sa, u, C.byref(a), L.stream()),
            f"sa_conv_gemm({cin},{cout},{sa},{u})")
    if e0 is not None:
        ntap = sum(len(p) for p in phases)
        esz = x.element_size()
        extra = (a_out.numel() * 2 if a_out is not None else 0)        # bf16 operand cache written
        if ep:
            extra += y.numel() * esz * (2 if ep.get("g2") is not None else 1)   # stored forward tensor (+ 2nd gradient) read
        if nb:
            extra += x.numel() * esz                                             # stored forward tensor of the layer above
        io = (x.numel() + y.numel()) * esz
        route = lib.sa_conv_gemm_route(kc, cin, cout, sa, u, C.byref(a))
        tname = {L.F32: "float", L.BF16: "bf16_t", L.BF16X3: "bf16x3_t", L.BF16X1F: "bf16x1f_t", L.FP8: "fp8_t"}[kc]
        ws_mode = (5 if s2 is not None else 0) if s1 is None else 3 if want_pro_stats else 4 if s2 is not None else 1
        # (the five instances <taps, span, prologue, epilogue> of the fused data-gradient kernel are ONE
        # kernel for the roofline record: same source, same structure, 1 launch per step each)
        kind = (f"sa_conv_wsd_kernel ({tname}, {cin}->{cout}; 5 instances)" if route == 3 else
                f"sa_conv_ws_kernel<{ws_mode},{ntap}> ({tname}, {cin}->{cout})" if route == 2 else
                f"sa_conv_pp_kernel<{tname},{cin},{cout},{sa},{u}>" if route == 1 else
                f"sa_conv_gemm_kernel<{tname},{cin},{cout},{sa},{u}{',nb prologue' if nb else ''}>")
        PROFILE.stop(e0, io + ntap * cin * cout * esz + extra,
                     2 * B * (-(-Lout // u)) * ntap * cin * cout,
                     alg_bytes=io + (y.numel() * esz if ep else 0), kind=kind)
    out_t = (y, stats) if want_stats else (y,)
    if nb and nb.get("want_colsum"):
        out_t = out_t + (colsum,)
    if want_pro_stats:
        out_t = out_t + (pro_stats,)
    return out_t if len(out_t) > 1 else y


if __import__("os").environ.get("SA_WS_FIRST_PLAIN") == "1":         # timing A/B: first tile of a range on the plain path
    L.load().sa_conv_ws_set_bcost(9 | 0x10000)
    L.load().sa_conv_wsd_set_bcost(9 | 0x10000)

WGRAD_TARGET_WGS = {True: 256, False: 512}
if __import__("os").environ.get("SA_WG_TARGETS"):                    # tuning override "big,small"
    _b, _s = __import__("os").environ["SA_WG_TARGETS"].split(",")
    WGRAD_TARGET_WGS = {True: int(_b), False: int(_s)}


def wgrad(x, dy, cin, cout, sa, u, taps, Mrows, dst, dst_strides, s1=None, t1=None, s2=None,
          t2=None, swish=False, accumulate=False, target_wgs=None, code=None, x_pre=False,
          dy_pre=False, defer=None):
    """taps: list of (row_offset, phase) per weight tap.  dst: fp32 parameter-gradient tensor in
    PyTorch layout; dst_strides = (s_ci, s_co, s_tap).  x_pre: x is the bf16 a_out tensor of the
    forward conv_gemm (already transformed; s1..swish are ignored); dy_pre: dy is the bf16 a_out of
    the data-gradient conv_gemm that formed it in its normalisation-backward prologue."""
    lib = L.load()
    B, Lin, _ = x.shape
    Ldy = dy.shape[1]
    if x_pre:
        assert x.dtype == torch.bfloat16 and code in (L.BF16X1F, L.BF16)
    nt = len(taps)
    kw = lib.sa_wgrad_kw(cin, cout)
    if target_wgs is None:
        # one 8-wave workgroup (128-wide channel blocks) or two-three 4-wave ones fit a CU: size
        # the row chunks so that the grid is one resident wave of workgroups over the 256 CUs
        target_wgs = WGRAD_TARGET_WGS[(cin // 32) * (cout // 32) >= 8]
    chunk = max(64, -(-Mrows * B // target_wgs))
    chunk = -(-chunk // 64) * 64
    nchunk = -(-Mrows // chunk)
    slabs = torch.empty(B * nchunk * kw * nt * cin * cout, dtype=torch.float32, device=x.device)
    a = L.SaWgradArgs()
    a.x, a.dy, a.slabs = _f(x), _f(dy), _f(slabs)
    a.s1, a.t1, a.s2, a.t2, a.swish = _f(s1), _f(t1), _f(s2), _f(t2), int(swish)
    a.B, a.Lin, a.Ldy, a.Mrows, a.chunk, a.nchunk, a.ntaps = B, Lin, Ldy, Mrows, chunk, nchunk, nt
    for i, (off, ph) in enumerate(taps):
        a.off[i], a.ph[i] = off, ph
    a.x_pre, a.dy_pre = int(x_pre), int(dy_pre)
    if dy_pre:
        assert x_pre and dy.dtype == torch.bfloat16
    L.check(lib.sa_wgrad((L.BF16X1F if dy_pre else L.dt_code(dy.dtype)) if code is None else code, cin, cout, sa, u,
                         C.byref(a), L.stream()),
            f"sa_wgrad({cin},{cout},{sa},{u})")
    sk, sn, st = dst_strides
    if defer is not None:           # the reducer joins the others of its backward stage (wgrad_reduce_multi)
        defer.append((slabs, dst, B * nchunk * kw, nt, cin, cout, sk, sn, st, int(accumulate)))
        return dst
    L.check(lib.sa_wgrad_reduce(_f(slabs), _f(dst), B * nchunk * kw, nt, cin, cout, sk, sn, st,
                                int(accumulate), L.stream()), "sa_wgrad_reduce")
    return dst


def wgrad_reduce_multi(items):
    """items: the deferred reducers of wgrad(..., defer=items): (slabs, dst, nslab, ntaps, cin, cout, sk, sn, st,
    accumulate) each -- one launch per eight of them (sa_wgrad_reduce_multi; same bits as one launch each)"""
    lib = L.load()
    for i0 in range(0, len(items), L.WRED_MAX):
        m = L.SaWredMulti()
        chunk = items[i0:i0 + L.WRED_MAX]
        m.n = len(chunk)
        for d, (slabs, dst, nslab, nt, cin, cout, sk, sn, st, acc) in zip(m.d, chunk):
            d.slabs, d.dst = slabs.data_ptr(), dst.data_ptr()
            d.nslab, d.ntaps, d.cin, d.cout, d.sk, d.sn, d.st, d.accumulate = nslab, nt, cin, cout, sk, sn, st, acc
        L.check(lib.sa_wgrad_reduce_multi(C.byref(m), L.stream()), "sa_wgrad_reduce_multi")


def conv1toC(x, w, bias, dtype, flip=False, want_stats=False, ep=None):
    """ep=dict(x=, s1=, t1=, mean=, rstd=): fused [InstanceNorm -> x*sigmoid(x)] backward epilogue."""
    lib = L.load()
    B, Ln = x.shape
    y = torch.empty(B, Ln, 32, dtype=dtype, device=x.device)
    nt = lib.sa_conv1toC_ntiles(Ln)
    stats = torch.empty(B, nt, 32, 2, dtype=torch.float32, device=x.device) if want_stats else None
    e = ep or {}
    L.check(lib.sa_conv1toC(L.dt_code(dtype), _f(x), _f(w), _f(bias), _f(y), B, Ln, int(flip),
                            _f(stats), _f(e.get("x")), _f(e.get("s1")), _f(e.get("t1")),
                            _f(e.get("mean")), _f(e.get("rstd")), L.stream()), "sa_conv1toC")
    return (y, stats) if want_stats else y


def convCto1(x, w, bias, s1=None, t1=None, swish=False, flip=False):
    lib = L.load()
    B, Ln, _ = x.shape
    y = torch.empty(B, Ln, dtype=torch.float32, device=x.device)
    L.check(lib.sa_convCto1(L.dt_code(x.dtype), _f(x), _f(w), _f(bias), _f(y), B, Ln, _f(s1), _f(t1),
                            int(swish), int(flip), L.stream()), "sa_convCto1")
    return y


def wgrad1C(u, v, dst, flip=False, s1=None, t1=None, swish=False, accumulate=False, chunk=2048):
    lib = L.load()
    B, Ln = u.shape
    nch = lib.sa_wgrad1C_nchunk(Ln, chunk)
    slabs = torch.empty(B * nch, 32 * 15, dtype=torch.float32, device=u.device)
    L.check(lib.sa_wgrad1C(L.dt_code(v.dtype), _f(u), _f(v), _f(slabs), B, Ln, chunk, int(flip),
                           _f(s1), _f(t1), int(swish), L.stream()), "sa_wgrad1C")
    L.check(lib.sa_sum_slabs(_f(slabs), _f(dst), B * nch, 32 * 15, int(accumulate), L.stream()),
            "sa_sum_slabs")
    return dst


def sum_partials(part, nbatch, n=None, rows=False):
    """part [nbatch][nslab][n] (contiguous) -> [nbatch, n] fixed-order sums.  n defaults to the
    product of the last two dims (the [.., C, 2] layout of the statistics slabs).  rows=True (with
    nbatch == 1): return the per-utterance partial rows [R, n] of the two-level reduction instead of
    their sum -- the BatchNorm finalisers add them (fin_bn_fwd / fin_norm_bwd take R rows)."""
    lib = L.load()
    if n is None:
        n = part.shape[-2] * part.shape[-1]
    total = part.numel()
    nslab = total // (nbatch * n)
    if nbatch == 1 and nslab >= 512 and part.dim() == 4 and part.shape[0] > 1:
        # two levels (per utterance, then over utterances): keeps the first level wide
        R = part.shape[0]
        mid = torch.empty(R, n, dtype=torch.float64, device=part.device)
        L.check(lib.sa_sum_partials(_f(part), _f(mid), R, nslab // R, n, L.stream()), "sa_sum_partials")
        if rows:
            return mid
        out = torch.empty(1, n, dtype=torch.float64, device=part.device)
        L.check(lib.sa_sum_rows_d(_f(mid), _f(out), R, n, L.stream()), "sa_sum_rows_d")
        return out
    out = torch.empty(nbatch, n, dtype=torch.float64, device=part.device)
    L.check(lib.sa_sum_partials(_f(part), _f(out), nbatch, nslab, n, L.stream()), "sa_sum_partials")
    return out


_tickets = {}


def reduce_finalize(mode, part, nbatch, Cc, ncomp=2, count=1.0, gamma=None, beta=None, mean=None, rstd=None,
                    sign=1.0, dgamma=None, dbeta=None, db=None, run_mean=None, run_var=None, eps=1e-5,
                    momentum=0.1):
    """sa_reduce_finalize: the slab sums of `part` ([nbatch][nslab][Cc*ncomp], any trailing layout)
    and the finaliser that consumes them in ONE launch (SaFinArgs in include/sa_hip.h).  Returns
    (mean, rstd, scale, shift) for the FWD modes, (c1, c2, c3) for the BWD modes, db for FIN_BIAS.
    Single-process statistics only (under SyncBatchNorm the sums are all-reduced between the two
    halves: sum_partials + fin_* stay separate there)."""
    n = Cc * ncomp
    dev = part.device
    nslab = part.numel() // (nbatch * n)
    a = L.SaFinArgs()
    a.part, a.nbatch, a.nslab, a.n, a.C, a.ncomp, a.mode = _f(part), nbatch, nslab, n, Cc, ncomp, mode
    a.count, a.eps, a.momentum, a.sign = float(count), eps, momentum, sign
    a.gamma, a.beta, a.mean, a.rstd = _f(gamma), _f(beta), _f(mean), _f(rstd)
    a.dgamma, a.dbeta, a.db, a.run_mean, a.run_var = _f(dgamma), _f(dbeta), _f(db), _f(run_mean), _f(run_var)
    out = None
    if mode != L.FIN_IN_FWD:
        rows = torch.empty(nbatch, n, dtype=torch.float64, device=dev)
        # one ticket per 32-output chunk (sa_reduce_finalize indexes tickets[blockIdx.x]); the buffer is
        # per (device, stream): launches on one stream are ordered, two streams must not share tickets
        nchunk = -(-n // 32)
        key = (dev, L.stream().value if hasattr(L.stream(), "value") else int(L.stream() or 0))
        tk = _tickets.get(key)
        if tk is None or tk.numel() < nchunk:   # zero-initialised once; the kernel resets what it used
            tk = _tickets[key] = torch.zeros(max(256, nchunk), dtype=torch.int32, device=dev)
        a.rows, a.tickets = _f(rows), _f(tk)
    if mode in (L.FIN_IN_FWD, L.FIN_BN_FWD):
        out = torch.empty(4, nbatch * Cc if mode == L.FIN_IN_FWD else Cc, dtype=torch.float32, device=dev)
        a.o0, a.o1, a.o2, a.o3 = _f(out[0]), _f(out[1]), _f(out[2]), _f(out[3])
    elif mode in (L.FIN_IN_BWD, L.FIN_BN_BWD):
        out = torch.empty(3, nbatch * Cc if mode == L.FIN_IN_BWD else Cc, dtype=torch.float32, device=dev)
        a.o0, a.o1, a.o2 = _f(out[0]), _f(out[1]), _f(out[2])
    L.check(L.load().sa_reduce_finalize(C.byref(a), L.stream()), "sa_reduce_finalize")
    if out is None:
        return db
    return tuple(out[i] for i in range(out.shape[0]))


def fin_in_fwd(sums, B, Cc, n, gamma, beta, eps=1e-5):
    lib = L.load()
    o = torch.empty(4, B, Cc, dtype=torch.float32, device=sums.device)
    L.check(lib.sa_fin_in_fwd(_f(sums), B, Cc, n, _f(gamma), _f(beta), C.c_float(eps), _f(o[0]),
                              _f(o[1]), _f(o[2]), _f(o[3]), L.stream()), "sa_fin_in_fwd")
    return o[0], o[1], o[2], o[3]          # mean, rstd, scale, shift


def fin_bn_fwd(sums, Cc, count, gamma, beta, run_mean=None, run_var=None, eps=1e-5, momentum=0.1,
               count_dev=None):
    """count_dev: fp64 device scalar holding the (all-reduced) element count; overrides count."""
    lib = L.load()
    o = torch.empty(4, Cc, dtype=torch.float32, device=sums.device)
    L.check(lib.sa_fin_bn_fwd(_f(sums), sums.numel() // (2 * Cc), Cc, C.c_double(count), _f(gamma), _f(beta), C.c_float(eps),
                              C.c_float(momentum), _f(run_mean), _f(run_var), _f(o[0]), _f(o[1]),
                              _f(o[2]), _f(o[3]), _f(count_dev), L.stream()), "sa_fin_bn_fwd")
    return o[0], o[1], o[2], o[3]


def fin_bn_eval(Cc, gamma, beta, run_mean, run_var, eps=1e-5):
    lib = L.load()
    o = torch.empty(4, Cc, dtype=torch.float32, device=gamma.device)
    L.check(lib.sa_fin_bn_eval(Cc, _f(gamma), _f(beta), C.c_float(eps), _f(run_mean), _f(run_var),
                               _f(o[0]), _f(o[1]), _f(o[2]), _f(o[3]), L.stream()), "sa_fin_bn_eval")
    return o[0], o[1], o[2], o[3]


def fin_norm_bwd(sums, lsums, groups, Cc, n, gamma, mean, rstd, sign=1.0, dgamma=None, dbeta=None,
                 n_dev=None):
    lib = L.load()
    o = torch.empty(3, groups, dtype=torch.float32, device=sums.device)
    R = sums.numel() // (2 * groups)
    assert lsums is None or lsums.numel() == sums.numel()
    L.check(lib.sa_fin_norm_bwd(_f(sums), _f(lsums), R, groups, Cc, C.c_double(n), _f(gamma), _f(mean),
                                _f(rstd), C.c_float(sign), _f(o[0]), _f(o[1]), _f(o[2]), _f(dgamma),
                                _f(dbeta), _f(n_dev), L.stream()), "sa_fin_norm_bwd")
    return o[0], o[1], o[2]


def bias_multi(items):
    """items: [(part [nbatch, nslab, C(, ncomp)] fp32 slabs, nbatch, C, ncomp, db fp32 [C])]: the bias gradients
    of several layers in two launches (sa_bias_multi; same bits as sum_partials + fin_bias per layer)."""
    lib = L.load()
    dev = items[0][0].device
    rows = torch.empty(sum(nb * cc for _, nb, cc, _, _ in items), dtype=torch.float64, device=dev)
    rp, esz = rows.data_ptr(), 8
    for i0 in range(0, len(items), L.BIAS_MAX):
        m = L.SaBiasMulti()
        chunk = items[i0:i0 + L.BIAS_MAX]
        m.n = len(chunk)
        for d, (part, nb, cc, ncomp, db) in zip(m.d, chunk):
            d.part, d.rows, d.db = part.data_ptr(), rp, db.data_ptr()
            d.nbatch, d.nslab, d.C, d.ncomp = nb, part.numel() // (nb * cc * ncomp), cc, ncomp
            rp += nb * cc * esz
        L.check(lib.sa_bias_multi(C.byref(m), L.stream()), "sa_bias_multi")


def fin_bias(sums, B, Cc, db, ncomp=2):
    """db[c] = sum_b sums[b][c][0]; sums [B, Cc, ncomp] fp64."""
    L.check(L.load().sa_fin_bias(_f(sums), B, Cc, ncomp, _f(db), L.stream()), "sa_fin_bias")
    return db


def ew(kind, g, x, Cc, out=None, g2=None, s1=None, t1=None, mean=None, rstd=None, c1=None, c2=None,
       c3=None, actbwd=False, xp_is_act=False, relu_mask=False, per_c=False, want_stats=True):
    """kind: "stats" | "apply".  Returns partial stats [B, ntiles, C, 2] (or None)."""
    lib = L.load()
    B, Ln, _ = x.shape
    nt = lib.sa_ew_ntiles(Ln)
    stats = torch.empty(B, nt, Cc, 2, dtype=torch.float32, device=x.device) if want_stats else None
    a = L.SaEwArgs()
    a.g, a.g2, a.x, a.out = _f(g), _f(g2), _f(x), _f(out)
    a.s1, a.t1, a.mean, a.rstd = _f(s1), _f(t1), _f(mean), _f(rstd)
    a.c1, a.c2, a.c3 = _f(c1), _f(c2), _f(c3)
    a.actbwd, a.xp_is_act, a.relu_mask = int(actbwd), int(xp_is_act), int(relu_mask)
    a.bstride = 0 if per_c else Cc
    a.stats, a.B, a.L = _f(stats), B, Ln
    fn = lib.sa_ew_stats if kind == "stats" else lib.sa_ew_apply
    L.check(fn(L.dt_code(x.dtype), Cc, C.byref(a), L.stream()), f"sa_ew_{kind}")
    return stats


def act_stats(x, s1, t1, swish=True):
    lib = L.load()
    B, Ln, Cc = x.shape
    nt = lib.sa_ew_ntiles(Ln)
    stats = torch.empty(B, nt, Cc, 2, dtype=torch.float32, device=x.device)
    L.check(lib.sa_act_stats(L.dt_code(x.dtype), Cc, _f(x), _f(s1), _f(t1), int(swish), _f(stats), B,
                             Ln, L.stream()), "sa_act_stats")
    return stats


def pool_fwd(r, scale, shift, noise=None, eps=1e-5):
    lib = L.load()
    B, Ln, _ = r.shape
    nseg = lib.sa_pool_nseg(B)
    part = torch.empty(B, nseg, 128, 128, 2, dtype=torch.float32, device=r.device)
    L.check(lib.sa_pool_fwd(L.dt_code(r.dtype), _f(r), _f(scale), _f(shift), _f(part), B, Ln, nseg,
                            L.stream()), "sa_pool_fwd")
    sums = torch.empty(B, 128, 2, dtype=torch.float64, device=r.device)
    L.check(lib.sa_pool_gather(_f(part), B, nseg, Ln, _f(sums), L.stream()), "sa_pool_gather")
    pooled = torch.empty(B, 256, dtype=torch.float32, device=r.device)
    mean = torch.empty(B, 128, dtype=torch.float32, device=r.device)
    sd = torch.empty(B, 128, dtype=torch.float32, device=r.device)
    L.check(lib.sa_pool_fin(_f(sums), B, Ln, _f(noise), C.c_float(eps), _f(pooled), _f(mean), _f(sd),
                            L.stream()), "sa_pool_fin")
    return pooled, mean, sd


def pool_bwd(r, scale, shift, dpooled, mean, sd, bn=None):
    """bn=(mean[128], rstd[128]) of the BatchNorm that produced the pooled tensor: also return the
    partial (sum g, sum g*xhat) slabs [B, ntiles, 128, 2] of its backward."""
    lib = L.load()
    B, Ln, _ = r.shape
    g = torch.empty_like(r)
    st = torch.empty(B, -(-Ln // 256), 128, 2, dtype=torch.float32, device=r.device) if bn else None
    L.check(lib.sa_pool_bwd(L.dt_code(r.dtype), _f(r), _f(scale), _f(shift), _f(dpooled), _f(mean),
                            _f(sd), _f(g), B, Ln, _f(bn[0]) if bn else None, _f(bn[1]) if bn else None,
                            _f(st), L.stream()), "sa_pool_bwd")
    return (g, st) if bn else g


def dense(X, W, bias, N, K, ps=None, pt=None, relu=False, transpose_w=False):
    """Y[M,N] = act(P(X)[M,K] @ Wm + bias).  W is an nn.Linear weight; transpose_w=False uses
    Wm[k][n] = W[n][k] (forward), True uses Wm[k][n] = W[k][n] (data gradient)."""
    lib = L.load()
    M = X.shape[0]
    Y = torch.empty(M, N, dtype=torch.float32, device=X.device)
    sbk, sbn = (W.shape[1], 1) if transpose_w else (1, W.shape[1])
    L.check(lib.sa_dense(_f(X), X.shape[1], _f(ps), _f(pt), _f(W), sbk, sbn, _f(bias), _f(Y), N, M, N,
                         K, int(relu), L.stream()), "sa_dense")
    return Y


def colsums(X, H=None, hmean=None, hrstd=None, out0=None, out1=None):
    """out0 / out1: fp32 [N] tensors (parameter-gradient views) that receive columns 0 / 1 of the
    fp64 sums rounded to fp32 -- no separate copy launch"""
    lib = L.load()
    M, N = X.shape
    s = torch.empty(N, 2, dtype=torch.float64, device=X.device)
    for o in (out0, out1):
        assert o is None or (o.dtype == torch.float32 and o.numel() == N and o.is_contiguous())
    L.check(lib.sa_colsums(_f(X), _f(H), _f(hmean), _f(hrstd), M, N, _f(s), _f(out0), _f(out1), L.stream()),
            "sa_colsums")
    return s


def bn2d_bwd(G, H, sums, count, gamma, mean, rstd, relu_mask, count_dev=None):
    lib = L.load()
    M, N = G.shape
    dH = torch.empty_like(G)
    L.check(lib.sa_bn2d_bwd(_f(G), _f(H), _f(sums), C.c_double(count), _f(gamma), _f(mean), _f(rstd),
                            int(relu_mask), M, N, _f(dH), _f(count_dev), L.stream()), "sa_bn2d_bwd")
    return dH


def dense_wgrad(dY, X, dW, ps=None, pt=None):
    lib = L.load()
    M, N = dY.shape
    K = X.shape[1]
    L.check(lib.sa_dense_wgrad(_f(dY), _f(X), _f(ps), _f(pt), M, N, K, _f(dW), L.stream()),
            "sa_dense_wgrad")
    return dW


def head_max_rows():
    return L.load().sa_head_max_rows()


def head_fwd(pooled, P, bn1, bn2, eps=1e-5, momentum=0.1):
    """The whole FC head in one launch (sa_head_fwd).  P: the sex_classifier.classify.* parameters by
    short name ("0.weight" ...); bn1 / bn2: the BatchNorm modules (running statistics are updated).
    Returns H1, f1 (mean, rstd, scale, shift), H2, f2, logp."""
    M = pooled.shape[0]
    dev = pooled.device
    H1 = torch.empty(M, 128, dtype=torch.float32, device=dev)
    H2 = torch.empty(M, 64, dtype=torch.float32, device=dev)
    f1 = torch.empty(4, 128, dtype=torch.float32, device=dev)
    f2 = torch.empty(4, 64, dtype=torch.float32, device=dev)
    logp = torch.empty(M, 2, dtype=torch.float32, device=dev)
    L.check(L.load().sa_head_fwd(_f(pooled), _f(P["0.weight"]), _f(P["0.bias"]), _f(P["2.weight"]), _f(P["2.bias"]),
                                 _f(bn1.running_mean), _f(bn1.running_var), _f(P["3.weight"]), _f(P["3.bias"]),
                                 _f(P["5.weight"]), _f(P["5.bias"]), _f(bn2.running_mean), _f(bn2.running_var),
                                 _f(P["6.weight"]), _f(P["6.bias"]), _f(H1), _f(f1), _f(H2), _f(f2), _f(logp), M,
                                 C.c_float(eps), C.c_float(momentum), L.stream()), "sa_head_fwd")
    return H1, tuple(f1[i] for i in range(4)), H2, tuple(f2[i] for i in range(4)), logp


def head_bwd(dlogp, logp, pooled, H1, f1, H2, f2, P, grads):
    """sa_head_bwd: grads maps the short parameter names to fp32 gradient views (or None); returns dpooled.
    f1 / f2: the tuples head_fwd returned (rows of one [4][N] tensor)."""
    M = pooled.shape[0]
    dpooled = torch.empty(M, 256, dtype=torch.float32, device=pooled.device)
    g = lambda k: _f(grads.get(k))
    L.check(L.load().sa_head_bwd(_f(dlogp), _f(logp), _f(pooled), _f(H1), _f(f1[0]), _f(H2), _f(f2[0]),
                                 _f(P["0.weight"]), _f(P["2.weight"]), _f(P["3.weight"]), _f(P["5.weight"]),
                                 _f(P["6.weight"]), g("0.weight"), g("0.bias"), g("2.weight"), g("2.bias"),
                                 g("3.weight"), g("3.bias"), g("5.weight"), g("5.bias"), g("6.weight"), g("6.bias"),
                                 _f(dpooled), M, L.stream()), "sa_head_bwd")
    return dpooled


def log_softmax(X):
    Y = torch.empty_like(X)
    L.check(L.load().sa_log_softmax(_f(X), _f(Y), X.shape[0], X.shape[1], L.stream()), "sa_log_softmax")
    return Y


def log_softmax_bwd(dY, Y):
    dX = torch.empty_like(Y)
    L.check(L.load().sa_log_softmax_bwd(_f(dY), _f(Y), _f(dX), Y.shape[0], Y.shape[1], L.stream()),
            "sa_log_softmax_bwd")
    return dX


def recon_loss(a, b, kind, want_grad=True):
    """kind "l1" | "mse"; returns (loss[1], grad like a or None)."""
    lib = L.load()
    n = a.numel()
    loss = torch.empty(1, dtype=torch.float32, device=a.device)
    grad = torch.empty_like(a) if want_grad else None
    ws = torch.empty(lib.sa_loss_workspace_bytes() // 8, dtype=torch.float64, device=a.device)
    L.check(lib.sa_recon_loss(_f(a), _f(b), C.c_longlong(n), 0 if kind == "l1" else 1, _f(grad),
                              _f(loss), _f(ws), L.stream()), "sa_recon_loss")
    return loss, grad


def cls_losses(logp, label, want_grad=True):
    lib = L.load()
    B, NC = logp.shape
    out = torch.empty(2, dtype=torch.float32, device=logp.device)
    dn = torch.empty_like(logp) if want_grad else None
    dc = torch.empty_like(logp) if want_grad else None
    L.check(lib.sa_cls_losses(_f(logp), _f(label), B, NC, _f(out), _f(dn), _f(dc), L.stream()),
            "sa_cls_losses")
    return out, dn, dc


def cosine_loss(x1, x2, want_grad=False):
    lib = L.load()
    B, S, D = x1.shape
    rl = torch.empty(B * S, dtype=torch.float32, device=x1.device)
    loss = torch.empty(1, dtype=torch.float32, device=x1.device)
    dx1 = torch.empty_like(x1) if want_grad else None
    L.check(lib.sa_cosine_loss(_f(x1), _f(x2), B, S, D, _f(rl), _f(loss), _f(dx1), L.stream()),
            "sa_cosine_loss")
    return loss, dx1


def cosine_rows(x1, x2):
    """per-row cosine similarity of two [B, D] tensors (sa_cosine_loss's row output: 1 - loss of
    the row; eps 1e-6 on |x1||x2|, where torch.nn.CosineSimilarity(eps=1e-8) of the reference's
    evaluation hook differs only for vectors of norm < 1e-3)."""
    lib = L.load()
    B, D = x1.shape
    x1, x2 = x1.contiguous().float(), x2.contiguous().float()
    rl = torch.empty(B, dtype=torch.float32, device=x1.device)
    loss = torch.empty(1, dtype=torch.float32, device=x1.device)
    L.check(lib.sa_cosine_loss(_f(x1), _f(x2), B, 1, D, _f(rl), _f(loss), None, L.stream()),
            "sa_cosine_loss")
    return 1.0 - rl


def cluster_mi(X, y, idx=None, ncls=2, k=3):
    lib = L.load()
    iters, n = (idx.shape if idx is not None else (1, X.shape[0]))
    mi = torch.empty(iters, dtype=torch.float32, device=X.device)
    L.check(lib.sa_cluster_mi(_f(X), _f(y), _f(idx), iters, n, X.shape[1], ncls, k, _f(mi), L.stream()),
            "sa_cluster_mi")
    return mi


# ---- element-wise passes of the frozen recogniser (csrc/sa_asr.hip; asr.py) ----
def add_layernorm(x, r, gamma, beta, eps, save):
    """y = LayerNorm(bf16(x + r)) * gamma + beta over the last dimension (r may be None); bf16 tensors.
    save: also return the stored sum s and the statistics the backward re-reads."""
    d = x.shape[-1]
    rows = x.numel() // d
    y = torch.empty_like(x)
    s = torch.empty_like(x) if save else None
    stat = torch.empty(rows, 2, dtype=torch.float32, device=x.device) if save else None
    L.check(L.load().sa_add_layernorm_fwd(_f(x), _f(r), _f(gamma), _f(beta), _f(y), _f(s), _f(stat), rows, d,
                                          C.c_float(eps), L.stream()), "sa_add_layernorm_fwd")
    return y, s, stat


def layernorm_bwd(dy, s, stat, gamma):
    d = s.shape[-1]
    ds = torch.empty_like(s)
    L.check(L.load().sa_layernorm_bwd(_f(dy), _f(s), _f(stat), _f(gamma), _f(ds), s.numel() // d, d, L.stream()),
            "sa_layernorm_bwd")
    return ds


def reflect_pad(x):
    """[B, T, F, C] bf16 -> [B, T + 2, F + 2, C], reflect padding of T and F by one"""
    B, T, F_, Cc = x.shape
    y = torch.empty(B, T + 2, F_ + 2, Cc, dtype=x.dtype, device=x.device)
    L.check(L.load().sa_reflect_pad_fwd(_f(x), _f(y), B, T, F_, Cc, L.stream()), "sa_reflect_pad_fwd")
    return y


def reflect_pad_bwd(dy):
    B, T2, F2, Cc = dy.shape
    dx = torch.empty(B, T2 - 2, F2 - 2, Cc, dtype=dy.dtype, device=dy.device)
    L.check(L.load().sa_reflect_pad_bwd(_f(dy), _f(dx), B, T2 - 2, F2 - 2, Cc, L.stream()), "sa_reflect_pad_bwd")
    return dx


def ln_leaky(x, gamma, beta, eps, slope, save):
    """leaky_relu(LayerNorm over the trailing gamma.numel() elements): [rows, d] bf16, d in {5120, 10240}"""
    d = gamma.numel()
    rows = x.numel() // d
    y = torch.empty_like(x)
    stat = torch.empty(rows, 2, dtype=torch.float32, device=x.device) if save else None
    L.check(L.load().sa_ln_leaky_fwd(_f(x), _f(gamma), _f(beta), _f(y), _f(stat), rows, d, C.c_float(eps),
                                     C.c_float(slope), L.stream()), "sa_ln_leaky_fwd")
    return y, stat


def ln_leaky_bwd(dy, x, stat, gamma, beta, slope):
    d = gamma.numel()
    dx = torch.empty_like(x)
    L.check(L.load().sa_ln_leaky_bwd(_f(dy), _f(x), _f(stat), _f(gamma), _f(beta), _f(dx), x.numel() // d, d,
                                     C.c_float(slope), L.stream()), "sa_ln_leaky_bwd")
    return dx


def asr_block0(x, w, bias, gamma, beta, eps, slope, save):
    """x [B, T, 80] bf16 -> leaky(LayerNorm(conv 1 -> 128, 3 x 3, stride 2, reflect "same")) [B, ceil(T/2), 40, 128]"""
    B, T, F_ = x.shape
    Cc = w.shape[0]
    To = (T - 1) // 2 + 1
    y = torch.empty(B, To, (F_ - 1) // 2 + 1, Cc, dtype=x.dtype, device=x.device)
    stat = torch.empty(B * To, 2, dtype=torch.float32, device=x.device) if save else None
    L.check(L.load().sa_asr_block0_fwd(_f(x), _f(w), _f(bias), _f(gamma), _f(beta), _f(y), _f(stat), B, T, F_, Cc,
                                       C.c_float(eps), C.c_float(slope), L.stream()), "sa_asr_block0_fwd")
    return y, stat


def asr_block0_bwd(dy, x, w, bias, gamma, beta, stat, slope):
    B, T, F_ = x.shape
    To = (T - 1) // 2 + 1
    part = torch.empty(B * To, 3, F_ + 2, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    L.check(L.load().sa_asr_block0_bwd(_f(dy), _f(x), _f(w), _f(bias), _f(gamma), _f(beta), _f(stat), _f(part), _f(dx),
                                       B, T, F_, w.shape[0], C.c_float(slope), L.stream()), "sa_asr_block0_bwd")
    return dx


# ---- per-XCD speed of the persistent kernels (csrc/sa_conv_ws.hip: sa_conv_ws_set_xcd_weights) ----
_xcd_weights = None
if __import__("os").environ.get("SA_XCD_WEIGHTS"):                      # experiment: "65,63,65,63,65,63,65,63"
    _xcd_weights = [int(v) for v in __import__("os").environ["SA_XCD_WEIGHTS"].split(",")]
    L.check(L.load().sa_conv_ws_set_xcd_weights((C.c_ubyte * 8)(*_xcd_weights)), "sa_conv_ws_set_xcd_weights")


def calibrate_xcd(device=None, rounds=2, B=16, Lin=20160, verbose=False):
    """Measure how fast each of the eight XCDs runs the persistent convolution kernels on THIS chip and hand the
    relative speeds to the tile-range balancing (workgroup i runs on XCD i % 8).  One synthetic fused
    data-gradient launch per round (128 -> 128, 5 taps; about 0.6 GB of scratch tensors), per-workgroup
    lifetimes read back (this synchronises: call it at set-up, not inside a step).  The weights change
    which workgroup computes which tiles, never a result.  Returns the eight weights (64 = nominal)."""
    global _xcd_weights
    lib = L.load()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Lin, 128, generator=g).to(dev)
    y2 = torch.randn(B, Lin, 128, generator=g).to(dev)
    xe = torch.randn(B, Lin, 128, generator=g).to(dev)
    w = (torch.randn(128, 128, 5, generator=g) * 0.05).to(dev)
    wd = pack_weights(w, "conv_dgrad", torch.float32, L.BF16X3)
    c = [(torch.rand(B, 128, generator=g) + 0.5).to(dev) for _ in range(3)]
    s1 = (torch.rand(B, 128, generator=g) + 0.5).to(dev)
    ao = torch.empty(B, Lin, 128, device=dev, dtype=torch.bfloat16)
    out = torch.empty(B, Lin, 128, device=dev)
    kw = dict(code=L.BF16X3, want_stats=True, a_out=ao, out=out,
              nb=dict(x=y2, c1=c[0], c2=c[1], c3=c[2], per_c=False, relu_mask=False, want_colsum=True),
              ep=dict(mode=1, x=xe, s1=s1, t1=s1, mean=s1, rstd=s1))
    weights = [64] * 8
    buf = (C.c_ulonglong * 1024)()
    for r in range(rounds + 1):
        L.check(lib.sa_conv_ws_set_xcd_weights((C.c_ubyte * 8)(*weights)), "sa_conv_ws_set_xcd_weights")
        if r == rounds:
            break
        for _ in range(3):
            conv_gemm(x, wd, None, 128, 128, 1, 1, taps_conv_dgrad_s1(5, 1, 2), Lin, **kw)
        torch.cuda.synchronize(dev)
        L.check(lib.sa_conv_ws_calibrate_read(buf), "sa_conv_ws_calibrate_read")
        life = [[] for _ in range(8)]
        for i in range(512):
            if buf[2 * i] and buf[2 * i + 1] > buf[2 * i]:
                life[i % 8].append(buf[2 * i + 1] - buf[2 * i])
        if any(len(v) < 4 for v in life):                   # (not the persistent route, or fewer workgroups than expected)
            break
        med = [sorted(v)[len(v) // 2] for v in life]
        # a workgroup's lifetime ~ its share / the speed of its XCD: new speed estimate = share / lifetime
        speed = [weights[xi] / med[xi] for xi in range(8)]
        mean = sum(speed) / 8.0
        weights = [max(32, min(128, int(round(64.0 * sp / mean)))) for sp in speed]
        if verbose:
            print(f"[calibrate_xcd] round {r}: median lifetime by XCD (us) {[round(m * 0.01, 1) for m in med]} -> weights {weights}",
                  flush=True)
    _xcd_weights = weights
    return weights


def clip_flats(flats, max_norm, eps=1e-6):
    """torch.nn.utils.clip_grad_norm_ on flat fp32 gradient buffers, in place (sa_clip_grads); returns the norm"""
    f = L.SaFlats()
    f.n = len(flats)
    for d, t in zip(f.f, flats):
        d.p, d.n = t.data_ptr(), t.numel()
    dev = flats[0].device
    partials = torch.empty(L.FLATS_MAX * 64, dtype=torch.float64, device=dev)
    total = torch.empty((), dtype=torch.float32, device=dev)
    L.check(L.load().sa_clip_grads(C.byref(f), C.c_float(max_norm), C.c_float(eps), _f(partials), _f(total), L.stream()),
            "sa_clip_grads")
    return total
