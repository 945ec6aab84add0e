"""ctypes binding of libsa_hip.so (the C ABI declared in include/sa_hip.h).

The product path has NO CPU fallback: if the library is missing, or a call returns a
non-zero code, this module raises.  torch is used only for device memory and streams.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SA_HIP_LIB points the binding at another build of the same ABI (kernel A/B experiments)
LIB_PATH = os.environ.get("SA_HIP_LIB") or os.path.join(_HERE, "libsa_hip.so")

F32, BF16, BF16X3, BF16X1F, FP8, F64 = 0, 1, 2, 3, 4, 5
MAX_TAPS = 5

c_fp = C.POINTER(C.c_float)
vp = C.c_void_p


class SaTaps(C.Structure):
    _fields_ = [("ntaps", C.c_int * 2), ("off", (C.c_int * MAX_TAPS) * 2),
                ("widx", (C.c_int * MAX_TAPS) * 2)]


class SaConvArgs(C.Structure):
    _fields_ = [("x", vp), ("wp", vp), ("bias", vp), ("y", vp),
                ("s1", vp), ("t1", vp), ("s2", vp), ("t2", vp),
                ("swish", C.c_int), ("relu", C.c_int), ("stats", vp),
                ("B", C.c_int), ("Lin", C.c_int), ("Lout", C.c_int), ("ntiles", C.c_int),
                ("rowmin", C.c_int), ("nrows", C.c_int), ("wlo_off", C.c_int), ("taps", SaTaps),
                ("ep_mode", C.c_int), ("ep_xp_is_act", C.c_int), ("ep_bstride", C.c_int),
                ("ep_x", vp), ("ep_g2", vp), ("ep_s1", vp), ("ep_t1", vp), ("ep_mean", vp), ("ep_rstd", vp),
                ("a_out", vp),
                ("nb_x", vp), ("nb_c1", vp), ("nb_c2", vp), ("nb_c3", vp),
                ("nb_bstride", C.c_int), ("nb_relu_mask", C.c_int), ("nb_colsum", vp),
                ("ep_g2k1", vp), ("ep_g2k2", vp), ("ep_g2k3", vp), ("pro_stats", vp), ("wscale", vp),
                ("tile_rows", C.c_int), ("pad2_", C.c_int)]


class SaPackDesc(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("dtype", C.c_int), ("ntaps", C.c_int), ("K", C.c_int),
                ("N", C.c_int), ("sk", C.c_int), ("sn", C.c_int), ("st", C.c_int), ("pad_", C.c_int),
                ("scale", vp)]


class SaWgradArgs(C.Structure):
    _fields_ = [("x", vp), ("dy", vp), ("slabs", vp),
                ("s1", vp), ("t1", vp), ("s2", vp), ("t2", vp), ("swish", C.c_int),
                ("B", C.c_int), ("Lin", C.c_int), ("Ldy", C.c_int), ("Mrows", C.c_int),
                ("chunk", C.c_int), ("nchunk", C.c_int),
                ("ntaps", C.c_int), ("off", C.c_int * MAX_TAPS), ("ph", C.c_int * MAX_TAPS),
                ("x_pre", C.c_int), ("dy_pre", C.c_int)]


class SaEwArgs(C.Structure):
    _fields_ = [("g", vp), ("g2", vp), ("x", vp), ("out", vp),
                ("s1", vp), ("t1", vp), ("mean", vp), ("rstd", vp),
                ("c1", vp), ("c2", vp), ("c3", vp),
                ("actbwd", C.c_int), ("xp_is_act", C.c_int), ("relu_mask", C.c_int),
                ("bstride", C.c_int), ("stats", vp),
                ("B", C.c_int), ("L", C.c_int), ("ntiles", C.c_int)]


class SaFinArgs(C.Structure):
    _fields_ = [("part", vp), ("rows", vp), ("tickets", vp),
                ("nbatch", C.c_int), ("nslab", C.c_int), ("n", C.c_int), ("C", C.c_int), ("ncomp", C.c_int),
                ("mode", C.c_int), ("count", C.c_double),
                ("eps", C.c_float), ("momentum", C.c_float), ("sign", C.c_float), ("pad_", C.c_float),
                ("gamma", vp), ("beta", vp), ("mean", vp), ("rstd", vp),
                ("o0", vp), ("o1", vp), ("o2", vp), ("o3", vp),
                ("dgamma", vp), ("dbeta", vp), ("db", vp), ("run_mean", vp), ("run_var", vp)]


FIN_IN_FWD, FIN_IN_BWD, FIN_BN_FWD, FIN_BN_BWD, FIN_BIAS = 1, 2, 3, 4, 5
BIAS_MAX = 8


class SaBiasDesc(C.Structure):
    _fields_ = [("part", vp), ("rows", vp), ("db", vp),
                ("nbatch", C.c_int), ("nslab", C.c_int), ("C", C.c_int), ("ncomp", C.c_int)]


class SaBiasMulti(C.Structure):
    _fields_ = [("n", C.c_int), ("pad_", C.c_int), ("d", SaBiasDesc * BIAS_MAX)]


WRED_MAX = 8
FLATS_MAX = 4


class SaFlat(C.Structure):
    _fields_ = [("p", vp), ("n", C.c_longlong)]


class SaFlats(C.Structure):
    _fields_ = [("n", C.c_int), ("pad_", C.c_int), ("f", SaFlat * FLATS_MAX)]



class SaWredDesc(C.Structure):
    _fields_ = [("slabs", vp), ("dst", vp)] + [(k, C.c_int) for k in ("nslab", "ntaps", "cin", "cout", "sk", "sn", "st",
                                                                      "accumulate", "vec", "pad_")]


class SaWredMulti(C.Structure):
    _fields_ = [("n", C.c_int), ("pad_", C.c_int), ("d", SaWredDesc * WRED_MAX)]


# every symbol include/sa_hip.h declares (checked by tests/test_abi.py on CPU)
SYMBOLS = [
    "sa_conv_gemm", "sa_abi_sizeof", "sa_conv_gemm_ntiles", "sa_conv_gemm_ntiles_tm", "sa_conv_gemm_set_tile_rows",
    "sa_conv_gemm_geometry", "sa_conv_gemm_set_impl", "sa_conv_gemm_route", "sa_conv_pp_set_tile_rows", "sa_pack_weights", "sa_pack_weights_multi", "sa_pack_scales_multi", "sa_wgrad", "sa_wgrad_kw", "sa_wgrad_reduce",
    "sa_conv1toC", "sa_conv1toC_ntiles", "sa_convCto1", "sa_wgrad1C", "sa_wgrad1C_nchunk",
    "sa_sum_slabs", "sa_ew_stats", "sa_ew_apply", "sa_ew_ntiles", "sa_act_stats",
    "sa_sum_partials", "sa_sum_rows_d", "sa_reduce_finalize", "sa_fin_in_fwd", "sa_fin_bn_fwd", "sa_fin_bn_eval", "sa_fin_norm_bwd", "sa_fin_bias",
    "sa_pool_fwd", "sa_pool_nseg", "sa_pool_gather", "sa_pool_fin", "sa_pool_bwd", "sa_dense", "sa_colsums",
    "sa_bn2d_bwd", "sa_dense_wgrad", "sa_log_softmax", "sa_log_softmax_bwd",
    "sa_loss_workspace_bytes", "sa_recon_loss", "sa_cls_losses", "sa_cosine_loss",
    "sa_tdnn_fwd", "sa_time_pool", "sa_leaky_affine", "sa_tdnn_bwd_input", "sa_tdnn_fold", "sa_time_pool_bwd",
    "sa_leaky_affine_bwd", "sa_cluster_mi", "sa_fbank", "sa_fbank_table_elems", "sa_fbank_ntiles", "sa_fbank_scratch_bytes", "sa_fbank_normalize",
    "sa_comm_unique_id", "sa_comm_init", "sa_comm_world", "sa_comm_allreduce", "sa_comm_allreduce_inline", "sa_comm_join", "sa_comm_ncalls",
    "sa_comm_destroy", "sa_head_fwd", "sa_head_bwd", "sa_head_max_rows", "sa_conv_ws_set_bcost", "sa_conv_wsd_set_bcost",
    "sa_add_layernorm_fwd", "sa_layernorm_bwd", "sa_reflect_pad_fwd", "sa_reflect_pad_bwd", "sa_ln_leaky_fwd", "sa_ln_leaky_bwd", "sa_bias_multi", "sa_asr_block0_fwd", "sa_asr_block0_bwd", "sa_conv_ws_set_xcd_weights", "sa_conv_ws_calibrate_read", "sa_wgrad_reduce_multi", "sa_clip_grads",
]

_lib = None


class SaHipError(RuntimeError):
    pass


def load():
    """Load libsa_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SaHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        for s in SYMBOLS:
            getattr(_lib, s).restype = C.c_int
        for i, rec in enumerate((SaConvArgs, SaWgradArgs, SaEwArgs, SaPackDesc, SaTaps, SaFinArgs, SaBiasMulti, SaWredMulti)):
            if _lib.sa_abi_sizeof(i) != C.sizeof(rec):
                raise SaHipError(f"{rec.__name__}: binding has {C.sizeof(rec)} bytes, {LIB_PATH} "
                                 f"{_lib.sa_abi_sizeof(i)} -- rebuild the library (stale build?)")
    return _lib


def stream():
    """raw handle of torch's current stream on the current device (the C accessor: the Python
    torch.cuda.current_stream() wrapper costs ~9 us, 50 times per step)"""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def ptr(t):
    """device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "HIP kernels take contiguous device tensors"
    return C.c_void_p(t.data_ptr())


def check(rc, what):
    if rc != 0:
        raise SaHipError(f"{what} failed with code {rc}")


def dt_code(dtype):
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise SaHipError(f"unsupported activation dtype {dtype}")


_taps_cache = {}


def make_taps(phases):
    """phases: list (len U) of lists of (row_offset, weight_index).  The record is immutable for
    the callers (it is copied into SaConvArgs by assignment), so one instance per tap table."""
    key = tuple(tuple(p) for p in phases)
    t = _taps_cache.get(key)
    if t is not None:
        return t
    t = _taps_cache[key] = SaTaps()
    for ph, lst in enumerate(phases):
        t.ntaps[ph] = len(lst)
        for i, (off, wi) in enumerate(lst):
            t.off[ph][i] = off
            t.widx[ph][i] = wi
    return t
