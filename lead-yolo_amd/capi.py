"""ctypes binding of the C-ABI in include/lead_yolo_hip.h (libleadyolo_hip.so, built in-tree from
lead-yolo_amd/csrc).  There is NO fallback: if the shared library is missing or a call fails the
product raises — the oracle / CPU code is never substituted."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libleadyolo_hip.so")

_lib = None

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_long
_F = ctypes.c_float
_LP = ctypes.POINTER(ctypes.c_long)



class LyGemmParams(ctypes.Structure):       # mirrors include/lead_yolo_hip.h
    _fields_ = [("M", _L), ("H", _I), ("W", _I), ("K", _I), ("N", _I),
                ("a0", _P), ("lda0", _I), ("k0", _I), ("a1", _P), ("lda1", _I),
                ("gather", _I), ("Hin", _I), ("Win", _I), ("Cin", _I), ("ks", _I), ("pk", _I),
                ("pro", _I), ("g_h", _P), ("g_w", _P), ("res", _P), ("ldres", _I),
                ("p_scale", _P), ("p_shift", _P), ("p_ca", _P),
                ("wp", _P), ("e_scale", _P), ("e_shift", _P), ("rowscale", _P), ("act", _I),
                ("out", _P), ("ldo", _I), ("stats", _P), ("dtype", _I), ("scat_ks", _I), ("scat_c", _I), ("eadd", _P), ("ldeadd", _I)]


class LyConv3Params(ctypes.Structure):
    _fields_ = [("M", _L), ("H", _I), ("W", _I), ("Cin", _I), ("N", _I), ("TH", _I), ("TW", _I), ("x", _P), ("ldx", _I), ("wp", _P),
                ("e_scale", _P), ("e_shift", _P), ("act", _I), ("out", _P), ("ldo", _I), ("stats", _P), ("dtype", _I)]


class LyRfcbam3Params(ctypes.Structure):
    _fields_ = [("n_img", _I), ("H", _I), ("W", _I), ("C", _I), ("Ho", _I), ("Wo", _I), ("N", _I), ("s", _I),
                ("TH", _I), ("TW", _I), ("x", _P), ("ldx", _I), ("wg", _P), ("ca", _P), ("rfa", _P), ("wp", _P),
                ("e_scale", _P), ("e_shift", _P), ("out", _P), ("ldo", _I), ("stats", _P), ("linear", _I), ("dtype", _I)]


class LyRf3cBwdParams(ctypes.Structure):
    _fields_ = [("n_img", _I), ("H", _I), ("W", _I), ("C", _I), ("Ho", _I), ("Wo", _I), ("O", _I), ("s", _I), ("TH", _I), ("TW", _I),
                ("x", _P), ("ldx", _I), ("du", _P), ("lddu", _I), ("wq", _P), ("wct", _P), ("ca", _P), ("rfa", _P), ("mm", _P), ("d_mm", _P),
                ("coef", _P), ("d_rfa_part", _P), ("d_ca", _P), ("sums", _P), ("dwg", _P), ("dx", _P), ("lddx", _I), ("dgap", _P),
                ("dgap_scale", _F), ("dwc_part", _P), ("ng", _I), ("dtype", _I)]


class LyRf1BwdParams(ctypes.Structure):
    _fields_ = [("n_img", _I), ("HW", _L), ("C", _I), ("x", _P), ("ldx", _I), ("dcd", _P), ("gw", _P), ("ag", _P), ("bg", _P), ("ca", _P), ("rfa", _P),
                ("cd", _P), ("d_rfa", _P), ("gmax_out", _P), ("d_ca", _P), ("gmax", _P), ("d_mm", _P), ("sums", _P),
                ("alpha", _P), ("kappa", _P), ("lambda_", _P), ("dgap", _P), ("dgap_scale", _F), ("dx", _P), ("lddx", _I), ("dgw", _P), ("dtype", _I),
                ("dgw_f64", _I)]


F64_ADD_MAX = 64


class LyF64AddTable(ctypes.Structure):
    _fields_ = [("src", _P * F64_ADD_MAX), ("dst", _P * F64_ADD_MAX), ("n", _I * F64_ADD_MAX), ("count", _I)]


class LyOptTensor(ctypes.Structure):
    _fields_ = [("p", _P), ("g", _P), ("buf", _P), ("ema", _P), ("n", _L), ("wd", _F), ("group", _I), ("taps", _I), ("cin", _I)]


class LyPackDesc(ctypes.Structure):
    _fields_ = [("src", _P), ("dst", _P), ("r_valid", _I), ("K", _I), ("planes", _I), ("S", _I), ("t0", _I), ("T", _I),
                ("nrb", _I), ("nb", _I), ("nc", _I), ("vb", _I), ("vc", _I),
                ("sra", _L), ("srb", _L), ("sa", _L), ("sb", _L), ("sc", _L), ("blk0", _L)]


class LyWgradParams(ctypes.Structure):
    _fields_ = [("M", _L), ("H", _I), ("W", _I), ("N", _I), ("du", _P), ("lddu", _I), ("x", _P), ("ldx", _I),
                ("Hin", _I), ("Win", _I), ("Cin", _I), ("ks", _I), ("stride", _I), ("pad", _I), ("nchw", _I), ("up2", _I),
                ("dw", _P), ("lddw", _I), ("dtype", _I), ("dw_ts", _I), ("dw_cs", _I), ("n_valid", _I), ("c_valid", _I),
                ("x_scale", _P), ("x_shift", _P), ("ws", _P), ("ws_floats", _L)]


STATS_STRIPES = 32
LY_F32, LY_BF16 = 0, 1                     # `dtype` codes of the C ABI


def dtype_code(t):
    """LY_F32 / LY_BF16 for a tensor (or torch dtype)"""
    import torch
    d = t if isinstance(t, torch.dtype) else t.dtype
    if d == torch.float32:
        return LY_F32
    if d == torch.bfloat16:
        return LY_BF16
    raise HipLibraryError(f"the HIP kernels are built for float32 and bfloat16 activations (got {d})")


ACT_NONE, ACT_RELU, ACT_SILU = 0, 1, 2
GATHER_ROWS, GATHER_UP2, GATHER_PATCH, GATHER_PATCH_NCHW, GATHER_PATCH_NCHW_U8, GATHER_PATCH_NCHW_BF16, GATHER_PATCH_NCHW_F16 = 0, 1, 2, 3, 4, 5, 6
PRO_NONE, PRO_GATE, PRO_AFFINE_RELU_CA = 0, 1, 2

# name -> argtypes  (every entry point returns int: 0 ok, <0 error with ly_last_error())
SIGNATURES = {
    "ly_abi_version": [],
    "ly_mlpblock_fwd": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P],
    "ly_mlpblock_hidden_tiles": [_I],
    "ly_mlpblock_bwd_ok": [_I, _I],
    "ly_mlpblock_bwd_dx_ok": [_I, _I, _I],
    "ly_mlpblock_bwd_dx": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _L, _P, _I, _I, _I, _I, _P],
    "ly_mlpblock_bwd_slab_floats": [_I],                 # (returns long: restype set in lib())
    "ly_mlpblock_bwd": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _I, _I, _P],
    "ly_gemm_fwd": [ctypes.POINTER(LyGemmParams), _P],
    "ly_conv3x3_fwd": [ctypes.POINTER(LyConv3Params), _P],
    "ly_pool_hw": [_P, _I, _I, _I, _I, _I, _P, _I, _P],
    "ly_coordatt_mlp": [_P, _I, _I, _I, _I, _I] + [_P] * 11,
    "ly_coordatt_mlp_bwd": [_P, _I, _I, _I, _I, _I] + [_P] * 22 + [_I, _P],
    "ly_coordatt_gate": [_P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _P],
    "ly_se_fwd": [_P, _I, _I, _I, _I, _P, _P, _I, _P, _I, _P, _I, _P],
    "ly_rfcbam_stats": [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _I, _I, _P],
    "ly_colsum": [_P, _I, _I, _I, _I, _P, _I, _I, _P],
    "ly_rfcbam_mid": [_P, _I, _I, _I, _P, _P, _I, _P, _I, _P, _I, _I, _P, _P, _P],
    "ly_rfa_map": [_P, _I, _I, _I, _P, _P, _P],
    "ly_rfcbam3_fwd": [ctypes.POINTER(LyRfcbam3Params), _P],
    "ly_chan_moments": [_P, _I, _L, _I, _P, _I, _P],
    "ly_rfcbam_tap_moments": [_P, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    "ly_rfcbam_gen_prepare": [_P, _I, _I, _P, _P, _P, _F, _F, ctypes.c_double, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "ly_rf3c_stats": [_P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _I, _I, _P],
    "ly_rf3c_fwd": [ctypes.POINTER(LyRfcbam3Params), _P, _I, _P],
    "ly_rf3m_stats": [_P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P, _I, _P],
    "ly_rf3m_fwd": [ctypes.POINTER(LyRfcbam3Params), _P],
    "ly_rf3c_bwd": [ctypes.POINTER(LyRf3cBwdParams), _I, _P],
    "ly_rf3c_wgrad": [ctypes.POINTER(LyRf3cBwdParams), _P],
    "ly_coordatt_conv1_stats": [_P, _L, _I, _I, _P, _P, _P, _P],
    "ly_sppf_pool": [_P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    "ly_bnact_fwd": [_P, _I, _L, _I, _P, _P, _I, _P, _I, _I, _P],
    "ly_bnact_bwd_reduce": [_P, _I, _P, _I, _L, _I, _P, _P, _I, _P, _I, _P],
    "ly_bnact_bwd_apply": [_P, _I, _P, _I, _L, _I, _P, _P, _I, _P, _P, _P, _P, _I, _I, _P],
    "ly_bnact_bwd_reduce_pair": [_P, _I, _P, _I, _I, _P, _I, _L, _I, _P, _P, _I, _P, _P, _I, _P],
    "ly_bnact_bwd_apply_pair": [_P, _I, _P, _I, _I, _P, _I, _L, _I, _P, _P, _I, _P, _P, _P, _P, _I, _I, _P],
    "ly_wgrad": [ctypes.POINTER(LyWgradParams), _P],
    "ly_wgrad_group": [_P, _I, _P],
    "ly_up2_bwd": [_P, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    "ly_unpatch": [_P, _I, _I, _I, _I, _I, _P, _I, _P],
    "ly_patch4_rows_u8": [_P, _I, _I, _I, _I, _P, _I, _P],
    "ly_patch4_wgrad_u8": [_P, _I, _I, _I, _I, _P, _I, _I, _F, _P, ctypes.c_long, _P, _I, _P],
    "ly_coordatt_gate_bwd": [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _I, _I, _I, _P],
    "ly_pool_hw_bwd": [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P],
    "ly_maxpool_bwd": [_P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    "ly_rf_generate": [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "ly_rf_bwd_attn": [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "ly_rfa_bwd": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _I, _P],
    "ly_rf_bwd_relu": [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "ly_rf_bwd_gen": [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "ly_rf_bwd_dx": [_I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _F, _I, _P],
    "ly_nms_candidates": [_P, _I, _I, _I, _F, ctypes.c_ulonglong, _P, _P, _P],
    "ly_nms_candidates_ml": [_P, _I, _I, _I, _F, ctypes.c_ulonglong, _P, _P, _P],
    "ly_nms_greedy": [_P, _P, _P, _I, _I, _F, _F, _I, _I, _P, _P, _P],
    "ly_maxpool_arg": [_P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    "ly_maxpool_gather": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    "ly_sppf_bwd": [_P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    "ly_mlpblock_pconv": [_P, _P, _I, _I, _I, _I, _P, _I, _P],
    "ly_mlp_dx": [_P, _P, _P, _I, _L, _I, _I, _P, _I, _P],
    "ly_se_bwd": [_P, _I, _I, _I, _I, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P],
    "ly_bn_finalize": [_P, _I, _I, _I, _I, _I, ctypes.c_double, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P],
    "ly_bn_bwd_coeffs": [_P, _I, _I, _I, ctypes.c_double, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _P],
    "ly_bn_finalize_pair": [_P, _I, _I, _I, ctypes.c_double, _P, _P, _F, _F, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P],
    "ly_bn_bwd_coeffs_pair": [_P, _P, _I, _I, _I, ctypes.c_double, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "ly_frag_pack3": [_P, _I, _I, _L, _L, _I, _I, _P, _P],
    "ly_loss_level": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _L, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _F, _P],
    "ly_loss_finish": [_P, _I, _P, _P, _F, _F, _F, _I, _I, _P, _P],
    "ly_detect_tail": [_P, _I, _I, _I, _I, _I, _I, _P, _F, _P, _P, _L, _L, _I, _P],
    "ly_detect_level": [_P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _I, _P, _F, _P, _P, _L, _L, _I, _P],
    "ly_detect_level_ok": [_I, _I, _I, _I],
    "ly_detect_head_bwd": [_P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _I, _P],
    "ly_f64_add": [ctypes.POINTER(LyF64AddTable), _P],
    "ly_pack_table": [_P, _P, _I, _P],
    "ly_optim_step": [_P, _P, _P, _I, _P, _P, _P, _P],
    "ly_sum_rows": [_P, _L, _L, _L, _P, _I, _P],
    "ly_sum_rows_f64": [_P, _I, _I, _P, _P],
    "ly_rf1_bwd": [ctypes.POINTER(LyRf1BwdParams), _I, _P],
    "ly_rf3s_bwd": [ctypes.POINTER(LyRf1BwdParams), _I, _I, _I, _P],
    "ly_tune_wgrad3": [_I],
    "ly_event_create": [ctypes.POINTER(_P)],
    "ly_event_destroy": [_P],
    "ly_event_record": [_P, _P],
    "ly_stream_wait_event": [_P, _P],
}


class HipLibraryError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C lead-yolo_amd/csrc`). There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        L.ly_last_error.restype = ctypes.c_char_p
        L.ly_last_error.argtypes = []
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)        # AttributeError if the symbol is not exported
            fn.restype = _I
            fn.argtypes = args
        L.ly_mlpblock_bwd_slab_floats.restype = ctypes.c_long
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise HipLibraryError(f"{what} failed (rc={rc}): {lib().ly_last_error().decode()}")


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
