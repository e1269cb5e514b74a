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

# name -> argtypes  (every entry point returns int: 0 ok, <0 error with ly_last_error())
SIGNATURES = {
    "ly_abi_version": [],
    "ly_mlpblock_fwd": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "ly_mlpblock_pack_sizes": [_I, _LP, _LP, _LP],
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
