"""ctypes binding of libptpp_hip.so (the C ABI declared in include/ptpp.h).

The library is built in-tree by ``promptttspp_amd/csrc/Makefile`` (see
``__graft_entry__.build``).  There is NO fallback: if the shared object is
missing or a call fails, an exception is raised -- the product path never
silently routes around the HIP kernels.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libptpp_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU, ACT_SWISH, ACT_TANH, ACT_MISH = 0, 1, 2, 3, 4, 5


class PtppError(RuntimeError):
    pass


class ConvArgs(Structure):
    """Mirror of ``ptpp_conv1d_args`` (include/ptpp.h)."""

    _fields_ = [
        ("x", c_void_p),
        ("wp", c_void_p),
        ("bias", c_void_p),
        ("res", c_void_p),
        ("y", c_void_p),
        ("lengths", c_void_p),
        ("B", c_int32),
        ("T", c_int32),
        ("Cin", c_int32),
        ("Cout", c_int32),
        ("ks", c_int32),
        ("dil", c_int32),
        ("pad", c_int32),
        ("ldx", c_int32),
        ("ldy", c_int32),
        ("ldr", c_int32),
        ("act", c_int32),
        ("in_mask", c_int32),
        ("out_mask", c_int32),
        ("out_scale", c_float),
        ("dtype", c_int32),
    ]


# name -> (restype, argtypes); every symbol include/ptpp.h declares.
SIGNATURES = {
    "ptpp_last_error": (ctypes.c_char_p, []),
    "ptpp_version": (c_int, []),
    "ptpp_conv_cin_padded": (c_int, [c_int, c_int]),
    "ptpp_pack_conv_weight": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ptpp_conv1d_fwd": (c_int, [POINTER(ConvArgs), c_void_p]),
    "ptpp_conv1d_fwd_ex": (c_int, [POINTER(ConvArgs), c_void_p, c_int, c_float, c_void_p]),
    "ptpp_conv1d_wgrad": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 11 + [c_void_p],
    ),
    "ptpp_layernorm_fwd": (
        c_int,
        [c_void_p] * 9 + [c_int, c_int, c_int, c_float, c_int, c_int, c_void_p],
    ),
    "ptpp_layernorm_bwd": (c_int, [c_void_p] * 9 + [c_int] * 5 + [c_void_p]),
    "ptpp_aa_snake_fwd": (
        c_int,
        [c_void_p, c_void_p, c_void_p, POINTER(c_float), POINTER(c_float), c_int, c_int, c_int, c_int, c_void_p],
    ),
    "ptpp_add3_scale": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int, c_void_p]),
    "ptpp_conv_post_tanh": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ptpp_bct_to_btc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ptpp_btc_to_bct": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
}

_lib = None


def load():
    """Load the shared library (once) and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PtppError(
            f"{LIB_PATH} not found: build it with `make -C promptttspp_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "promptttspp_amd has no CPU / eager fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().ptpp_last_error().decode("utf-8", "replace")
        raise PtppError(f"{what} failed with status {status}: {msg}")
