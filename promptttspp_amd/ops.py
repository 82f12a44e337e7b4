"""Thin torch-tensor wrappers over the C ABI (include/ptpp.h).

Every function here launches hand-written HIP kernels from libptpp_hip.so on
torch's current stream.  Tensors must live on a ROCm device; nothing here has a
CPU implementation (the CPU restatement lives in ``oracle/`` and is test-only).

Layout: activations are channels-last ``(B, T, C)`` contiguous tensors in the
compute dtype (torch.float32 or torch.bfloat16).
"""
import ctypes

import torch

from . import _lib
from ._lib import ACT_NONE, BF16, F32, ConvArgs, check

_ACT = {
    None: 0,
    "none": 0,
    "relu": 1,
    "gelu": 2,
    "swish": 3,
    "tanh": 4,
    "mish": 5,
}


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise TypeError(f"promptttspp_amd: unsupported compute dtype {dt}")


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _need_gpu(t):
    if not t.is_cuda:
        raise _lib.PtppError(
            "promptttspp_amd ops run only on a ROCm device (got a CPU tensor); "
            "there is no CPU fallback in the product path"
        )


def _i32(lengths, device):
    if lengths is None:
        return None
    if lengths.dtype != torch.int32 or lengths.device != device:
        lengths = lengths.to(device=device, dtype=torch.int32)
    return lengths.contiguous()


# ----------------------------------------------------------------------------
# weight packing
# ----------------------------------------------------------------------------
def cin_padded(cin, dtype):
    return _lib.load().ptpp_conv_cin_padded(int(cin), dtype_code(dtype))


def pack_conv_weight(w, dtype, mode=0):
    """``w``: (Cout, Cin, ks) f32 (nn.Conv1d layout; nn.Linear weights are
    viewed as ks=1).  Returns the packed K-contiguous operand in ``dtype``:
    mode 0 -> (Cout, ks, CinP) forward operand, mode 1 -> (Cin, ks, CoutP)
    data-gradient operand (taps flipped)."""
    _need_gpu(w)
    if w.dim() == 2:
        w = w.unsqueeze(-1)
    w = w.detach().contiguous().float()
    cout, cin, ks = w.shape
    rows, inner = (cout, cin) if mode == 0 else (cin, cout)
    innerp = cin_padded(inner, dtype)
    wp = torch.empty((rows, ks, innerp), device=w.device, dtype=dtype)
    check(
        _lib.load().ptpp_pack_conv_weight(_ptr(w), _ptr(wp), cout, cin, ks, mode, dtype_code(dtype), _stream()),
        "ptpp_pack_conv_weight",
    )
    return wp


# ----------------------------------------------------------------------------
# conv1d / linear
# ----------------------------------------------------------------------------
def conv1d(
    x,
    wp,
    bias,
    cout,
    ks=1,
    dil=1,
    pad=0,
    act=None,
    lengths=None,
    in_mask=False,
    out_mask=False,
    res=None,
    out_scale=1.0,
    res2=None,
    res_scale=1.0,
    out=None,
):
    """Channels-last conv / linear with the fused epilogue (see ptpp.h).

    x: (B, T, Cin) ; wp: packed weight from :func:`pack_conv_weight` ;
    bias: (cout) f32 or None.  Returns y: (B, T, cout) in x.dtype.
    """
    _need_gpu(x)
    assert x.dim() == 3 and x.stride(2) == 1 and x.stride(0) == x.shape[1] * x.stride(1), "x must be (B,T,C) rows"
    B, T, cin = x.shape
    y = out if out is not None else torch.empty((B, T, cout), device=x.device, dtype=x.dtype)
    lengths = _i32(lengths, x.device)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.is_contiguous()
    a = ConvArgs()
    a.x, a.wp, a.bias, a.res, a.y = x.data_ptr(), wp.data_ptr(), (bias.data_ptr() if bias is not None else None), (
        res.data_ptr() if res is not None else None
    ), y.data_ptr()
    a.lengths = lengths.data_ptr() if lengths is not None else None
    a.B, a.T, a.Cin, a.Cout, a.ks, a.dil, a.pad = B, T, cin, cout, ks, dil, pad
    a.ldx, a.ldy = x.stride(1), y.stride(1)
    a.ldr = res.stride(1) if res is not None else 0
    a.act = _ACT[act]
    a.in_mask, a.out_mask = int(bool(in_mask)), int(bool(out_mask))
    a.out_scale = float(out_scale)
    a.dtype = dtype_code(x.dtype)
    if res is not None:
        assert res.dtype == x.dtype and res.shape == y.shape
    if res2 is not None:
        assert res2.dtype == x.dtype and res2.shape == y.shape
    lib = _lib.load()
    if res2 is None and res_scale == 1.0:
        check(lib.ptpp_conv1d_fwd(ctypes.byref(a), _stream()), "ptpp_conv1d_fwd")
    else:
        check(
            lib.ptpp_conv1d_fwd_ex(
                ctypes.byref(a), _ptr(res2), res2.stride(1) if res2 is not None else 0, float(res_scale), _stream()
            ),
            "ptpp_conv1d_fwd_ex",
        )
    return y


def conv1d_wgrad(x, dy, cin, cout, ks, dil, pad, lengths=None, in_mask=False, want_bias=True):
    """Returns (dw (Cout,Cin,ks) f32, dbias (Cout) f32 or None)."""
    _need_gpu(x)
    B, T, _ = x.shape
    dw = torch.zeros((cout, cin, ks), device=x.device, dtype=torch.float32)
    db = torch.zeros((cout,), device=x.device, dtype=torch.float32) if want_bias else None
    lengths = _i32(lengths, x.device)
    check(
        _lib.load().ptpp_conv1d_wgrad(
            _ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(lengths), B, T, cin, cout, ks, dil, pad,
            x.stride(1), dy.stride(1), int(bool(in_mask)), dtype_code(x.dtype), _stream(),
        ),
        "ptpp_conv1d_wgrad",
    )
    return dw, db


# ----------------------------------------------------------------------------
# layer norm
# ----------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps, res=None, lengths=None, out_mask=False, save_stats=False, save_sum=False):
    _need_gpu(x)
    assert x.is_contiguous() and x.dim() == 3
    B, T, C = x.shape
    y = torch.empty_like(x)
    mean = rstd = xsum = None
    if save_stats:
        mean = torch.empty((B * T,), device=x.device, dtype=torch.float32)
        rstd = torch.empty((B * T,), device=x.device, dtype=torch.float32)
    if save_sum and res is not None:
        xsum = torch.empty_like(x)
    lengths = _i32(lengths, x.device)
    check(
        _lib.load().ptpp_layernorm_fwd(
            _ptr(x), _ptr(res), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(xsum), _ptr(mean), _ptr(rstd), _ptr(lengths),
            B, T, C, float(eps), int(bool(out_mask)), dtype_code(x.dtype), _stream(),
        ),
        "ptpp_layernorm_fwd",
    )
    return y, mean, rstd, xsum


def layernorm_bwd(dy, xsum, gamma, mean, rstd, lengths=None, out_mask=False):
    _need_gpu(dy)
    B, T, C = dy.shape
    dx = torch.empty_like(dy)
    dgamma = torch.zeros((C,), device=dy.device, dtype=torch.float32)
    dbeta = torch.zeros((C,), device=dy.device, dtype=torch.float32)
    lengths = _i32(lengths, dy.device)
    check(
        _lib.load().ptpp_layernorm_bwd(
            _ptr(dy.contiguous()), _ptr(xsum), _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(dgamma),
            _ptr(dbeta), _ptr(lengths), B, T, C, int(bool(out_mask)), dtype_code(dy.dtype), _stream(),
        ),
        "ptpp_layernorm_bwd",
    )
    return dx, dgamma, dbeta


# ----------------------------------------------------------------------------
# anti-aliased snake, misc
# ----------------------------------------------------------------------------
def _taps(filt):
    arr = (ctypes.c_float * 12)()
    vals = filt.detach().reshape(-1).float().cpu().tolist() if isinstance(filt, torch.Tensor) else list(filt)
    assert len(vals) == 12, "anti-alias filters must have 12 taps"
    for i, v in enumerate(vals):
        arr[i] = v
    return arr


def aa_snake(x, log_alpha, taps_up, taps_down, out=None):
    """x: (B,T,C) ; log_alpha: (C) f32 device tensor ; taps_*: ctypes float[12]
    (see :func:`_taps`)."""
    _need_gpu(x)
    assert x.is_contiguous()
    B, T, C = x.shape
    y = out if out is not None else torch.empty_like(x)
    check(
        _lib.load().ptpp_aa_snake_fwd(
            _ptr(x), _ptr(y), _ptr(log_alpha), taps_up, taps_down, B, T, C, dtype_code(x.dtype), _stream()
        ),
        "ptpp_aa_snake_fwd",
    )
    return y


def add3_scale(a, b, c, scale):
    _need_gpu(a)
    y = torch.empty_like(a)
    check(
        _lib.load().ptpp_add3_scale(_ptr(a), _ptr(b), _ptr(c), _ptr(y), float(scale), a.numel(), dtype_code(a.dtype), _stream()),
        "ptpp_add3_scale",
    )
    return y


def conv_post_tanh(x, w, bias):
    """x: (B,T,C) ; w: (ks, C) f32 ; returns (B, T) f32."""
    _need_gpu(x)
    B, T, C = x.shape
    y = torch.empty((B, T), device=x.device, dtype=torch.float32)
    check(
        _lib.load().ptpp_conv_post_tanh(
            _ptr(x), _ptr(w), float(bias), _ptr(y), B, T, C, w.shape[0], dtype_code(x.dtype), _stream()
        ),
        "ptpp_conv_post_tanh",
    )
    return y


def bct_to_btc(x, dtype):
    """(B, C, T) f32 -> (B, T, C) ``dtype`` (the boundary transpose)."""
    _need_gpu(x)
    x = x.contiguous().float()
    B, C, T = x.shape
    y = torch.empty((B, T, C), device=x.device, dtype=dtype)
    check(_lib.load().ptpp_bct_to_btc(_ptr(x), _ptr(y), B, C, T, dtype_code(dtype), _stream()), "ptpp_bct_to_btc")
    return y


def btc_to_bct(x):
    """(B, T, C) compute dtype -> (B, C, T) f32."""
    _need_gpu(x)
    assert x.is_contiguous()
    B, T, C = x.shape
    y = torch.empty((B, C, T), device=x.device, dtype=torch.float32)
    check(_lib.load().ptpp_btc_to_bct(_ptr(x), _ptr(y), B, T, C, dtype_code(x.dtype), _stream()), "ptpp_btc_to_bct")
    return y
