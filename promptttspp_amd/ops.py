"""Thin torch-tensor wrappers over the C ABI (include/ptpp.h) -- no autograd here
(see functional.py).

Every function launches hand-written HIP kernels from libptpp_hip.so on torch's
current stream.  Tensors must live on a ROCm device; nothing here has a CPU
implementation (the CPU restatement lives in ``oracle/`` and is test-only).

Layout: activations are channels-last ``(B, T, C)`` contiguous tensors in the
compute dtype (torch.float32 or torch.bfloat16).
"""
import ctypes
import struct

import torch

from . import _lib
from ._lib import BF16, F16, F32, ConvArgs, check

_ACT = {None: 0, "none": 0, "relu": 1, "gelu": 2, "swish": 3, "tanh": 4, "mish": 5, "gate": 6}
_VARIANT = {"new": 0, "legacy": 1, "plain": 2}


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float16:  # the vocoder's inference kernels only (include/ptpp.h: PTPP_F16)
        return F16
    raise TypeError(f"promptttspp_amd: unsupported compute dtype {dt}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


_stream_override = None  # set by functional.wgrad_stream: launch on this raw stream instead of torch's current one


_stream_pinned = None  # set by pinned_stream(): the handle looked up once for a whole step


# Branch streams (models/.../model.py, PTPP_BRANCH_STREAMS, on by default): autograd runs part of the backward on other
# streams, so the per-step pinning of the stream handle is off unless they are disabled
_NO_PIN = __import__("os").environ.get("PTPP_BRANCH_STREAMS", "2") not in ("", "0", "off", "no")


class pinned_stream:
    """Look torch's current stream up ONCE for a region that does not switch streams (a training step: forward,
    backward -- autograd runs the backward nodes on the forward's stream -- and the optimizer) instead of once per
    kernel launch (~2000 launches per step, 0.17 us each way plus the ctypes object).  ``functional.wgrad_stream`` still
    redirects the weight-gradient launches inside the region.  Re-entrant.  NOT for regions that switch streams or
    capture a graph (the sampler's graph capture, the vocoder's per-block side streams): torch ops would follow the
    switch while this package's launches stayed on the pinned stream -- pinning is skipped while a capture is under way,
    ``unpinned()`` suspends it for such a region, and ``PTPP_CHECK_PINNED_STREAM=1`` asserts on every launch that the
    pinned handle still is torch's current stream."""

    def __enter__(self):
        global _stream_pinned
        self.prev = _stream_pinned
        if _stream_pinned is None and not _NO_PIN and not torch.cuda.is_current_stream_capturing():
            _stream_pinned = _stream()
        return self

    def __exit__(self, *exc):
        global _stream_pinned
        _stream_pinned = self.prev
        return False


class unpinned:
    """Suspend ``pinned_stream`` for a region that switches streams or captures a graph."""

    def __enter__(self):
        global _stream_pinned
        self.prev = _stream_pinned
        _stream_pinned = None
        return self

    def __exit__(self, *exc):
        global _stream_pinned
        _stream_pinned = self.prev
        return False


_CHECK_PINNED = bool(__import__("os").environ.get("PTPP_CHECK_PINNED_STREAM"))


def _stream():
    """torch's current HIP stream as a raw handle (the C calls: ~0.3 us instead of ~7 us for the
    torch.cuda.current_stream() object -- this runs once per kernel launch)."""
    if _stream_override is not None:
        return _stream_override
    if _stream_pinned is not None:
        if _CHECK_PINNED:
            assert _stream_pinned.value == torch.cuda.current_stream().cuda_stream, "pinned_stream: torch switched streams"
        return _stream_pinned
    if _raw_stream is not None and _cur_device is not None:
        return ctypes.c_void_p(_raw_stream(_cur_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    """Device address for a ``void*`` argument (ctypes converts a Python int / None itself)."""
    return t.data_ptr() if t is not None else None


def _need_gpu(t):
    if not t.is_cuda:
        raise _lib.PtppError(
            "promptttspp_amd ops run only on a ROCm device (got a CPU tensor); "
            "there is no CPU fallback in the product path"
        )


def i32(lengths, device):
    if lengths is None:
        return None
    if lengths.dtype != torch.int32 or lengths.device != device:
        lengths = lengths.to(device=device, dtype=torch.int32)
    return lengths.contiguous()


def _ld(t):
    """Row stride (elements) of a (B, T, C) tensor whose rows are C contiguous elements
    and whose batches are T consecutive rows; size-1 dims may carry arbitrary strides."""
    assert t.dim() == 3, "expected a (B, T, C) tensor"
    B, T, C = t.shape
    assert C == 1 or t.stride(2) == 1, "channels must be contiguous"
    if T > 1:
        ld = t.stride(1)
        assert B == 1 or t.stride(0) == T * ld, "batches must be T consecutive rows"
    else:
        ld = t.stride(0) if B > 1 else C
    return ld


def _rows3(x):
    _ld(x)


# ----------------------------------------------------------------------------
# weight packing
# ----------------------------------------------------------------------------
def cin_padded(cin, dtype):
    return _lib.load().ptpp_conv_cin_padded(int(cin), dtype_code(dtype))


def pack_conv_weight(w, dtype, mode=0, out=None):
    """``w``: (Cout, Cin, ks) f32 (nn.Conv1d layout; nn.Linear weights are
    viewed as ks=1).  Returns the packed K-contiguous operand in ``dtype``:
    mode 0 -> (Cout, ks, CinP) forward operand, mode 1 -> (Cin, ks, CoutP)
    data-gradient operand (taps flipped), mode 2 -> mode 0 with the rows of a (gate | filter) weight in the
    interleaved order of the fused DiffNet gate epilogue.  ``out``: a (rows, ks, innerP) slice to fill."""
    _need_gpu(w)
    if w.dim() == 2:
        w = w.unsqueeze(-1)
    w = w.detach().contiguous().float()
    cout, cin, ks = w.shape
    rows, inner = (cout, cin) if mode not in (1, 4) else (cin, cout)  # (modes 3 / 4: the stream forms of modes 0 / 1, same size)
    wp = out if out is not None else torch.empty((rows, ks, cin_padded(inner, dtype)), device=w.device, dtype=dtype)
    assert wp.is_contiguous() and wp.shape == (rows, ks, cin_padded(inner, dtype)) and wp.dtype == dtype
    check(
        _lib.load().ptpp_pack_conv_weight(_ptr(w), _ptr(wp), cout, cin, ks, mode, dtype_code(dtype), _stream()),
        "ptpp_pack_conv_weight",
    )
    return wp


def pack_conv_weights_batched(table, n_entries, block_map, total_blocks):
    check(_lib.load().ptpp_pack_conv_weights_batched(_ptr(table), int(n_entries), _ptr(block_map), int(total_blocks),
                                                     _stream()),
          "ptpp_pack_conv_weights_batched")


# ----------------------------------------------------------------------------
# conv1d / linear
# ----------------------------------------------------------------------------
# One reusable argument block (the C side copies it before returning), filled by ONE struct.pack_into
# instead of 21 ctypes attribute stores; the format mirrors ptpp_conv1d_args / _lib.ConvArgs field by field.
_CONV_FMT = struct.Struct("6Q13ifi")
assert _CONV_FMT.size <= ctypes.sizeof(ConvArgs) and ctypes.sizeof(ConvArgs) == 112
_conv_buf = bytearray(ctypes.sizeof(ConvArgs))
_conv_args = ConvArgs.from_buffer(_conv_buf)
_conv_args_ref = ctypes.byref(_conv_args)


def _ld_fast(t):
    return t.shape[2] if t.is_contiguous() else _ld(t)


CONV_RT = not __import__("os").environ.get("PTPP_NO_CONV_RT")  # the row-tile kernel for the frame-level 256-channel layers
CONV_RT_MIN_ROWS = 24576  # (below ~200 row tiles of 128 the chip is not filled by one-workgroup-per-CU blocks: the tile kernel wins, 55.6 vs 62.0 us at 32 x 540)
_rt_ok = {}


_rt_min_pushed = [None]


def conv_rt_min_rows():
    """The row count from which a frame-level launch goes to the row-tile kernel.  ONE rule for both layers: the C drivers
    (csrc/stacks.cpp::rt_takes) read PTPP_CONV_RT_MIN_ROWS, so this side reads it too (the module constant is the default and
    what tests patch together with the variable); a driver that is handed an operand stream it would not use refuses the call."""
    e = __import__("os").environ.get("PTPP_CONV_RT_MIN_ROWS")
    v = int(e) if e else CONV_RT_MIN_ROWS
    if v != _rt_min_pushed[0]:  # ONE source of truth: the C-side drivers take this side's threshold (ptpp_conv_rt_set_min_rows)
        _lib.load().ptpp_conv_rt_set_min_rows(v)
        _rt_min_pushed[0] = v
    return v


def conv1d_rt_ok(x, cout, ks, dil, act, res2=None, drop_p=0.0):
    """Whether ``conv1d`` would take the row-tile kernel (csrc/conv1d_rt.hip) for this launch when handed the operand stream
    (pack mode 3 / 4): bf16, 256 output channels, Cin % 64 == 0, ks >= 3, frame-level row counts, plain epilogue."""
    if not (CONV_RT and x.is_cuda and x.dtype == torch.bfloat16 and cout == 256 and res2 is None and drop_p == 0.0):
        return False
    if x.shape[0] * x.shape[1] < conv_rt_min_rows() or x.stride(2) != 1:
        return False
    key = (x.shape[2], ks, dil, act)
    ok = _rt_ok.get(key)
    if ok is None:
        ok = _rt_ok[key] = act in (None, "relu") and bool(_lib.load().ptpp_conv1d_rt_supported(x.shape[2], cout, ks, dil, _ACT[act], BF16))
    return ok


_rt_ex_ok = {}


def conv1d_rt_ex_ok(x, cout, ks, dil, act, res2=None):
    """Whether ``conv1d`` takes ptpp_conv1d_rt_fwd_ex for this launch when handed the operand stream: the Conformer blocks'
    phone-level feed-forward convs (bf16, k = 9, Cout % 256 == 0, Cin % 128 == 0, ReLU / none, dropout allowed, any row count)."""
    if not (CONV_RT and x.is_cuda and x.dtype == torch.bfloat16 and res2 is None and x.stride(2) == 1):
        return False
    key = (x.shape[2], cout, ks, dil, act)
    ok = _rt_ex_ok.get(key)
    if ok is None:
        ok = _rt_ex_ok[key] = act in (None, "relu") and bool(_lib.load().ptpp_conv1d_rt_ex_supported(x.shape[2], cout, ks, dil, _ACT[act], BF16))
    return ok


def conv1d_rt_ex_relu_bwd(x, wstream, saved, cout, ks, pad, lengths, drop_p=0.0):
    """ptpp_conv1d_rt_fwd_ex_relu_bwd: dz = [saved > 0, t < len] * bf16(conv(x)) / (1 - drop_p) with the conv result as an
    intermediate (tests; the product calls it from ptpp_conformer_block_bwd).  x: (B, T, Cin) bf16 -> dz (B, T, cout)."""
    B, T, cin = x.shape
    y = torch.empty((B, T, cout), device=x.device, dtype=x.dtype)
    dz = torch.empty_like(y)
    lengths = i32(lengths, x.device)
    _CONV_FMT.pack_into(_conv_buf, 0, x.data_ptr(), 0, 0, 0, y.data_ptr(), lengths.data_ptr(), B, T, cin, cout, ks, 1, pad, _ld_fast(x),
                        cout, 0, _ACT[None], 0, 0, 1.0, BF16)
    ws = workspace(x.device)
    check(_lib.load().ptpp_conv1d_rt_fwd_ex_relu_bwd(_conv_args_ref, wstream.data_ptr(), saved.data_ptr(), dz.data_ptr(), float(drop_p),
                                                     ws.data_ptr(), ws.numel(), _stream()), "ptpp_conv1d_rt_fwd_ex_relu_bwd")
    return dz


COLPART = __import__("os").environ.get("PTPP_DIFFNET_COLPART", "1") != "0"


def conv1d_rt_colpart_ok(cin, ks, dils, B, T):
    """Whether row-tile launches of this geometry (every dilation of ``dils``) can emit per-tile column sums of their output
    (ptpp_conv1d_rt_fwd_cs); PTPP_DIFFNET_COLPART=0 turns the only user -- the DiffNet backward -- back to the pass over gx."""
    lib = _lib.load()
    return COLPART and all(bool(lib.ptpp_conv1d_rt_colpart_supported(cin, ks, d, B, T)) for d in set(dils))


def conv1d(x, wp, bias, cout, ks=1, dil=1, pad=0, act=None, lengths=None, in_mask=False, out_mask=False, res=None,
           out_scale=1.0, res2=None, res_scale=1.0, drop_p=0.0, drop_seed=0, out=None, wstream=None, colpart=None):
    """Channels-last conv / linear with the fused epilogue (see ptpp.h).
    x: (B, T, Cin); wp: packed weight; bias: (cout) f32 or None -> (B, T, cout).  ``wstream``: the same weight in pack mode
    3 (forward) / 4 (data gradient): where ``conv1d_rt_ok`` holds the launch goes to the row-tile kernel (bit-identical).
    (Called ~170 times per training step: written for low host overhead.)"""
    if not x.is_cuda:
        _need_gpu(x)
    B, T, cin = x.shape
    y = out if out is not None else torch.empty((B, T, cout), device=x.device, dtype=x.dtype)
    if lengths is not None:
        lengths = i32(lengths, x.device)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.is_contiguous()
    ldr = ldr2 = 0
    if res is not None:
        assert res.dtype == x.dtype and res.shape[0] == B and res.shape[1] == T and res.shape[2] == cout
        ldr = _ld_fast(res)
    if res2 is not None:
        assert res2.dtype == x.dtype and res2.shape[0] == B and res2.shape[1] == T and res2.shape[2] == cout
        ldr2 = _ld_fast(res2)
    _CONV_FMT.pack_into(_conv_buf, 0, x.data_ptr(), wp.data_ptr() if wp is not None else 0, bias.data_ptr() if bias is not None else 0,
                        res.data_ptr() if res is not None else 0, y.data_ptr(),
                        lengths.data_ptr() if lengths is not None else 0, B, T, cin, cout, ks, dil, pad, _ld_fast(x),
                        _ld_fast(y), ldr, _ACT[act], 1 if in_mask else 0, 1 if out_mask else 0, out_scale,
                        BF16 if x.dtype == torch.bfloat16 else dtype_code(x.dtype))
    lib = _lib.load()
    if wstream is not None and conv1d_rt_ok(x, cout, ks, dil, act, res2, drop_p):
        if colpart is not None:  # (B, ceil(T / 32), 256) f32: per-tile column sums of y from the epilogue (conv1d_rt_colpart_ok)
            check(lib.ptpp_conv1d_rt_fwd_cs(_conv_args_ref, wstream.data_ptr(), float(res_scale), None, 0, 0.0, colpart.data_ptr(),
                                            _stream()), "ptpp_conv1d_rt_fwd_cs")
            return y
        check(lib.ptpp_conv1d_rt_fwd(_conv_args_ref, wstream.data_ptr(), float(res_scale), _stream()), "ptpp_conv1d_rt_fwd")
        return y
    assert colpart is None, "conv1d: the column-sum output exists on the row-tile kernel only"
    if wstream is not None and conv1d_rt_ex_ok(x, cout, ks, dil, act, res2):
        ws = None if torch.cuda.is_current_stream_capturing() else workspace(x.device)
        check(lib.ptpp_conv1d_rt_fwd_ex(_conv_args_ref, wstream.data_ptr(), float(res_scale), float(drop_p), int(drop_seed),
                                        _ptr(ws), ws.numel() if ws is not None else 0, _stream()), "ptpp_conv1d_rt_fwd_ex")
        return y
    assert wp is not None, "conv1d: an operand stream was passed for a launch no row-tile kernel takes"
    if T <= 512 and ks * cin >= 2048 and not torch.cuda.is_current_stream_capturing():
        # few rows per utterance and a long K: hand the kernel the per-stream scratch so it may split K
        ws = workspace(x.device)
        check(
            lib.ptpp_conv1d_fwd_ws(_conv_args_ref, _ptr(res2), ldr2, float(res_scale), float(drop_p), int(drop_seed),
                                   _ptr(ws), ws.numel(), _stream()),
            "ptpp_conv1d_fwd_ws",
        )
    elif res2 is None and res_scale == 1.0 and drop_p == 0.0:
        check(lib.ptpp_conv1d_fwd(_conv_args_ref, _stream()), "ptpp_conv1d_fwd")
    else:
        check(
            lib.ptpp_conv1d_fwd_ex(_conv_args_ref, _ptr(res2), ldr2, float(res_scale), float(drop_p), int(drop_seed),
                                   _stream()),
            "ptpp_conv1d_fwd_ex",
        )
    return y


_WS_BYTES = 64 << 20
_ws = {}
_ws_hot = {}  # raw stream handle -> workspace (fast path of conv1d_wgrad)


def workspace(device):
    """Per-device scratch for the deterministic split-K reduction of the weight gradients.
    One buffer is enough: its users are ordered on the stream."""
    # one buffer per (device, stream): its users are ordered on that stream
    idx = device.index
    key = (idx if idx is not None else torch.cuda.current_device(), _stream().value)
    w = _ws.get(key)
    if w is None:
        w = _ws[key] = torch.empty(_WS_BYTES, device=device, dtype=torch.uint8)
    return w


# Two auxiliary streams per device, shared by every part of the package that runs something beside the main stream (training
# branches, BigVGAN's parallel resblocks, the sampler's second half-batch): HIP multiplexes streams onto a few hardware queues
# in creation order, and with one private stream per user a process that had trained / run the vocoder before sampling put the
# sampler's second stream on the main stream's queue (sampler 137 -> 187 ms inside bench.py's full run).  main + weight-gradient
# side stream + these two = four streams in all.
_aux_streams = {}


def aux_stream(device, i):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, int(i) % 2)
    st = _aux_streams.get(key)
    if st is None:
        st = _aux_streams[key] = torch.cuda.Stream(device=device)
    return st


def workspace_of(device, stream_handle):
    """The scratch of ``workspace`` that belongs to the raw stream ``stream_handle`` (a ctypes.c_void_p)."""
    idx = device.index
    key = (idx if idx is not None else torch.cuda.current_device(), stream_handle.value)
    w = _ws.get(key)
    if w is None:
        w = _ws[key] = torch.empty(_WS_BYTES, device=device, dtype=torch.uint8)
    return w


_RED_BYTES = 32 * 4 * 4352  # 32 replicas of the widest sum: ptpp_linear_small_bwd's Cout * Cin + Cout = 4100 floats (LayerNorm: 2 * 1024)
_red = {}


def reduction_scratch(device):
    """The zero-filled scratch of the cross-block column sums (include/ptpp.h "Reduction scratch"),
    one per (device, stream).  Returns (pointer, bytes)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device(), _stream().value)
    w = _red.get(key)
    if w is None:
        w = _red[key] = torch.zeros(_RED_BYTES, device=device, dtype=torch.uint8)
    return _ptr(w), _RED_BYTES


def _acc_target(t, n):
    """An f32 contiguous accumulation target handed in by the caller (e.g. ``p.grad``)."""
    assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n
    return t


def conv1d_wgrad(x, dy, cin, cout, ks, dil, pad, lengths=None, in_mask=False, want_bias=True, dw_out=None,
                 db_out=None):
    """Returns (dw (Cout,Cin,ks) f32, dbias (Cout) f32 or None).  ``dw_out`` / ``db_out``: f32 buffers the
    kernel ACCUMULATES into (it adds with atomics) instead of fresh zero-filled ones."""
    if not x.is_cuda:
        _need_gpu(x)
    B, T, _ = x.shape
    dev = x.device
    dw = _acc_target(dw_out, cout * cin * ks) if dw_out is not None else \
        torch.zeros((cout, cin, ks), device=dev, dtype=torch.float32)
    db = None
    if want_bias:
        db = _acc_target(db_out, cout) if db_out is not None else torch.zeros((cout,), device=dev, dtype=torch.float32)
    if lengths is not None:
        lengths = i32(lengths, dev)
    st = _stream()
    ws = _ws_hot.get(st.value)  # (called ~110 times per backward: skip the (device, stream) key of workspace())
    if ws is None or ws.device != dev:
        ws = _ws_hot[st.value] = workspace(dev)
    check(
        _lib.load().ptpp_conv1d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else None,
                                      lengths.data_ptr() if lengths is not None else None, B, T, cin, cout, ks, dil, pad,
                                      _ld_fast(x), _ld_fast(dy), 1 if in_mask else 0,
                                      BF16 if x.dtype == torch.bfloat16 else dtype_code(x.dtype), ws.data_ptr(), _WS_BYTES, st),
        "ptpp_conv1d_wgrad",
    )
    return dw, db


def conv1d_wgrad_batched(problems, cin, cout, ks, lengths=None, in_mask=False):
    """Weight / bias gradients of several layers of ONE shape in one call (ptpp_conv1d_wgrad_batched).  ``problems``: list of
    (x (B,T,cin), dy (B,T,cout) view, dw f32 (cout,cin,ks), db f32 (cout) or None, dil, pad); every dw / db is ACCUMULATED
    into.  x / dy of all problems share their row strides."""
    x0, dy0 = problems[0][0], problems[0][1]
    B, T, _ = x0.shape
    dev = x0.device
    arr = (_lib.WgradProblem * len(problems))()
    for i, (x, dy, dw, db, dil, pad) in enumerate(problems):
        assert x.shape == x0.shape and dy.shape == dy0.shape and _ld_fast(x) == _ld_fast(x0) and _ld_fast(dy) == _ld_fast(dy0)
        assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == cout * cin * ks
        arr[i].x, arr[i].dy, arr[i].dw = x.data_ptr(), dy.data_ptr(), dw.data_ptr()
        arr[i].dbias = db.data_ptr() if db is not None else None
        arr[i].dil, arr[i].pad = int(dil), int(pad)
    if lengths is not None:
        lengths = i32(lengths, dev)
    ws = workspace(dev)
    check(_lib.load().ptpp_conv1d_wgrad_batched(arr, len(problems), lengths.data_ptr() if lengths is not None else None, B, T, cin,
                                                cout, ks, _ld_fast(x0), _ld_fast(dy0), 1 if in_mask else 0, dtype_code(x0.dtype),
                                                ws.data_ptr(), _WS_BYTES, _stream()), "ptpp_conv1d_wgrad_batched")


def conv1d_wgrad_grouped(problems):
    """Weight / bias gradients of several layers of DIFFERENT shapes in one call (ptpp_conv1d_wgrad_grouped).  ``problems``:
    list of (x (B,T,cin) view, dy (B,T,cout) view, dw f32 (cout,cin,ks), db f32 (cout) or None, ks, dil, pad, lengths or
    None (= the input mask)); every dw / db is ACCUMULATED into."""
    x0 = problems[0][0]
    dev = x0.device
    arr = (_lib.WgradGProblem * len(problems))()
    keep = []
    for i, (x, dy, dw, db, ks, dil, pad, lengths) in enumerate(problems):
        B, T, cin = x.shape
        cout = dy.shape[2]
        assert dy.shape[:2] == x.shape[:2] and dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == cout * cin * ks
        a = arr[i]
        a.x, a.dy, a.dw, a.dbias = x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else None
        if lengths is not None:
            lengths = i32(lengths, dev)
            keep.append(lengths)
            a.lengths = lengths.data_ptr()
        a.B, a.T, a.Cin, a.Cout, a.ks, a.dil, a.pad, a.ldx, a.lddy = B, T, cin, cout, int(ks), int(dil), int(pad), _ld_fast(x), _ld_fast(dy)
    ws = workspace(dev)
    check(_lib.load().ptpp_conv1d_wgrad_grouped(arr, len(problems), dtype_code(x0.dtype), ws.data_ptr(), _WS_BYTES, _stream()),
          "ptpp_conv1d_wgrad_grouped")


def epilogue_bwd(dy, y=None, lengths=None, scale=1.0, relu=False, out_mask=False, drop_p=0.0, seed=0):
    _need_gpu(dy)
    dy = dy.contiguous()
    B, T, C = dy.shape
    dz = torch.empty_like(dy)
    lengths = i32(lengths, dy.device)
    check(
        _lib.load().ptpp_epilogue_bwd(_ptr(dy), _ptr(y), _ptr(dz), _ptr(lengths), B, T, C, float(scale), int(relu),
                                      int(bool(out_mask)), float(drop_p), int(seed), dtype_code(dy.dtype), _stream()),
        "ptpp_epilogue_bwd",
    )
    return dz


# ----------------------------------------------------------------------------
# layer norm
# ----------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps, res=None, lengths=None, out_mask=False, save_stats=False, save_sum=False,
                  act_in=None, drop_in=(0.0, 0), drop_out=(0.0, 0)):
    _need_gpu(x)
    assert x.is_contiguous() and x.dim() == 3
    B, T, C = x.shape
    y = torch.empty_like(x)
    mean = rstd = xsum = None
    if save_stats:
        mean = torch.empty((B * T,), device=x.device, dtype=torch.float32)
        rstd = torch.empty((B * T,), device=x.device, dtype=torch.float32)
    if save_sum:
        xsum = torch.empty_like(x)
    lengths = i32(lengths, x.device)
    check(
        _lib.load().ptpp_layernorm_fwd(_ptr(x), _ptr(res), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(xsum), _ptr(mean),
                                       _ptr(rstd), _ptr(lengths), B, T, C, float(eps), int(bool(out_mask)), _ACT[act_in],
                                       float(drop_in[0]), int(drop_in[1]), float(drop_out[0]), int(drop_out[1]),
                                       dtype_code(x.dtype), _stream()),
        "ptpp_layernorm_fwd",
    )
    return y, mean, rstd, xsum


def layernorm_bwd(dy, xsum, gamma, mean, rstd, lengths=None, out_mask=False, z=None, act_in=None, drop_in=(0.0, 0),
                  drop_out=(0.0, 0), want_dz=False, dgamma_out=None, dbeta_out=None):
    """Returns (dsum, dz or None, dgamma, dbeta); ``*_out``: f32 buffers to accumulate into."""
    _need_gpu(dy)
    dy = dy.contiguous()
    B, T, C = dy.shape
    dsum = torch.empty_like(dy)
    dz = torch.empty_like(dy) if want_dz else None
    dgamma = _acc_target(dgamma_out, C) if dgamma_out is not None else \
        torch.zeros((C,), device=dy.device, dtype=torch.float32)
    dbeta = _acc_target(dbeta_out, C) if dbeta_out is not None else \
        torch.zeros((C,), device=dy.device, dtype=torch.float32)
    lengths = i32(lengths, dy.device)
    with red_immediate(dgamma_out is None or dbeta_out is None):  # (fresh buffers are read by autograd right away)
        check(
            _lib.load().ptpp_layernorm_bwd(_ptr(dy), _ptr(xsum), _ptr(z), _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dsum),
                                           _ptr(dz), _ptr(dgamma), _ptr(dbeta), _ptr(lengths), B, T, C, int(bool(out_mask)),
                                           _ACT[act_in], float(drop_in[0]), int(drop_in[1]), float(drop_out[0]),
                                           int(drop_out[1]), dtype_code(dy.dtype), *reduction_scratch(dy.device), _stream()),
            "ptpp_layernorm_bwd",
        )
    return dsum, dz, dgamma, dbeta


# ----------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------
def attention_fwd(q, k, v, pos, bias_u, bias_v, lengths, heads, variant, save_probs=False, drop_p=0.0, drop_seed=0):
    """q,k,v: (B,T,C) views with a common row stride (e.g. slices of a fused
    (B,T,3C) projection); pos: (L, C) or None; returns (ctx (B,T,C), probs)."""
    _need_gpu(q)
    B, T, C = q.shape
    assert _ld(k) == _ld(q) == _ld(v)
    dk = C // heads
    ctx = torch.empty((B, T, C), device=q.device, dtype=q.dtype)
    probs = torch.empty((B, heads, T, T), device=q.device, dtype=torch.float32) if save_probs else None
    lengths = i32(lengths, q.device)
    check(
        _lib.load().ptpp_attention_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(pos), _ptr(bias_u), _ptr(bias_v), _ptr(ctx),
                                       _ptr(probs), _ptr(lengths), B, T, heads, dk, _ld(q),
                                       pos.stride(0) if pos is not None else 0, C, _VARIANT[variant],
                                       float(drop_p), int(drop_seed), dtype_code(q.dtype), _stream()),
        "ptpp_attention_fwd",
    )
    return ctx, probs


def attention_bwd(q, k, v, pos, bias_u, bias_v, probs, dctx, lengths, heads, variant, dq, dk_, dv, du_out=None,
                  dvb_out=None, drop_p=0.0, drop_seed=0):
    """dq/dk_/dv: preallocated (B,T,C) views sharing a row stride (slices of a
    (B,T,3C) buffer).  Returns (dpos (L,C) f32 or None, du, dvb)."""
    _need_gpu(q)
    B, T, C = q.shape
    dkh = C // heads
    dctx = dctx.contiguous()
    dS = torch.empty((B, heads, T, T), device=q.device, dtype=torch.float32)
    dpos = du = dvb = None
    if variant in ("new", "legacy"):  # (the legacy table has T rows, the new one 2T - 1)
        dpos = torch.empty((2 * T - 1 if variant == "new" else T, C), device=q.device, dtype=torch.float32)
        du = _acc_target(du_out, C) if du_out is not None else torch.zeros((C,), device=q.device, dtype=torch.float32)
        dvb = _acc_target(dvb_out, C) if dvb_out is not None else \
            torch.zeros((C,), device=q.device, dtype=torch.float32)
    lengths = i32(lengths, q.device)
    with red_immediate(du_out is None or dvb_out is None):
        check(
            _lib.load().ptpp_attention_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(pos), _ptr(bias_u), _ptr(bias_v), _ptr(probs),
                                           _ptr(dctx), _ptr(dS), _ptr(dq), _ptr(dk_), _ptr(dv), _ptr(dpos), _ptr(du),
                                           _ptr(dvb), _ptr(lengths), B, T, heads, dkh, _ld(q),
                                           pos.stride(0) if pos is not None else 0, C, _ld(dq), _VARIANT[variant],
                                           float(drop_p), int(drop_seed), dtype_code(q.dtype),
                                           *reduction_scratch(q.device), _stream()),
            "ptpp_attention_bwd",
        )
    return dpos, du, dvb


def attention_win_fwd(q, k, v, emb_k, emb_v, lengths, heads, window, save_probs=False, drop_p=0.0, drop_seed=0):
    """Windowed relative-position attention core (ptpp_attention_win_fwd; reference modules/transformer.py:59-137): q, k, v
    (B,T,C) views with a common row stride, emb_k / emb_v (2w+1, C/heads) f32 -> (ctx (B,T,C), probs)."""
    _need_gpu(q)
    B, T, C = q.shape
    assert _ld(k) == _ld(q) == _ld(v)
    dk = C // heads
    assert emb_k.shape == emb_v.shape == (2 * window + 1, dk) and emb_k.dtype == emb_v.dtype == torch.float32
    assert emb_k.is_contiguous() and emb_v.is_contiguous()
    ctx = torch.empty((B, T, C), device=q.device, dtype=q.dtype)
    probs = torch.empty((B, heads, T, T), device=q.device, dtype=torch.float32) if save_probs else None
    lengths = i32(lengths, q.device)
    check(_lib.load().ptpp_attention_win_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(emb_k), _ptr(emb_v), _ptr(ctx), _ptr(probs), _ptr(lengths), B, T,
                                             heads, dk, _ld(q), C, int(window), float(drop_p), int(drop_seed), dtype_code(q.dtype),
                                             _stream()), "ptpp_attention_win_fwd")
    return ctx, probs


def attention_win_bwd(q, k, v, emb_k, emb_v, probs, dctx, lengths, heads, window, dq, dk_, dv, drop_p=0.0, drop_seed=0):
    """Backward of ``attention_win_fwd``: fills the preallocated dq / dk_ / dv views, returns (demb_k, demb_v) f32."""
    _need_gpu(q)
    B, T, C = q.shape
    dkh = C // heads
    dctx = dctx.contiguous()
    dS = torch.empty((B, heads, T, T), device=q.device, dtype=torch.float32)
    dek = torch.empty((2 * window + 1, dkh), device=q.device, dtype=torch.float32)
    dev_ = torch.empty_like(dek)
    lengths = i32(lengths, q.device)
    check(_lib.load().ptpp_attention_win_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(emb_k), _ptr(emb_v), _ptr(probs), _ptr(dctx), _ptr(dS), _ptr(dq),
                                             _ptr(dk_), _ptr(dv), _ptr(dek), _ptr(dev_), _ptr(lengths), B, T, heads, dkh, _ld(q), C, _ld(dq),
                                             int(window), float(drop_p), int(drop_seed), dtype_code(q.dtype), _stream()),
          "ptpp_attention_win_bwd")
    return dek, dev_


# ----------------------------------------------------------------------------
# length regulator, positional encoding, DiffNet glue
# ----------------------------------------------------------------------------
def length_regulate_fwd(x, cum, Tf):
    _need_gpu(x)
    assert x.is_contiguous() and cum.dtype == torch.int32 and cum.is_contiguous()
    B, Tp, C = x.shape
    y = torch.empty((B, Tf, C), device=x.device, dtype=x.dtype)
    check(_lib.load().ptpp_length_regulate_fwd(_ptr(x), _ptr(cum), _ptr(y), B, Tp, Tf, C, dtype_code(x.dtype), _stream()),
          "ptpp_length_regulate_fwd")
    return y


def length_regulate_bwd(dy, cum, Tp):
    _need_gpu(dy)
    dy = dy.contiguous()
    B, Tf, C = dy.shape
    dx = torch.empty((B, Tp, C), device=dy.device, dtype=dy.dtype)
    check(_lib.load().ptpp_length_regulate_bwd(_ptr(dy), _ptr(cum), _ptr(dx), B, Tp, Tf, C, dtype_code(dy.dtype), _stream()),
          "ptpp_length_regulate_bwd")
    return dx


def posenc(x, pe, scale, drop_p=0.0, seed=0):
    _need_gpu(x)
    assert x.is_contiguous()
    B, T, C = x.shape
    y = torch.empty_like(x)
    check(_lib.load().ptpp_posenc_fwd(_ptr(x), _ptr(pe), _ptr(y), B, T, C, float(scale), float(drop_p), int(seed),
                                      dtype_code(x.dtype), _stream()), "ptpp_posenc_fwd")
    return y


def gate_fwd(a):
    _need_gpu(a)
    assert a.is_contiguous()
    B, T, C2 = a.shape
    g = torch.empty((B, T, C2 // 2), device=a.device, dtype=a.dtype)
    check(_lib.load().ptpp_gate_fwd(_ptr(a), _ptr(g), B * T, C2 // 2, dtype_code(a.dtype), _stream()), "ptpp_gate_fwd")
    return g


def gate_bwd(a, dg, da):
    """da: (B,T,2C) view with row stride >= 2C (written in place)."""
    B, T, C2 = a.shape
    check(_lib.load().ptpp_gate_bwd(_ptr(a), _ptr(dg.contiguous()), _ptr(da), B * T, C2 // 2, _ld(da),
                                    dtype_code(a.dtype), _stream()), "ptpp_gate_bwd")
    return da


def diffnet_post_fwd(o, x, skip, dnext, init, want_yin=True):
    """Returns (xn, yin)."""
    B, T, C = x.shape
    xn = torch.empty_like(x) if o is not None else x
    yin = torch.empty_like(x) if (want_yin and dnext is not None) else None
    check(_lib.load().ptpp_diffnet_post_fwd(_ptr(o), _ptr(x), _ptr(skip), _ptr(dnext), _ptr(xn) if o is not None else None,
                                            _ptr(yin), B, T, C, int(init), dtype_code(x.dtype), _stream()),
          "ptpp_diffnet_post_fwd")
    return xn, yin


_post_ok = {}


def conv1d_diffnet_post_supported(C, cin, dtype):
    key = (C, cin, dtype)
    ok = _post_ok.get(key)
    if ok is None:
        ok = _post_ok[key] = bool(_lib.load().ptpp_conv1d_diffnet_post_supported(int(C), int(cin), dtype_code(dtype)))
    return ok


def conv1d_diffnet_post(g, wp, bias, x, skip, dnext, init, lengths=None, out_mask=False, want_yin=True):
    """The DiffNet layer's 1 x 1 output projection of ``g`` with ``diffnet_post_fwd`` fused into its epilogue
    (ptpp_conv1d_diffnet_post): returns (xn, yin) and updates ``skip`` in place; the 2C-channel projection output is
    never stored.  Bit-identical to ``conv1d`` followed by ``diffnet_post_fwd``."""
    B, T, cin = g.shape
    C = x.shape[2]
    xn = torch.empty_like(x)
    yin = torch.empty_like(x) if (want_yin and dnext is not None) else None
    if lengths is not None:
        lengths = i32(lengths, g.device)
    assert x.is_contiguous() and skip.is_contiguous() and skip.dtype == torch.float32
    assert dnext is None or (dnext.is_contiguous() and dnext.dtype == torch.float32)
    _CONV_FMT.pack_into(_conv_buf, 0, g.data_ptr(), wp.data_ptr(), bias.data_ptr() if bias is not None else 0, 0, 0,
                        lengths.data_ptr() if lengths is not None else 0, B, T, cin, 2 * C, 1, 1, 0, _ld_fast(g), C, 0,
                        _ACT[None], 0, 1 if out_mask else 0, 1.0, BF16)
    check(_lib.load().ptpp_conv1d_diffnet_post(_conv_args_ref, x.data_ptr(), skip.data_ptr(),
                                               dnext.data_ptr() if dnext is not None else None, xn.data_ptr(),
                                               yin.data_ptr() if yin is not None else None, 1 if init else 0, _stream()),
          "ptpp_conv1d_diffnet_post")
    return xn, yin


def diffnet_layer_supported(C, dtype):
    return bool(_lib.load().ptpp_diffnet_layer_supported(int(C), dtype_code(dtype)))


def diffnet_pack_wstream(dil_wps, out_wps, C, cond_wps=None):
    """(L, bytes) uint8 operand stream of ``diffnet_layer_fwd`` from the layers' mode-2 dilated-conv and mode-0
    output-projection operands (ptpp_diffnet_pack_wstream); with ``cond_wps`` (mode-2 (2C, 1, 256) operands of the
    conditioner projections) the 80-stage form of the layer that projects the conditioner input itself."""
    lib = _lib.load()
    L = len(dil_wps)
    nbytes = lib.ptpp_diffnet_wstream_bytes_cond(int(C)) if cond_wps is not None else lib.ptpp_diffnet_wstream_bytes(int(C))
    ws = torch.empty((L, nbytes), device=dil_wps[0].device, dtype=torch.uint8)
    t1 = (ctypes.c_void_p * L)(*[t.data_ptr() for t in dil_wps])
    t2 = (ctypes.c_void_p * L)(*[t.data_ptr() for t in out_wps])
    if cond_wps is not None:
        t3 = (ctypes.c_void_p * L)(*[t.data_ptr() for t in cond_wps])
        check(lib.ptpp_diffnet_pack_wstream_cond(ctypes.cast(t1, ctypes.c_void_p), ctypes.cast(t3, ctypes.c_void_p),
                                                 ctypes.cast(t2, ctypes.c_void_p), ws.data_ptr(), L, int(C), _stream()),
              "ptpp_diffnet_pack_wstream_cond")
        return ws
    check(lib.ptpp_diffnet_pack_wstream(ctypes.cast(t1, ctypes.c_void_p), ctypes.cast(t2, ctypes.c_void_p), ws.data_ptr(), L, int(C),
                                        _stream()), "ptpp_diffnet_pack_wstream")
    return ws


def diffnet_layer_fwd(yin, x, cond, wstream, dil_b, out_b, dnext, skip, dil, init, lengths=None, save=False, want_yin=True,
                      skip_scaled=None, skip_scale=1.0, condx=None):
    """One DiffNet residual layer in ONE launch (ptpp_diffnet_layer_fwd, reference modules/denoiser.py:69-83): returns
    (xn, yin_next, a, g); ``skip`` (f32) is updated in place; ``cond``: this layer's (B, T, 2C) slice (a view with the row
    stride of the all-layer tensor), gate-interleaved like ``dil_b``.  ``save``: keep a (B,T,2C) and g (B,T,C) (training).
    ``skip_scaled``: optional (B,T,C) tensor of x's dtype that receives (skip * skip_scale) rounded once (last layer).
    ``condx``: the conditioner INPUT (B,T,256) instead of ``cond`` (None then): the layer projects it itself (``wstream`` in the
    80-stage form, ``dil_b`` = dilated-conv + conditioner biases)."""
    B, T, C = x.shape
    assert yin.is_contiguous() and x.is_contiguous() and skip.is_contiguous() and skip.dtype == torch.float32
    assert (cond is not None and cond.stride(2) == 1) or (condx is not None and condx.stride(2) == 1 and condx.shape[2] == 256)
    xn = torch.empty_like(x)
    yn = torch.empty_like(x) if (want_yin and dnext is not None) else None
    a = torch.empty((B, T, 2 * C), device=x.device, dtype=x.dtype) if save else None
    g = torch.empty_like(x) if save else None
    if lengths is not None:
        lengths = i32(lengths, x.device)
    args = _lib.DiffNetLayerArgs()
    args.yin, args.x, args.wstream = yin.data_ptr(), x.data_ptr(), wstream.data_ptr()
    args.cond = cond.data_ptr() if cond is not None else None
    if condx is not None:
        args.condx, args.ldcx = condx.data_ptr(), condx.stride(1)
    args.dil_b, args.out_b, args.skip, args.xn = dil_b.data_ptr(), out_b.data_ptr(), skip.data_ptr(), xn.data_ptr()
    args.dnext = dnext.data_ptr() if dnext is not None else None
    args.yin_next = yn.data_ptr() if yn is not None else None
    args.a_out = a.data_ptr() if save else None
    args.g_out = g.data_ptr() if save else None
    args.lengths = lengths.data_ptr() if lengths is not None else None
    if skip_scaled is not None:
        assert skip_scaled.is_contiguous() and skip_scaled.shape == x.shape and skip_scaled.dtype == x.dtype
        args.skip_scaled, args.skip_scale = skip_scaled.data_ptr(), float(skip_scale)
    args.B, args.T, args.C, args.dil, args.init, args.dtype = B, T, C, int(dil), 1 if init else 0, dtype_code(x.dtype)
    args.ldc = cond.stride(1) if cond is not None else 0
    check(_lib.load().ptpp_diffnet_layer_fwd(ctypes.byref(args), _stream()), "ptpp_diffnet_layer_fwd")
    return xn, yn, a, g


_gsave_ok = {}


def conv1d_gate_fwd_save_supported(C, cin, dtype):
    key = (C, cin, dtype)
    ok = _gsave_ok.get(key)
    if ok is None:
        ok = _gsave_ok[key] = bool(_lib.load().ptpp_conv1d_gate_fwd_save_supported(int(C), int(cin), dtype_code(dtype)))
    return ok


def conv1d_gate_fwd_save(x, wp, bias, C, ks, dil, pad, res, g, a, lengths=None):
    """DiffNet dilated conv (+ ``res`` = conditioner slice) with the gate in its epilogue and the pre-activation kept
    (ptpp_conv1d_gate_fwd_save): weights / bias / res in the gate-interleaved row order (pack mode 2); writes ``g`` (B,T,C)
    and ``a`` (B,T,2C, standard order).  Bit-identical to ``conv1d`` followed by ``gate_fwd``."""
    B, T, cin = x.shape
    assert g.is_contiguous() and a.is_contiguous() and a.shape == (B, T, 2 * C) and x.dtype == torch.bfloat16
    if lengths is not None:  # ragged batch: output rows past an utterance's end are zero (their row tiles skip the K loop)
        lengths = i32(lengths, x.device)
    _CONV_FMT.pack_into(_conv_buf, 0, x.data_ptr(), wp.data_ptr(), bias.data_ptr() if bias is not None else 0,
                        res.data_ptr() if res is not None else 0, g.data_ptr(), lengths.data_ptr() if lengths is not None else 0,
                        B, T, cin, 2 * C, ks, dil, pad, _ld_fast(x), C,
                        _ld_fast(res) if res is not None else 0, _ACT["gate"], 0, 1 if lengths is not None else 0, 1.0, BF16)
    check(_lib.load().ptpp_conv1d_gate_fwd_save(_conv_args_ref, a.data_ptr(), 2 * C, _stream()), "ptpp_conv1d_gate_fwd_save")


_gbwd_ok = {}


def conv1d_gate_bwd_supported(C, cin, dtype):
    key = (C, cin, dtype)
    ok = _gbwd_ok.get(key)
    if ok is None:
        ok = _gbwd_ok[key] = bool(_lib.load().ptpp_conv1d_gate_bwd_supported(int(C), int(cin), dtype_code(dtype)))
    return ok


def conv1d_gate_bwd(do, wpt, a, da, lengths=None):
    """dg = conv1x1(do, wpt) with ``gate_bwd(a, dg, da)`` fused into its epilogue (ptpp_conv1d_gate_bwd): ``da`` is a
    (B, T, 2C) view with row stride >= 2C written in place; dg is never stored.  Bit-identical to the two launches."""
    B, T, cin = do.shape
    C = a.shape[2] // 2
    assert a.is_contiguous() and a.dtype == do.dtype == torch.bfloat16
    if lengths is not None:  # ragged batch: do is zero past an utterance's end -- the (exact) input mask lets those row tiles skip
        lengths = i32(lengths, do.device)
    _CONV_FMT.pack_into(_conv_buf, 0, do.data_ptr(), wpt.data_ptr(), 0, 0, 0, lengths.data_ptr() if lengths is not None else 0,
                        B, T, cin, C, 1, 1, 0, _ld_fast(do), 0, 0, _ACT[None], 1 if lengths is not None else 0, 0, 1.0, BF16)
    check(_lib.load().ptpp_conv1d_gate_bwd(_conv_args_ref, a.data_ptr(), da.data_ptr(), _ld(da), _stream()),
          "ptpp_conv1d_gate_bwd")
    return da


_rt_gbwd_ok = {}


def conv1d_rt_gate_bwd_ok(do, C):
    """Whether the fused gate backward may run on the row-tile engine (ptpp_conv1d_rt_gate_bwd) for this launch: bf16, C = 256,
    Cin % 128 == 0, frame-level row counts (the threshold of ``conv1d_rt_ok``)."""
    if not (CONV_RT and do.is_cuda and do.dtype == torch.bfloat16 and do.stride(2) == 1):
        return False
    if do.shape[0] * do.shape[1] < conv_rt_min_rows():
        return False
    key = (C, do.shape[2])
    ok = _rt_gbwd_ok.get(key)
    if ok is None:
        ok = _rt_gbwd_ok[key] = bool(_lib.load().ptpp_conv1d_rt_gate_bwd_supported(int(C), int(do.shape[2]), BF16))
    return ok


def conv1d_rt_gate_bwd(do, wstream, a, da, lengths=None):
    """``conv1d_gate_bwd`` with the projection weight as the row-tile engine's operand stream (pack mode 4): same result bit
    for bit (ptpp_conv1d_rt_gate_bwd)."""
    B, T, cin = do.shape
    C = a.shape[2] // 2
    assert a.is_contiguous() and a.dtype == do.dtype == torch.bfloat16
    if lengths is not None:
        lengths = i32(lengths, do.device)
    _CONV_FMT.pack_into(_conv_buf, 0, do.data_ptr(), 0, 0, 0, 0, lengths.data_ptr() if lengths is not None else 0,
                        B, T, cin, C, 1, 1, 0, _ld_fast(do), 0, 0, _ACT[None], 1 if lengths is not None else 0, 0, 1.0, BF16)
    check(_lib.load().ptpp_conv1d_rt_gate_bwd(_conv_args_ref, wstream.data_ptr(), a.data_ptr(), da.data_ptr(), _ld(da), _stream()),
          "ptpp_conv1d_rt_gate_bwd")
    return da


_l1_scratch = {}


def _l1_scratch_of(device):
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream().value)
    w = _l1_scratch.get(key)
    if w is None:  # zero before the first call; every call leaves it zero (one per stream: calls on a stream are ordered)
        w = _l1_scratch[key] = torch.zeros(int(_lib.load().ptpp_l1_scratch_bytes()), device=device, dtype=torch.uint8)
    return w


def l1_masked_mean_fwd(pred, target, mask, denom, scale):
    """sum |pred - target| * mask[row] / denom / scale as a 0-dim f32 tensor (ptpp_l1_masked_mean_fwd): pred (..., cols) f32 or
    bf16 contiguous, target f32 of the same shape, mask (rows) f32 or None, denom a device scalar."""
    _need_gpu(pred)
    if not (pred.is_contiguous() and target.is_contiguous() and target.dtype == torch.float32 and target.shape == pred.shape):
        raise ValueError("l1_masked_mean: pred and an f32 target of the SAME shape, both contiguous (the kernel reads them element for element)")
    if not (denom.dtype == torch.float32 and denom.numel() == 1 and (mask is None or (mask.dtype == torch.float32 and mask.is_contiguous()))):
        raise ValueError("l1_masked_mean: denom must be an f32 device scalar, mask an f32 contiguous tensor")
    cols = pred.shape[-1] if mask is not None else 1
    rows = pred.numel() // cols
    if mask is not None and mask.numel() != rows:
        raise ValueError("l1_masked_mean: one mask entry per row expected")
    out = torch.empty((), device=pred.device, dtype=torch.float32)
    check(_lib.load().ptpp_l1_masked_mean_fwd(pred.data_ptr(), target.data_ptr(), mask.data_ptr() if mask is not None else None,
                                              denom.data_ptr(), float(scale), rows, cols, dtype_code(pred.dtype), out.data_ptr(),
                                              _l1_scratch_of(pred.device).data_ptr(), _stream()), "ptpp_l1_masked_mean_fwd")
    return out


def l1_masked_mean_bwd(pred, target, mask, denom, gout, scale):
    cols = pred.shape[-1] if mask is not None else 1
    rows = pred.numel() // cols
    dpred = torch.empty_like(pred)
    gout = gout.contiguous()
    check(_lib.load().ptpp_l1_masked_mean_bwd(pred.data_ptr(), target.data_ptr(), mask.data_ptr() if mask is not None else None,
                                              denom.data_ptr(), gout.data_ptr(), float(scale), rows, cols, dtype_code(pred.dtype),
                                              dpred.data_ptr(), _stream()), "ptpp_l1_masked_mean_bwd")
    return dpred


def mdn_nll_fwd(log_pi, log_sigma, mu, target, mask, lp_min, ls_min):
    """(rows.., G, D) f32 x3, target (rows.., D), mask (rows..) bool or None -> loss (rows.., D)."""
    G, D = mu.shape[-2], mu.shape[-1]
    loss = torch.empty(target.shape, device=mu.device, dtype=torch.float32)
    check(_lib.load().ptpp_mdn_nll_fwd(log_pi.data_ptr(), log_sigma.data_ptr(), mu.data_ptr(), target.data_ptr(),
                                       mask.data_ptr() if mask is not None else None, loss.data_ptr(), loss.numel() // D, G, D,
                                       float(lp_min), float(ls_min), _stream()), "ptpp_mdn_nll_fwd")
    return loss


def mdn_nll_bwd(log_pi, log_sigma, mu, target, mask, loss, gout, lp_min, ls_min):
    G, D = mu.shape[-2], mu.shape[-1]
    dlp, dls, dmu = torch.empty_like(log_pi), torch.empty_like(log_sigma), torch.empty_like(mu)
    check(_lib.load().ptpp_mdn_nll_bwd(log_pi.data_ptr(), log_sigma.data_ptr(), mu.data_ptr(), target.data_ptr(),
                                       mask.data_ptr() if mask is not None else None, loss.data_ptr(), gout.data_ptr(),
                                       dlp.data_ptr(), dls.data_ptr(), dmu.data_ptr(), loss.numel() // D, G, D, float(lp_min),
                                       float(ls_min), _stream()), "ptpp_mdn_nll_bwd")
    return dlp, dls, dmu


def ddpm_step(x, eps, noise, t, sra, srm1, c1, c2, logvar, want_lp=False):
    """x (B, ...) f32, eps same shape (f32 / bf16), noise f32 or None, t (B) int64 on the device -> x_{t-1} f32
    (ptpp_ddpm_step: the reverse-diffusion update as one pass).  ``want_lp``: also return x_{t-1} in eps's dtype (what the
    next step feeds the denoiser) from the same launch -> (out, out_lp)."""
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and eps.is_contiguous() and eps.shape == x.shape
    assert t.dtype == torch.int64 and t.is_contiguous() and t.numel() == x.shape[0]
    assert noise is None or (noise.dtype == torch.float32 and noise.is_contiguous() and noise.shape == x.shape)
    out = torch.empty_like(x)
    lp = torch.empty_like(eps) if want_lp else None
    B = x.shape[0]
    check(_lib.load().ptpp_ddpm_step_lp(x.data_ptr(), eps.data_ptr(), noise.data_ptr() if noise is not None else None,
                                        t.data_ptr(), sra.data_ptr(), srm1.data_ptr(), c1.data_ptr(), c2.data_ptr(),
                                        logvar.data_ptr(), out.data_ptr(), lp.data_ptr() if lp is not None else None, B,
                                        x.numel() // B, dtype_code(eps.dtype), _stream()),
          "ptpp_ddpm_step")
    return (out, lp) if want_lp else out


def sampler_head_supported(C, M, dtype):
    return bool(_lib.load().ptpp_sampler_head_supported(int(C), int(M), dtype_code(dtype)))


def sampler_head(s, ws_p, ws_b, wo_p, wo_b, x, noise, t, sra, srm1, c1, c2, logvar, win_p=None, win_b=None, ds0=None):
    """Everything between two DiffNet stacks of the reverse loop in one launch (ptpp_sampler_head; reference
    modules/denoiser.py:147-152 + modules/diffusion.py:283-302 + denoiser.py:131,76): s (B,T,C) bf16 = skip / sqrt(L) ->
    (x_{t-1} (B,T,M) f32, h0, yin0 (B,T,C) bf16 of the next step or None, None without ``win_p``)."""
    _need_gpu(s)
    B, T, C = s.shape
    M = x.shape[2]
    assert s.is_contiguous() and x.is_contiguous() and x.dtype == torch.float32 and x.shape[:2] == (B, T)
    assert noise is None or (noise.is_contiguous() and noise.dtype == torch.float32 and noise.shape == x.shape)
    assert t.dtype == torch.int64 and t.is_contiguous() and t.numel() == B
    x_out = torch.empty_like(x)
    a = _lib.SamplerHeadArgs()
    a.s, a.ws_p, a.ws_b, a.wo_p, a.wo_b = s.data_ptr(), ws_p.data_ptr(), ws_b.data_ptr(), wo_p.data_ptr(), wo_b.data_ptr()
    a.x, a.noise, a.t, a.x_out = x.data_ptr(), noise.data_ptr() if noise is not None else None, t.data_ptr(), x_out.data_ptr()
    a.sra, a.srm1, a.c1, a.c2, a.logvar = sra.data_ptr(), srm1.data_ptr(), c1.data_ptr(), c2.data_ptr(), logvar.data_ptr()
    h0 = yin0 = None
    if win_p is not None:
        assert ds0.is_contiguous() and ds0.dtype == torch.float32 and ds0.shape == (B, C)
        h0, yin0 = torch.empty_like(s), torch.empty_like(s)
        a.win_p, a.win_b, a.ds0, a.h0, a.yin0 = win_p.data_ptr(), win_b.data_ptr(), ds0.data_ptr(), h0.data_ptr(), yin0.data_ptr()
    a.B, a.T, a.C, a.M, a.dtype = B, T, C, M, dtype_code(s.dtype)
    check(_lib.load().ptpp_sampler_head(ctypes.byref(a), _stream()), "ptpp_sampler_head")
    return x_out, h0, yin0


def diffnet_post_bwd(gx, gskip, lengths):
    B, T, C = gx.shape
    dout = torch.empty((B, T, 2 * C), device=gx.device, dtype=gx.dtype)
    lengths = i32(lengths, gx.device)
    check(_lib.load().ptpp_diffnet_post_bwd(_ptr(gx), _ptr(gskip), _ptr(dout), _ptr(lengths), B, T, C,
                                            dtype_code(gx.dtype), _stream()), "ptpp_diffnet_post_bwd")
    return dout


def nsf_source_ok(f0, dim):
    return f0.is_cuda and f0.dtype == torch.float32 and bool(_lib.load().ptpp_nsf_source_supported(int(dim)))


def nsf_source(f0, rand_ini, noise, w, bias, sampling_rate, sine_amp, noise_std, voiced_threshold):
    """ptpp_nsf_source: f0 (B, L) f32, rand_ini (B, dim), noise (B, L, dim), w (dim) -> (B, L) f32 merged source."""
    B, L = f0.shape
    dim = rand_ini.shape[1]
    assert noise.shape == (B, L, dim) and w.numel() == dim
    f0, rand_ini, noise, w = f0.contiguous(), rand_ini.contiguous().float(), noise.contiguous(), w.detach().reshape(-1).float().contiguous()
    out = torch.empty((B, L), device=f0.device, dtype=torch.float32)
    lib = _lib.load()
    tab = torch.empty(int(lib.ptpp_nsf_source_scratch_bytes(B, L, dim)), device=f0.device, dtype=torch.uint8)
    check(lib.ptpp_nsf_source(_ptr(f0), _ptr(rand_ini), _ptr(noise), _ptr(w), float(bias), _ptr(out), B, L, dim,
                              float(sampling_rate), float(sine_amp), float(noise_std), float(voiced_threshold), _ptr(tab), tab.numel(),
                              _stream()), "ptpp_nsf_source")
    return out


def colsum_batch(x, out=None):
    B, T, C = x.shape
    if out is None:
        out = torch.empty((B, C), device=x.device, dtype=torch.float32)
    assert out.shape == (B, C) and out.dtype == torch.float32 and out.is_contiguous()
    check(_lib.load().ptpp_colsum_batch(_ptr(x), _ptr(out), B, T, C, dtype_code(x.dtype), _stream()), "ptpp_colsum_batch")
    return out


# ----------------------------------------------------------------------------
# anti-aliased snake, vocoder tail, layout bridges
# ----------------------------------------------------------------------------
def _taps(filt):
    arr = (ctypes.c_float * 12)()
    vals = filt.detach().reshape(-1).float().cpu().tolist() if isinstance(filt, torch.Tensor) else list(filt)
    assert len(vals) == 12, "anti-alias filters must have 12 taps"
    for i, v in enumerate(vals):
        arr[i] = v
    return arr


def aa_snake(x, log_alpha, taps_up, taps_down, out=None):
    """x: (B,T,C); log_alpha: (C) f32 device tensor; taps_*: ctypes float[12]."""
    _need_gpu(x)
    assert x.is_contiguous()
    B, T, C = x.shape
    y = out if out is not None else torch.empty_like(x)
    check(_lib.load().ptpp_aa_snake_fwd(_ptr(x), _ptr(y), _ptr(log_alpha), taps_up, taps_down, B, T, C,
                                        dtype_code(x.dtype), _stream()), "ptpp_aa_snake_fwd")
    return y


def amp_layer_supported(C, dtype):
    return bool(_lib.load().ptpp_amp_layer_supported(int(C), dtype_code(dtype)))


def amp_pack_wstream(wp, C, ks):
    """Packed (C, ks, C) conv operand -> the fragment stream the 16-bit fused AMP layer reads (ptpp_amp_pack_wstream)."""
    _need_gpu(wp)
    assert wp.is_contiguous() and wp.numel() == C * ks * C
    out = torch.empty_like(wp)
    check(_lib.load().ptpp_amp_pack_wstream(wp.data_ptr(), out.data_ptr(), int(C), int(ks), dtype_code(wp.dtype), _stream()),
          "ptpp_amp_pack_wstream")
    return out


_AMP_WSTREAMS = {}


def _amp_wstream(wp, C, ks):
    """Fragment stream of a packed operand, cached per (storage, version) -- callers that hold their own copy (BigVGAN._prepare)
    pass it as ``ws1`` / ``ws2`` instead."""
    key = (wp.data_ptr(), wp._version, int(C), int(ks), wp.dtype)
    hit = _AMP_WSTREAMS.get(key)
    if hit is None or hit[0]() is None:
        import weakref

        if len(_AMP_WSTREAMS) > 256:
            _AMP_WSTREAMS.clear()
        hit = (weakref.ref(wp), amp_pack_wstream(wp, C, ks))
        _AMP_WSTREAMS[key] = hit
    return hit[1]


def amp_layer(x, w1p, b1, w2p, b2, log_alpha1, log_alpha2, taps1, taps2, ks, dil, res2=None, out_scale=1.0,
              res_scale=1.0, out=None, ws1=None, ws2=None):
    """One fused AMP layer (ptpp_amp_layer_fwd): x (B,T,C) -> res_scale*x + out_scale*(conv2(act2(conv1(act1(x))))
    + b2) [+ res2].  taps1 / taps2: (up, down) ctypes float[12] pairs of act1 / act2 (``_taps``).  16-bit tensors read the
    weights as fragment streams (``ws1`` / ``ws2`` = amp_pack_wstream(w1p / w2p); built and cached here when not given)."""
    _need_gpu(x)
    assert x.is_contiguous() and x.dim() == 3
    B, T, C = x.shape
    if x.dtype != torch.float32:
        ws1 = ws1 if ws1 is not None else _amp_wstream(w1p, C, ks)
        ws2 = ws2 if ws2 is not None else _amp_wstream(w2p, C, ks)
    y = out if out is not None else torch.empty_like(x)
    a = _lib.AmpLayerArgs()
    a.x, a.y, a.res2 = x.data_ptr(), y.data_ptr(), _ptr(res2)
    a.w1p, a.w2p, a.b1, a.b2 = w1p.data_ptr(), w2p.data_ptr(), b1.data_ptr(), b2.data_ptr()
    a.w1s, a.w2s = _ptr(ws1), _ptr(ws2)
    a.log_alpha1, a.log_alpha2 = log_alpha1.data_ptr(), log_alpha2.data_ptr()
    a.up1, a.dn1 = taps1
    a.up2, a.dn2 = taps2
    a.B, a.T, a.C, a.ks, a.dil = B, T, C, int(ks), int(dil)
    a.out_scale, a.res_scale, a.dtype = float(out_scale), float(res_scale), dtype_code(x.dtype)
    if res2 is not None:
        assert res2.is_contiguous() and res2.shape == x.shape and res2.dtype == x.dtype
    check(_lib.load().ptpp_amp_layer_fwd(ctypes.byref(a), _stream()), "ptpp_amp_layer_fwd")
    return y


def snake_conv1d_supported(C, dtype):
    return bool(_lib.load().ptpp_snake_conv1d_supported(int(C), dtype_code(dtype)))


def snake_conv1d(x, ws, bias, log_alpha, taps, ks, dil, res=None, res2=None, out_scale=1.0, res_scale=1.0, out=None):
    """res_scale * res + out_scale * (conv(aa_snake(x)) + bias) [+ res2] in one launch (ptpp_snake_conv1d_fwd; wide BigVGAN
    stages).  ws: amp_pack_wstream of the packed (C, ks, C) weight; taps: the (up, down) ctypes float[12] pair (``_taps``)."""
    _need_gpu(x)
    assert x.is_contiguous() and x.dim() == 3
    B, T, C = x.shape
    y = out if out is not None else torch.empty_like(x)
    a = _lib.SnakeConvArgs()
    a.x, a.y, a.res, a.res2 = x.data_ptr(), y.data_ptr(), _ptr(res), _ptr(res2)
    a.ws, a.bias, a.log_alpha = ws.data_ptr(), bias.data_ptr(), log_alpha.data_ptr()
    a.up, a.dn = taps
    a.B, a.T, a.C, a.ks, a.dil = B, T, C, int(ks), int(dil)
    a.out_scale, a.res_scale, a.dtype = float(out_scale), float(res_scale), dtype_code(x.dtype)
    for t in (res, res2):
        assert t is None or (t.is_contiguous() and t.shape == x.shape and t.dtype == x.dtype)
    check(_lib.load().ptpp_snake_conv1d_fwd(ctypes.byref(a), _stream()), "ptpp_snake_conv1d_fwd")
    return y


def add3_scale(a, b, c, scale):
    _need_gpu(a)
    y = torch.empty_like(a)
    check(_lib.load().ptpp_add3_scale(_ptr(a), _ptr(b), _ptr(c), _ptr(y), float(scale), a.numel(), dtype_code(a.dtype),
                                      _stream()), "ptpp_add3_scale")
    return y


def conv_post_tanh(x, w, bias):
    """x: (B,T,C); w: (ks, C) f32; returns (B, T) f32."""
    _need_gpu(x)
    B, T, C = x.shape
    y = torch.empty((B, T), device=x.device, dtype=torch.float32)
    check(_lib.load().ptpp_conv_post_tanh(_ptr(x), _ptr(w), float(bias), _ptr(y), B, T, C, w.shape[0],
                                          dtype_code(x.dtype), _stream()), "ptpp_conv_post_tanh")
    return y


def snake_conv_post_supported(C, ks, dtype):
    return bool(_lib.load().ptpp_snake_conv_post_supported(int(C), int(ks), dtype_code(dtype)))


def snake_conv_post_tanh(x, log_alpha, taps, w, bias):
    """tanh(conv_post(aa_snake(x))) in one launch (ptpp_snake_conv_post_tanh): x (B,T,C) 16-bit; w (ks, C) f32 -> (B, T) f32."""
    _need_gpu(x)
    assert x.is_contiguous() and x.dim() == 3
    B, T, C = x.shape
    y = torch.empty((B, T), device=x.device, dtype=torch.float32)
    up, dn = taps
    check(_lib.load().ptpp_snake_conv_post_tanh(_ptr(x), _ptr(log_alpha), up, dn, _ptr(w), float(bias), _ptr(y), B, T, C,
                                                w.shape[0], dtype_code(x.dtype), _stream()), "ptpp_snake_conv_post_tanh")
    return y


def filtfilt(x, b, a, lengths=None):
    """Zero-phase IIR along the last axis of a contiguous f32 tensor (..., T) on the device (ptpp_filtfilt);
    b, a: coefficient sequences (host).  No host synchronisation."""
    _need_gpu(x)
    x = x.contiguous().float()
    T = x.shape[-1]
    rows = x.numel() // T
    y = torch.empty_like(x)
    tmp = torch.empty(rows * T, device=x.device, dtype=torch.float64)
    n = len(b) - 1
    assert len(a) == len(b)
    bb, aa = (ctypes.c_double * (n + 1))(*map(float, b)), (ctypes.c_double * (n + 1))(*map(float, a))
    check(_lib.load().ptpp_filtfilt(_ptr(x), _ptr(y), _ptr(tmp), _ptr(i32(lengths, x.device)) if lengths is not None else None,
                                    bb, aa, n, rows, T, T, _stream()), "ptpp_filtfilt")
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
    x = x.contiguous()
    B, T, C = x.shape
    y = torch.empty((B, C, T), device=x.device, dtype=torch.float32)
    check(_lib.load().ptpp_btc_to_bct(_ptr(x), _ptr(y), B, T, C, dtype_code(x.dtype), _stream()), "ptpp_btc_to_bct")
    return y


# ----------------------------------------------------------------------------
# Deferred finishing of parameter-gradient sums (include/ptpp.h "Deferred reduction")
# ----------------------------------------------------------------------------
_red_arena = {"t": None}


def red_defer_enable(device, mbytes=32):
    """Queue the finishing launches of the parameter-gradient column sums (LayerNorm dgamma / dbeta, attention position
    biases, scalar embeddings) until ``red_flush()``.  Only for callers whose destinations stay valid and unread until
    then (the trainer's flat gradient buffer: functional.enable_direct_grads)."""
    if _red_arena["t"] is None:
        t = torch.zeros(int(mbytes) << 20, device=device, dtype=torch.uint8)
        check(_lib.load().ptpp_red_defer(t.data_ptr(), t.numel()), "ptpp_red_defer")
        _red_arena["t"] = t


def red_defer_disable():
    if _red_arena["t"] is not None:
        red_flush()
        check(_lib.load().ptpp_red_defer(None, 0), "ptpp_red_defer")
        _red_arena["t"] = None


def red_deferred():
    return _red_arena["t"] is not None


def red_flush():
    """Finish every queued sum (one launch per producing stream) and order the current stream behind them."""
    if _red_arena["t"] is not None:
        check(_lib.load().ptpp_red_flush(_stream()), "ptpp_red_flush")


class red_immediate:
    """Calls inside finish their column sums at once (their destinations are read right away, e.g. tensors handed back to
    autograd)."""

    def __init__(self, on=True):
        self.on = on and _red_arena["t"] is not None

    def __enter__(self):
        if self.on:
            _lib.load().ptpp_red_defer_suspend(1)
        return self

    def __exit__(self, *exc):
        if self.on:
            _lib.load().ptpp_red_defer_suspend(-1)
        return False


# ----------------------------------------------------------------------------
# Training-step glue (csrc/glue.hip)
# ----------------------------------------------------------------------------
_loss_scratch = {}


def _loss_scratch_of(device):
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream().value)
    w = _loss_scratch.get(key)
    if w is None:
        w = _loss_scratch[key] = torch.zeros(int(_lib.load().ptpp_tts_losses_scratch_bytes()), device=device, dtype=torch.uint8)
    return w


def _tts_loss_args(pred, noise, flen, pv, cf0_t, vuv_t, y_dur, dur, plen, y_sty, sty_t, G_dur, G_sty, dec_scale, lp_min, ls_min,
                   total, comps, nll_dur, nll_sty):
    a = _lib.TtsLossArgs()
    B, Tf, M = pred.shape
    Tp = y_dur.shape[1]
    D = sty_t.shape[-1]
    for t in (pred, noise, pv, cf0_t, vuv_t, y_dur, dur, y_sty, sty_t):
        if not t.is_contiguous():
            raise ValueError("tts_losses: contiguous tensors expected")
    if not (noise.dtype == cf0_t.dtype == vuv_t.dtype == y_dur.dtype == dur.dtype == y_sty.dtype == sty_t.dtype == torch.float32):
        raise TypeError("tts_losses: float32 targets / head outputs expected")
    if pv.dtype != pred.dtype or pv.shape != (B, Tf, 2) or noise.shape != pred.shape or cf0_t.shape != (B, Tf) or vuv_t.shape != (B, Tf):
        raise ValueError("tts_losses: frame-level shapes do not match")
    if y_dur.shape != (B, Tp, 3 * G_dur) or dur.shape != (B, Tp) or y_sty.numel() != B * 3 * G_sty * D or sty_t.numel() != B * D:
        raise ValueError("tts_losses: phone-level / style shapes do not match")
    if flen.dtype != torch.int32 or plen.dtype != torch.int32 or flen.numel() != B or plen.numel() != B:
        raise ValueError("tts_losses: int32 lengths (B) expected")
    a.pred, a.noise, a.flen, a.pv = pred.data_ptr(), noise.data_ptr(), flen.data_ptr(), pv.data_ptr()
    a.cf0_tgt, a.vuv_tgt, a.y_dur, a.dur, a.plen = cf0_t.data_ptr(), vuv_t.data_ptr(), y_dur.data_ptr(), dur.data_ptr(), plen.data_ptr()
    a.y_sty, a.sty_tgt = y_sty.data_ptr(), sty_t.data_ptr()
    a.total, a.comps, a.nll_dur, a.nll_sty = total.data_ptr(), comps.data_ptr(), nll_dur.data_ptr(), nll_sty.data_ptr()
    a.scratch = _loss_scratch_of(pred.device).data_ptr()
    a.B, a.Tf, a.Tp, a.M, a.G_dur, a.G_sty, a.D_sty, a.dtype = B, Tf, Tp, M, G_dur, G_sty, D, dtype_code(pred.dtype)
    a.dec_scale, a.lp_min, a.ls_min = float(dec_scale), float(lp_min), float(ls_min)
    return a


def tts_losses_fwd(pred, noise, flen, pv, cf0_t, vuv_t, y_dur, dur, plen, y_sty, sty_t, G_dur, G_sty, dec_scale, lp_min=-7.0,
                   ls_min=-7.0):
    """All losses of the training step in one launch (ptpp_tts_losses_fwd).  Returns (total (), comps (7,), nll_dur, nll_sty)."""
    _need_gpu(pred)
    dev = pred.device
    total = torch.empty((), device=dev, dtype=torch.float32)
    comps = torch.empty((7,), device=dev, dtype=torch.float32)
    nll_dur = torch.empty(dur.shape, device=dev, dtype=torch.float32)
    nll_sty = torch.empty((sty_t.numel(),), device=dev, dtype=torch.float32)
    a = _tts_loss_args(pred, noise, flen, pv, cf0_t, vuv_t, y_dur, dur, plen, y_sty, sty_t, G_dur, G_sty, dec_scale, lp_min, ls_min,
                       total, comps, nll_dur, nll_sty)
    check(_lib.load().ptpp_tts_losses_fwd(ctypes.byref(a), _stream()), "ptpp_tts_losses_fwd")
    return total, comps, nll_dur, nll_sty


def tts_losses_bwd(saved, g_total, g_comps, G_dur, G_sty, dec_scale, lp_min=-7.0, ls_min=-7.0):
    (pred, noise, flen, pv, cf0_t, vuv_t, y_dur, dur, plen, y_sty, sty_t, comps, nll_dur, nll_sty) = saved
    dev = pred.device
    dpred, dpv = torch.empty_like(pred), torch.empty_like(pv)
    dy_dur, dy_sty = torch.empty_like(y_dur), torch.empty_like(y_sty)
    total = comps  # (unused by the backward)
    a = _tts_loss_args(pred, noise, flen, pv, cf0_t, vuv_t, y_dur, dur, plen, y_sty, sty_t, G_dur, G_sty, dec_scale, lp_min, ls_min,
                       total, comps, nll_dur, nll_sty)
    check(_lib.load().ptpp_tts_losses_bwd(ctypes.byref(a), _ptr(g_total), _ptr(g_comps), dpred.data_ptr(), dpv.data_ptr(),
                                          dy_dur.data_ptr(), dy_sty.data_ptr(), _stream()), "ptpp_tts_losses_bwd")
    return dpred, dpv, dy_dur, dy_sty


def q_sample_bct(mel, noise, step, sqrt_ac, sqrt_1mac, norm_scale, a_min, a_max, dtype):
    """mel (B, M, T) f32, noise (B, T, M) f32, step (B,) int64 -> noised normalised mel (B, T, M) in ``dtype``."""
    _need_gpu(mel)
    B, M, T = mel.shape
    assert mel.dtype == noise.dtype == torch.float32 and mel.is_contiguous() and noise.is_contiguous() and noise.shape == (B, T, M)
    assert step.dtype == torch.int64 and step.numel() == B and sqrt_ac.dtype == sqrt_1mac.dtype == torch.float32
    out = torch.empty((B, T, M), device=mel.device, dtype=dtype)
    use_scale = norm_scale is not None
    check(_lib.load().ptpp_q_sample_bct(mel.data_ptr(), noise.data_ptr(), step.data_ptr(), sqrt_ac.data_ptr(), sqrt_1mac.data_ptr(),
                                        sqrt_ac.numel(), float(norm_scale) if use_scale else 1.0, float(a_min), float(a_max),
                                        int(use_scale), out.data_ptr(), B, M, T, dtype_code(dtype), _stream()), "ptpp_q_sample_bct")
    return out


def step_sinusoid(step, dim, scale=1):
    """SinusoidalPosEmb: step (B,) int64 -> (B, dim) f32."""
    import math

    _need_gpu(step)
    assert step.dtype == torch.int64 and step.is_contiguous() and float(scale) == int(scale)
    half = dim // 2
    out = torch.empty((step.numel(), 2 * half), device=step.device, dtype=torch.float32)
    check(_lib.load().ptpp_step_sinusoid(step.data_ptr(), int(scale), -(math.log(10000) / (half - 1)), step.numel(), half,
                                         out.data_ptr(), _stream()), "ptpp_step_sinusoid")
    return out


def mish_fwd(x):
    assert x.dtype == torch.float32 and x.is_contiguous()
    _need_gpu(x)
    y = torch.empty_like(x)
    check(_lib.load().ptpp_mish_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "ptpp_mish_fwd")
    return y


def mish_bwd(x, gy):
    gy = gy.contiguous()
    gx = torch.empty_like(x)
    check(_lib.load().ptpp_mish_bwd(x.data_ptr(), gy.data_ptr(), gx.data_ptr(), x.numel(), _stream()), "ptpp_mish_bwd")
    return gx


def embed_cl_fwd(ids, table, lengths, scale, dtype):
    _need_gpu(ids)
    B, T = ids.shape
    V, C = table.shape
    assert ids.dtype == torch.int64 and ids.is_contiguous() and table.dtype == torch.float32 and table.is_contiguous()
    out = torch.empty((B, T, C), device=ids.device, dtype=dtype)
    check(_lib.load().ptpp_embed_cl_fwd(ids.data_ptr(), table.data_ptr(), _ptr(lengths), float(scale or 1.0), int(scale is not None),
                                        out.data_ptr(), B, T, C, V, dtype_code(dtype), _stream()), "ptpp_embed_cl_fwd")
    return out


def embed_cl_bwd(ids, dout, lengths, scale, dtable, padding_idx):
    B, T = ids.shape
    V, C = dtable.shape
    dout = dout.contiguous()
    assert dtable.dtype == torch.float32 and dtable.is_contiguous()
    check(_lib.load().ptpp_embed_cl_bwd(ids.data_ptr(), dout.data_ptr(), _ptr(lengths), float(scale or 1.0), int(scale is not None),
                                        dtable.data_ptr(), B, T, C, V, -1 if padding_idx is None else int(padding_idx),
                                        dtype_code(dout.dtype), _stream()), "ptpp_embed_cl_bwd")
    return dtable


def scalar_embed_add(x, track, w, bias, lengths):
    _need_gpu(x)
    B, T, C = x.shape
    assert x.is_contiguous() and track.is_contiguous() and track.dtype == torch.float32 and track.shape == (B, T)
    assert w.dtype == bias.dtype == torch.float32 and w.numel() == bias.numel() == C and w.is_contiguous() and bias.is_contiguous()
    out = torch.empty_like(x)
    check(_lib.load().ptpp_scalar_embed_add(x.data_ptr(), track.data_ptr(), w.data_ptr(), bias.data_ptr(), _ptr(lengths), out.data_ptr(),
                                            B, T, C, dtype_code(x.dtype), _stream()), "ptpp_scalar_embed_add")
    return out


def scalar_embed_bwd(dout, track, lengths, dw, db):
    """dw / db: f32 (C) buffers the sums are ADDED to."""
    B, T, C = dout.shape
    dout = dout.contiguous()
    assert dw.dtype == db.dtype == torch.float32 and dw.numel() == db.numel() == C and dw.is_contiguous() and db.is_contiguous()
    check(_lib.load().ptpp_scalar_embed_bwd(dout.data_ptr(), track.data_ptr(), _ptr(lengths), dw.data_ptr(), db.data_ptr(), B, T, C,
                                            dtype_code(dout.dtype), *reduction_scratch(dout.device), _stream()), "ptpp_scalar_embed_bwd")


def l2norm_fwd(x, eps=1e-12):
    """x (rows, C) f32 contiguous -> (x / max(||x||, eps), norms)."""
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
    y = torch.empty_like(x)
    n = torch.empty((x.shape[0],), device=x.device, dtype=torch.float32)
    check(_lib.load().ptpp_l2norm_fwd(x.data_ptr(), y.data_ptr(), n.data_ptr(), x.shape[0], x.shape[1], float(eps), _stream()),
          "ptpp_l2norm_fwd")
    return y, n


def l2norm_bwd(y, n, gy, eps=1e-12):
    gy = gy.contiguous()
    gx = torch.empty_like(y)
    check(_lib.load().ptpp_l2norm_bwd(y.data_ptr(), n.data_ptr(), gy.data_ptr(), gx.data_ptr(), y.shape[0], y.shape[1], float(eps),
                                      _stream()), "ptpp_l2norm_bwd")
    return gx


def durations_cumsum(dur):
    """(B, Tp) float32 (integer valued) or int64 durations -> int32 running sums."""
    _need_gpu(dur)
    if dur.dtype not in (torch.float32, torch.int64):
        dur = dur.to(torch.int64)
    dur = dur.contiguous()
    cum = torch.empty(dur.shape, device=dur.device, dtype=torch.int32)
    check(_lib.load().ptpp_durations_cumsum(dur.data_ptr(), int(dur.dtype == torch.float32), cum.data_ptr(), dur.shape[0], dur.shape[1],
                                            _stream()), "ptpp_durations_cumsum")
    return cum


def bcast_add_rows(x, e):
    """x (B, T, C) + e (B, C) f32 (rounded to x's dtype first) on every row."""
    _need_gpu(x)
    B, T, C = x.shape
    assert x.is_contiguous() and e.dtype == torch.float32 and e.is_contiguous() and e.shape == (B, C)
    y = torch.empty_like(x)
    check(_lib.load().ptpp_bcast_add_rows(x.data_ptr(), e.data_ptr(), y.data_ptr(), B, T, C, dtype_code(x.dtype), _stream()),
          "ptpp_bcast_add_rows")
    return y


def rows_sum(dy):
    """(B, T, C) -> (B, C) f32 sum over T."""
    dy = dy.contiguous()
    B, T, C = dy.shape
    de = torch.empty((B, C), device=dy.device, dtype=torch.float32)
    check(_lib.load().ptpp_rows_sum(dy.data_ptr(), de.data_ptr(), B, T, C, dtype_code(dy.dtype), _stream()), "ptpp_rows_sum")
    return de


def linear_small_fwd(x, w, bias, lengths):
    """(B, T, Cin) x (Cout <= 4, Cin) f32 weights -> (B, T, Cout) in x's dtype, rows past ``lengths`` zero (ptpp_linear_small_fwd)."""
    _need_gpu(x)
    B, T, Cin = x.shape
    Cout = w.shape[0]
    assert x.is_contiguous() and w.dtype == torch.float32 and w.is_contiguous() and w.numel() == Cout * Cin
    y = torch.empty((B, T, Cout), device=x.device, dtype=x.dtype)
    check(_lib.load().ptpp_linear_small_fwd(x.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(lengths), y.data_ptr(), B, T, Cin, Cout,
                                            dtype_code(x.dtype), _stream()), "ptpp_linear_small_fwd")
    return y


def linear_small_bwd(x, dy, w, lengths, dw, db, want_dx=True):
    """dx (or None); dw (Cout * Cin) / db (Cout): f32 buffers the sums are ADDED to."""
    B, T, Cin = x.shape
    Cout = dy.shape[-1]
    dy = dy.contiguous()
    assert dy.dtype == x.dtype and dw.dtype == db.dtype == torch.float32 and dw.is_contiguous() and db.is_contiguous()
    assert dw.numel() == Cout * Cin and db.numel() == Cout
    dx = torch.empty_like(x) if want_dx else None
    check(_lib.load().ptpp_linear_small_bwd(x.data_ptr(), dy.data_ptr(), w.data_ptr(), _ptr(lengths), _ptr(dx), dw.data_ptr(),
                                            db.data_ptr(), B, T, Cin, Cout, dtype_code(x.dtype), *reduction_scratch(x.device),
                                            _stream()), "ptpp_linear_small_bwd")
    return dx
