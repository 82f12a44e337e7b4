"""Differentiable ops of the hot path: ``torch.autograd.Function`` wrappers whose
forward AND backward are hand-written HIP kernels (promptttspp_amd/csrc) called
through the C ABI.  All activations are channels-last (B, T, C) in the compute
dtype; parameters stay f32 (master weights) and are packed/cast per version.
"""
import ctypes
import math
import os
import weakref
from types import SimpleNamespace

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib, ops

# ----------------------------------------------------------------------------
# dropout seeds: every dropout site of every forward draws a fresh counter value;
# the (p, seed) pair is all the backward needs to regenerate the mask.
# ----------------------------------------------------------------------------
_seed_state = {"base": None, "ctr": 0}


def manual_seed(seed):
    _seed_state["base"] = int(seed) & 0xFFFFFFFF
    _seed_state["ctr"] = 0


def next_seed():
    if _seed_state["base"] is None:
        _seed_state["base"] = torch.initial_seed() & 0xFFFFFFFF
    _seed_state["ctr"] += 1
    return ((_seed_state["base"] << 32) ^ (_seed_state["ctr"] * 0x9E3779B1)) & 0xFFFFFFFFFFFFFFFF


# ----------------------------------------------------------------------------
# packed-weight cache.  Every conv / linear weight is consumed as a packed K-contiguous operand in the
# compute dtype (mode 0: forward, mode 1: flipped / transposed for the data gradient); fused projections
# (q|k|v, the DiffNet conditioner stack) pack SEVERAL parameters into one operand.  An entry is valid
# for the same live Parameter objects at the same Tensor._version and storage.  After an optimiser step
# every trainable entry is stale: ``repack_all()`` refreshes all of them in ONE launch
# (ptpp_pack_conv_weights_batched, a device pointer table rebuilt only when the set of entries changes);
# anything still stale when it is next used is re-packed on its own.
# ----------------------------------------------------------------------------
class _PackEntry:
    __slots__ = ("refs", "vers", "wp", "mode", "dtype", "late", "__weakref__")

    def srcs(self):
        out = [r() for r in self.refs]
        return None if any(t is None for t in out) else out

    @staticmethod
    def stamp(ws):
        return tuple((w._version, w.data_ptr()) for w in ws)


_pack_cache = {}
_repack = {"key": None, "table": None, "map": None, "n": 0, "blocks": 0, "ents": [], "gen": None, "late": None}
_pack_gen = [0]  # bumped whenever an entry is added to / dropped from _pack_cache
# Round 6: operands the forward does not read for its first ~2 ms (variance adaptor, decoder: ``mark_late_pack``) are re-packed by a
# SECOND launch on the weight-gradient side stream -- idle during the forward -- beside the phone encoder instead of in front of it
# (the re-pack is 0.32 ms of serial HBM-bound work at the head of every step, ~40 % of it for these operands).  Whoever reads such
# an operand first joins that launch (``join_late_pack``: the current stream, the main stream and every registered gradient stream
# wait for its event).  PTPP_LATE_PACK=0: one launch on the caller's stream, as before.
_late = {"ids": set(), "ev": None, "pending": False, "on": os.environ.get("PTPP_LATE_PACK", "1") != "0"}


def mark_late_pack(params):
    """``params``: parameters whose packed operands are first read late in the forward (see ``_late`` above)."""
    n = len(_late["ids"])
    _late["ids"].update(id(p) for p in params)
    if len(_late["ids"]) != n:
        _repack["key"] = _repack["gen"] = None  # the launch tables are rebuilt with the new split


def join_late_pack():
    """Order the current stream (and the main / gradient streams) after the late re-pack launch, if one is in flight."""
    if not _late["pending"]:
        return
    _late["pending"] = False
    ev = _late["ev"]
    if torch.cuda.is_current_stream_capturing():  # (no cross-stream edge into a capture: the host waits instead)
        ev.synchronize()
        return
    cur = torch.cuda.current_stream()
    cur.wait_event(ev)
    main = _direct.get("main")
    for st in ([main] if main is not None else []) + _grad_streams:
        if st != cur:
            st.wait_event(ev)


def _pack_now(ws, dtype, mode):
    if mode == 2 and len(ws) > 1:  # the gate interleave is per source: pack each into its rows of the common operand
        w0 = ws[0].detach().reshape(ws[0].shape[0], ws[0].shape[1], -1)
        rows = [t.shape[0] for t in ws]
        wp = torch.empty((sum(rows), w0.shape[2], ops.cin_padded(w0.shape[1], dtype)), device=w0.device, dtype=dtype)
        off = 0
        for t, r in zip(ws, rows):
            ops.pack_conv_weight(t.detach().reshape(r, t.shape[1], -1), dtype, 2, out=wp[off:off + r])
            off += r
        return wp
    w = ws[0] if len(ws) == 1 else torch.cat([t.detach().reshape(t.shape[0], -1) for t in ws], dim=0)
    return ops.pack_conv_weight(w, dtype, mode)


def _packed_entry(ws, dtype, mode):
    if len(ws) == 1:  # (the common case, ~250 calls per training step: no generator / list objects on the hit path)
        w = ws[0]
        key = ((id(w),), mode, dtype)
        ent = _pack_cache.get(key)
        if ent is not None and ent.refs[0]() is w:
            v = ent.vers[0]
            if v[0] == w._version and v[1] == w.data_ptr():
                if ent.late and _late["pending"]:
                    join_late_pack()
                return ent
    else:
        key = (tuple(id(w) for w in ws), mode, dtype)
        ent = _pack_cache.get(key)
    if ent is not None:
        live = ent.srcs()
        if live is not None and all(a is b for a, b in zip(live, ws)):
            if ent.vers != _PackEntry.stamp(ws):
                ent.wp, ent.vers = _pack_now(ws, dtype, mode), _PackEntry.stamp(ws)
            elif ent.late and _late["pending"]:
                join_late_pack()
            return ent
    ent = _PackEntry()
    ent.late = False
    _pack_gen[0] += 1
    ent.refs = [weakref.ref(w, lambda _r, k=key, cache=_pack_cache, gen=_pack_gen: (cache.pop(k, None), gen.__setitem__(0, gen[0] + 1)))
                for w in ws]  # (tables bound as defaults: at interpreter shutdown the module globals go before the last parameters)
    ent.mode, ent.dtype = mode, dtype
    ent.wp, ent.vers = _pack_now(ws, dtype, mode), _PackEntry.stamp(ws)
    _pack_cache[key] = ent
    return ent


def packed(w, dtype, mode=0):
    """Packed K-contiguous operand of weight ``w`` (f32, (Cout,Cin[,ks]))."""
    if not isinstance(w, torch.nn.Parameter):
        return ops.pack_conv_weight(w, dtype, mode)
    return _packed_entry((w,), dtype, mode).wp


def rt_stream(w, x, cout, ks, dil, act, transposed=False, drop_p=0.0):
    """The weight as the operand stream of the row-tile conv kernel (pack mode 3, or 4 for the data gradient) when the launch
    ``ops.conv1d(x, ..)`` qualifies for it (ops.conv1d_rt_ok: bf16, 256 output channels, frame-level row count ...; or
    ops.conv1d_rt_ex_ok: the Conformer feed-forward convs, dropout allowed), else None."""
    if not isinstance(w, torch.nn.Parameter) or w.dim() != 3:
        return None
    if not ((drop_p == 0 and ops.conv1d_rt_ok(x, cout, ks, dil, act)) or ops.conv1d_rt_ex_ok(x, cout, ks, dil, act)):
        return None
    return packed(w, x.dtype, mode=4 if transposed else 3)


def packed_cat(ws, dtype, mode=0):
    """Packed operand of the row-wise concatenation of several (Cout_i, Cin[,1]) weights."""
    ws = tuple(ws)
    if not all(isinstance(w, torch.nn.Parameter) for w in ws):
        return _pack_now(ws, dtype, mode)
    return _packed_entry(ws, dtype, mode).wp


def _launch_repack():
    ops.pack_conv_weights_batched(_repack["table"], _repack["n"], _repack["map"], _repack["blocks"])
    late = _repack["late"]
    if late is None:
        return
    d = _direct
    if d["async"] and d["side"] is not None and not torch.cuda.is_current_stream_capturing():
        join_late_pack()  # (a previous late launch nobody read from: its event object is about to be re-recorded)
        side_h = d["side_h"]
        _lib.check(_lib.load().ptpp_stream_wait(side_h, ops._stream()), "ptpp_stream_wait")
        prev, ops._stream_override = ops._stream_override, side_h
        try:
            ops.pack_conv_weights_batched(*late)
        finally:
            ops._stream_override = prev
        if _late["ev"] is None:
            _late["ev"] = torch.cuda.Event()
        _late["ev"].record(d["side"])
        _late["pending"] = True
    else:
        ops.pack_conv_weights_batched(*late)


def repack_all(bumped=None):
    """Refresh every stale cached operand in one launch (call right after an in-place parameter update).
    ``bumped``: the caller (FusedAdamW) advanced ``Tensor._version`` of exactly these parameters by one since
    the previous call -- then, as long as the cache holds the same entries, the scan over ~450 entries
    (weakref + version + pointer reads, ~1 ms of host time per step) is skipped: the cached launch table is
    reused and the stamps of the entries it covers are advanced arithmetically.  An entry whose parameter
    changed in any other way simply fails its stamp check at the next use and is re-packed on its own."""
    refresh_bias_cats()
    if bumped is not None and _repack["key"] is not None and _repack["gen"] == (_pack_gen[0], id(bumped), len(bumped)):
        _launch_repack()
        for ent in _repack["ents"]:
            ent.vers = tuple((v + 1, p) for v, p in ent.vers)
        return
    todo = []
    for ent in list(_pack_cache.values()):
        ws = ent.srcs()
        if ws is None or not ws[0].is_cuda:
            continue
        st = _PackEntry.stamp(ws)
        if st != ent.vers:
            todo.append((ent, ws, st))
    if not todo:
        return
    ids = _late["ids"] if _late["on"] else ()
    for ent, ws, _ in todo:
        ent.late = bool(ids) and all(id(w) in ids for w in ws)
    todo.sort(key=lambda t: t[0].late)  # (stable: early operands first, then the late ones)
    key = tuple((id(e), e.wp.data_ptr(), st) for e, _, st in todo)
    key = tuple((k[0], k[1], tuple(p for _, p in k[2])) for k in key)  # entry, dst, source pointers
    if key != _repack["key"]:
        import numpy as np

        def table(part):
            rows, blk, owners = [], 0, []
            for ent, ws, _ in part:
                w0 = ws[0]
                cin = w0.shape[1]
                ks = w0.shape[2] if w0.dim() == 3 else 1
                dcode = ops.dtype_code(ent.dtype)
                total_cout = sum(w.shape[0] for w in ws)
                innerp = ops.cin_padded(cin if ent.mode not in (1, 4) else total_cout, ent.dtype)
                off = 0
                for w in ws:
                    assert w.dtype == torch.float32 and w.is_contiguous()
                    rows.append([w.data_ptr(), ent.wp.data_ptr(), w.shape[0], cin, ks, ent.mode, dcode, innerp, off, blk])
                    nb = ((w.shape[0] + 31) // 32) * ((cin + 31) // 32)
                    owners.append(np.full(nb, len(rows) - 1, dtype=np.int32))
                    blk += nb
                    off += w.shape[0]
            if not rows:
                return None
            dev = part[0][1][0].device
            return (torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev), len(rows),
                    torch.from_numpy(np.concatenate(owners)).to(dev), blk)  # table, rows, block -> table row, blocks

        early = table([t for t in todo if not t[0].late])
        _repack["late"] = table([t for t in todo if t[0].late])
        _repack["key"] = key
        if early is None:  # (everything is late: nothing to hide it behind)
            early, _repack["late"] = _repack["late"], None
        _repack["table"], _repack["n"], _repack["map"], _repack["blocks"] = early
    _launch_repack()
    for ent, _, st in todo:
        ent.vers = st
    # the fast path is valid while the cache holds exactly these entries and the caller bumps the same list
    bumped_ids = None if bumped is None else {id(p) for p in bumped}
    if bumped is not None and all(id(w) in bumped_ids for _, ws, _ in todo for w in ws):
        _repack["ents"] = [ent for ent, _, _ in todo]
        _repack["gen"] = (_pack_gen[0], id(bumped), len(bumped))
    else:
        _repack["gen"] = None


_cat_cache = {}


def _cat_cached(ts, kind, make):
    """Cache of a tensor derived from a LIST of live parameters (fused biases, permuted operands)."""
    key = (tuple(id(t) for t in ts), kind)
    vers = tuple((t._version, t.data_ptr()) for t in ts)
    ent = _cat_cache.get(key)
    if ent is not None and ent[0] == vers and all(r() is t for r, t in zip(ent[2], ts)):
        return ent[1]
    val = make()
    _cat_cache[key] = (vers, val, [weakref.ref(t, lambda _r, k=key, cache=_cat_cache: cache.pop(k, None)) for t in ts])
    return val


# Concatenated biases of layers that share a GEMM (q | k | v, the MDN heads, the DiffNet projections): a PERSISTENT f32 buffer per
# parameter list.  After an optimiser step ``repack_all`` refreshes every stale buffer with ONE multi-tensor copy
# (``refresh_bias_cats``) instead of one torch.cat per list at its next use (~10 launches per training step); a buffer that is
# still stale when it is used is rebuilt on its own.
_bias_cats = {}
_bias_gen = [0]
_bias_plan = {"gen": -1, "dst": [], "src": [], "ents": []}


class _BiasCat:
    __slots__ = ("refs", "vers", "buf", "views")


def bias_cat(bs):
    bs = tuple(bs)
    if not all(isinstance(b, torch.nn.Parameter) for b in bs):
        return torch.cat([b.detach().float() for b in bs], dim=0)
    key = tuple(id(b) for b in bs)
    ent = _bias_cats.get(key)
    stamp = tuple((b._version, b.data_ptr()) for b in bs)
    if ent is not None and all(r() is b for r, b in zip(ent.refs, bs)):
        if ent.vers != stamp:
            torch.cat([b.detach().float() for b in bs], dim=0, out=ent.buf)
            ent.vers = stamp
        return ent.buf
    ent = _BiasCat()
    ent.refs = [weakref.ref(b, lambda _r, k=key, cats=_bias_cats, gen=_bias_gen: (cats.pop(k, None), gen.__setitem__(0, gen[0] + 1))) for b in bs]
    # (the tables are bound as defaults: at interpreter shutdown the module globals are gone before the last parameters die)
    ent.buf = torch.cat([b.detach().float() for b in bs], dim=0)
    ent.vers = stamp
    off, ent.views = 0, []
    for b in bs:
        ent.views.append(ent.buf[off:off + b.numel()].view(b.shape))
        off += b.numel()
    _bias_cats[key] = ent
    _bias_gen[0] += 1
    return ent.buf


def refresh_bias_cats():
    """One multi-tensor copy of every f32 device source into its slice of its concatenated buffer (after a parameter update)."""
    pl = _bias_plan
    if pl["gen"] != _bias_gen[0]:
        pl["dst"], pl["src"], pl["ents"] = [], [], []
        for ent in _bias_cats.values():
            srcs = [r() for r in ent.refs]
            if any(b is None or not b.is_cuda or b.dtype != torch.float32 for b in srcs):
                continue
            pl["ents"].append((ent, srcs))
            pl["dst"].extend(ent.views)
            pl["src"].extend(b.detach() for b in srcs)
        pl["gen"] = _bias_gen[0]
    if pl["dst"]:
        torch._foreach_copy_(pl["dst"], pl["src"])
        for ent, srcs in pl["ents"]:
            ent.vers = tuple((b._version, b.data_ptr()) for b in srcs)


def clear_caches():
    _pack_cache.clear()
    _cat_cache.clear()
    _bias_cats.clear()
    _bias_gen[0] += 1
    _pack_gen[0] += 1
    _repack["key"] = None
    _repack["gen"] = None
    _repack["late"] = None
    if _late["pending"]:
        _late["ev"].synchronize()
        _late["pending"] = False
    _late["ids"].clear()


def _f32c(t):
    return None if t is None else t.detach().float().contiguous()


# ----------------------------------------------------------------------------
# direct gradient accumulation: the weight-gradient kernels ADD (atomics) into whatever
# f32 buffer they are given.  When the trainer has laid all ``p.grad`` out as views of one
# pre-zeroed flat buffer (parallel.FlatGradReducer), the backward functions hand the kernels
# ``p.grad`` itself and return None to autograd: no zero-fill, no ``grad +=`` launch, no
# temporary per parameter.  Autograd's post-accumulate hooks do not fire for such
# parameters, so uses are counted in forward and the reducer is notified when the last
# pending use of a parameter has been accumulated.
# ----------------------------------------------------------------------------
_direct = {"on": False, "notify": None, "uses": {}, "async": False, "side": None, "side_h": None, "keep": []}


def enable_direct_grads(on=True, notify=None, async_wgrad=True):
    """``async_wgrad``: in direct mode the weight-gradient kernels (which nothing downstream in backward
    consumes) run on a side stream, concurrently with the data-gradient chain on the main stream -- each
    kernel alone leaves the machine under-used (latency / write-phase bound), together they overlap.
    ``sync_wgrad_stream()`` joins the streams (FlatGradReducer.finish / before a bucket all-reduce)."""
    _direct["on"], _direct["notify"] = bool(on), notify
    _direct["async"] = bool(on) and bool(async_wgrad) and not os.environ.get("PTPP_NO_ASYNC_WGRAD")
    _direct["uses"].clear()
    sync_wgrad_stream()  # also releases the tensors held for the side stream
    # the finishing launches of the parameter-gradient column sums wait for ops.red_flush() (FlatGradReducer.finish, before a
    # bucket's collective, FusedAdamW.step): include/ptpp.h "Deferred reduction"; PTPP_NO_DEFERRED_SUMS=1 finishes each at once
    if torch.cuda.is_available():
        if on and not os.environ.get("PTPP_NO_DEFERRED_SUMS"):
            ops.red_defer_enable(torch.device("cuda", torch.cuda.current_device()))
        else:
            ops.red_defer_disable()
    if _direct["async"]:
        create_side_stream()


class wgrad_stream:
    """``with wgrad_stream(t1, t2, ...):`` -- run the enclosed launches on the weight-gradient side stream
    (after everything already enqueued on the current stream); the tensors are kept alive for it.
    Default: only this package's own launches move (``ops._stream_override`` + one ``ptpp_stream_wait``
    call: ~3 us of host time); ``torch_ops=True`` also switches torch's current stream, for blocks that
    allocate or run torch ops (~25 us).  The tensors the side stream reads are held in a list until the
    streams are joined again (``sync_wgrad_stream``) instead of ``Tensor.record_stream`` (cheaper, and the
    allocator cannot hand their memory to the main stream before the join)."""

    __slots__ = ("tensors", "torch_ops", "ctx", "on")

    def __init__(self, *tensors, torch_ops=False):
        self.tensors = tensors
        self.torch_ops = torch_ops
        self.ctx = None
        self.on = False

    def __enter__(self):
        d = _direct
        if not d["async"] or not self.tensors or self.tensors[0] is None or not self.tensors[0].is_cuda:
            return self
        side_h = d["side_h"]
        if side_h is None:
            if torch.cuda.is_current_stream_capturing():
                return self
            d["side"] = torch.cuda.Stream(device=self.tensors[0].device)
            side_h = d["side_h"] = ctypes.c_void_p(d["side"].cuda_stream)
        _lib.check(_lib.load().ptpp_stream_wait(side_h, ops._stream()), "ptpp_stream_wait")
        self.on = True
        if self.torch_ops:
            self.ctx = torch.cuda.stream(d["side"])
            self.ctx.__enter__()
        # this package's own launches follow the override in both modes (under ops.pinned_stream the handle of the
        # main stream is cached, so torch's stream context alone would not move them)
        ops._stream_override = side_h
        return self

    def __exit__(self, *exc):
        if self.on:
            ops._stream_override = None
            if self.ctx is not None:
                self.ctx.__exit__(*exc)
            keep = _direct["keep"]
            keep.extend(self.tensors)
            if len(keep) > 4096:  # nobody joined the streams for a long time: join here
                sync_wgrad_stream()
        return False


def create_side_stream(device=None):
    """Create the weight-gradient side stream now (idempotent).  Call it BEFORE the process group is set up:
    HIP multiplexes streams onto a few hardware queues in creation order, and a side stream created after
    RCCL's own streams (7th stream of the process in the one-rank RCCL experiment) shared the main stream's
    queue -- its kernels then ran strictly after the main stream's, never beside them
    (tools/prof_overlap.py: 0.0 of 64 ms overlapped, against 61 of 82 ms without RCCL)."""
    if _direct["side"] is None and torch.cuda.is_available():
        _direct["side"] = torch.cuda.Stream(device=device)
        _direct["side_h"] = ctypes.c_void_p(_direct["side"].cuda_stream)
    return _direct["side"]


_grad_streams = []  # streams besides the caller's that run backward nodes (the model's branch streams)


def register_gradient_stream(stream):
    """A stream on which autograd will run part of the backward (a branch of the forward issued on it): a bucket
    collective must be ordered after the gradient kernels of ALL such streams, not only of the one whose node reported
    the bucket's last parameter."""
    if all(s is not stream for s in _grad_streams):
        _grad_streams.append(stream)


SECOND_WGRAD_STREAM = __import__("os").environ.get("PTPP_WGRAD_STREAM2", "1") != "0"


def second_wgrad_stream(device):
    """Raw handle of a second stream for weight-gradient launches that are independent of the ones on the side stream (the
    Conformer block's depthwise / 1 x 1 group beside its k = 9 group, ptpp_conformer_block_bwd), or None.  It is the model's
    prompt-branch stream when that one is registered as a gradient stream: its own backward is long over when the phone encoder's
    runs (profiles/r06_streams.txt), ``sync_wgrad_stream`` and ``side_stream_for_collective`` already join it, and no new stream
    means no new hardware-queue assignment.  PTPP_WGRAD_STREAM2=0 turns it off."""
    s = _direct.get("wgrad2")
    if not SECOND_WGRAD_STREAM or s is None or _direct["side"] is None or s.device != device:
        return None
    if s == torch.cuda.current_stream() or all(g is not s for g in _grad_streams):
        return None
    return ctypes.c_void_p(s.cuda_stream)


def offer_second_wgrad_stream(stream):
    """The model names the registered gradient stream that is idle during the phone encoder's backward."""
    _direct["wgrad2"] = stream


def side_stream_for_collective():
    """The weight-gradient side stream, made to wait for everything enqueued so far on the current stream and on every
    registered gradient stream -- or None when it is not in use.  A collective issued under ``torch.cuda.stream(<it>)`` is
    then ordered after every gradient kernel of all streams without stalling the current one."""
    d = _direct
    if not d["async"] or d["side"] is None:
        if _grad_streams:  # no side stream: the collective follows the current stream, which then waits for the others
            cur = torch.cuda.current_stream()
            for s in _grad_streams + ([d["main"]] if d.get("main") is not None else []):
                if s != cur:
                    cur.wait_stream(s)
        return None
    lib = _lib.load()
    _lib.check(lib.ptpp_stream_wait(d["side_h"], ops._stream()), "ptpp_stream_wait")
    for s in _grad_streams:
        _lib.check(lib.ptpp_stream_wait(d["side_h"], ctypes.c_void_p(s.cuda_stream)), "ptpp_stream_wait")
    main = d.get("main_h")
    if main is not None:
        _lib.check(lib.ptpp_stream_wait(d["side_h"], main), "ptpp_stream_wait")
    return d["side"]


def sync_wgrad_stream():
    """The current stream waits for the weight-gradient kernels enqueued so far; the tensors held for the
    side stream are released (whatever reuses their memory is enqueued after this wait)."""
    cur = torch.cuda.current_stream()
    main = _direct.get("main")
    join_late_pack()  # (a late re-pack launch still in flight on the side stream: every stream that may read its operands waits)
    if _direct["side"] is not None:
        cur.wait_stream(_direct["side"])
        if main is not None and main != cur:
            # called from a backward node that runs on a branch stream (the keep-flush of wgrad_stream): the held tensors
            # were allocated on the main stream, whose pool gets them back -- it has to wait for the side stream too
            main.wait_stream(_direct["side"])
            for s in _grad_streams:  # (the second weight-gradient stream reads held tensors too)
                if s != main:
                    main.wait_stream(s)
        _direct["keep"].clear()
    # explicit join of the branch streams: their backward kernels write p.grad in place (LayerNorm / BatchNorm / GRU
    # parameters, the owner-block weight gradients); whatever the caller enqueues next (the optimizer, a collective) must
    # not rely on autograd's implicit leaf-stream synchronisation for that
    for s in _grad_streams:
        if s != cur:
            cur.wait_stream(s)


def reset_direct_uses():
    _direct["uses"].clear()


def direct_grads_enabled():
    return _direct["on"]


def _sink(p):
    """``p.grad`` if gradients of parameter ``p`` can be accumulated in place, else None."""
    if not _direct["on"] or not isinstance(p, torch.nn.Parameter) or not p.requires_grad:
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous():
        return None
    return g


def _use(p):
    """forward side: one more pending in-place accumulation into p.grad"""
    if _direct["notify"] is not None:
        u = _direct["uses"]
        u[id(p)] = u.get(id(p), 0) + 1


def _done(p):
    """backward side: an in-place accumulation into p.grad has been enqueued"""
    if _direct["notify"] is not None:
        u = _direct["uses"]
        n = u.get(id(p), 1) - 1
        if n <= 0:
            u.pop(id(p), None)
            _direct["notify"](p)
        else:
            u[id(p)] = n


def _kc(dtype):
    """elements per 16-byte MFMA operand chunk"""
    return 4 if dtype == torch.float32 else 8


def conv_cfg(ks=1, dil=1, pad=0, act=None, lengths=None, in_mask=False, out_mask=False, out_scale=1.0, drop_p=0.0):
    assert act in (None, "none", "relu"), "training-capable conv epilogues: none | relu"
    return SimpleNamespace(ks=ks, dil=dil, pad=pad, act=act, lengths=lengths, in_mask=in_mask, out_mask=out_mask,
                           out_scale=out_scale, drop_p=drop_p)


class Conv1dFn(Function):
    """y = res + out_scale * drop(mask(act(conv(x_masked) + b)))  (channels-last)."""

    @staticmethod
    def forward(ctx, x, w, b, res, cfg):
        w3 = w if w.dim() == 3 else w.unsqueeze(-1)
        cout, cin, ks = w3.shape
        assert ks == cfg.ks and cin == x.shape[-1]
        seed = next_seed() if cfg.drop_p > 0 else 0
        kc = _kc(x.dtype)
        xk, wk = x, w
        if cin % kc:  # tiny channel counts: zero-pad K to the 16-byte operand granule
            xk = torch.nn.functional.pad(x, (0, kc - cin % kc))
            wk = torch.nn.functional.pad(w3.detach(), (0, 0, 0, kc - cin % kc))
        ws = rt_stream(w, x, cout, ks, cfg.dil, cfg.act, drop_p=cfg.drop_p) if cin % kc == 0 else None
        y = ops.conv1d(xk, packed(wk, x.dtype) if ws is None else None, _f32c(b), cout, ks=ks, dil=cfg.dil, pad=cfg.pad, act=cfg.act,
                       lengths=cfg.lengths, in_mask=cfg.in_mask, out_mask=cfg.out_mask, res=res,
                       out_scale=cfg.out_scale, drop_p=cfg.drop_p, drop_seed=seed, wstream=ws)
        ctx.cfg, ctx.seed, ctx.has_res, ctx.has_b = cfg, seed, res is not None, b is not None
        # in-place gradient targets (None -> autograd accumulates the returned tensors)
        ctx.direct = None
        if any(ctx.needs_input_grad) and cin % kc == 0 and cout % kc == 0 and _sink(w) is not None and (b is None or _sink(b) is not None):
            ctx.direct = (w, b)
            _use(w)
            if b is not None:
                _use(b)
        ctx.save_for_backward(x, w, y if cfg.act == "relu" else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        cfg = ctx.cfg
        w3 = w if w.dim() == 3 else w.unsqueeze(-1)
        cout, cin, ks = w3.shape
        relu = cfg.act == "relu"
        dy = dy.contiguous()
        if cout % 4 and (relu or cfg.out_mask or cfg.out_scale != 1.0):
            # 1-3 output channels (pitch/V-UV head): a (B,T,2) tensor, not worth a vector kernel
            assert cfg.drop_p == 0
            dz = dy * cfg.out_scale
            if cfg.out_mask:
                t = torch.arange(dy.shape[1], device=dy.device)
                dz = dz * (t[None, :] < cfg.lengths[:, None]).unsqueeze(-1).to(dz.dtype)
            if relu:
                dz = dz * (y > 0).to(dz.dtype)
        elif relu or cfg.out_mask or cfg.out_scale != 1.0 or cfg.drop_p > 0:
            dz = ops.epilogue_bwd(dy, y, cfg.lengths, cfg.out_scale, relu, cfg.out_mask, cfg.drop_p, ctx.seed)
        else:
            dz = dy
        dx = dw = db = None
        kc = _kc(dz.dtype)
        wk, coutp = w, cout
        if cout % kc:  # e.g. the 2-channel pitch head: pad the gradient rows to the operand granule
            coutp = cout + kc - cout % kc
            dz = torch.nn.functional.pad(dz, (0, coutp - cout))
            wk = torch.nn.functional.pad(w3.detach(), (0, 0, 0, 0, 0, coutp - cout))
        if ctx.needs_input_grad[0]:
            ws = rt_stream(w, dz, cin, ks, cfg.dil, None, transposed=True) if coutp == cout else None
            dx = ops.conv1d(dz, packed(wk, dz.dtype, mode=1) if ws is None else None, None, cin, ks=ks, dil=cfg.dil,
                            pad=(ks - 1) * cfg.dil - cfg.pad, lengths=cfg.lengths, out_mask=cfg.in_mask, wstream=ws)
        if ctx.direct is not None:
            pw, pb = ctx.direct
            with wgrad_stream(x, dz):
                ops.conv1d_wgrad(x, dz, cin, cout, ks, cfg.dil, cfg.pad, cfg.lengths, cfg.in_mask, pb is not None,
                                 dw_out=pw.grad, db_out=pb.grad if pb is not None else None)
            _done(pw)
            if pb is not None:
                _done(pb)
        elif ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2]):
            xk = x if cin % 4 == 0 else torch.nn.functional.pad(x, (0, 4 - cin % 4))
            dw, db = ops.conv1d_wgrad(xk, dz, cin, coutp, ks, cfg.dil, cfg.pad, cfg.lengths, cfg.in_mask, ctx.has_b)
            dw = dw[:cout].reshape(w.shape)
            db = db[:cout] if db is not None else None
        return dx, dw, db, (dy if ctx.has_res else None), None


def conv1d(x, w, b=None, res=None, **kw):
    return Conv1dFn.apply(x, w, b, res, conv_cfg(**kw))


class FusedLinearFn(Function):
    """y = x @ cat(w_0..w_n)^T + cat(b_0..b_n): several nn.Linear layers that read the same
    input (q|k|v) as ONE GEMM, each layer keeping its own parameters and gradients."""

    @staticmethod
    def forward(ctx, x, n, *params):
        ws, bs = params[:n], params[n:]
        ctx.n, ctx.couts = n, [w.shape[0] for w in ws]
        y = ops.conv1d(x, packed_cat(ws, x.dtype), bias_cat(bs), sum(ctx.couts))
        ctx.direct = any(ctx.needs_input_grad) and all(_sink(p) is not None for p in params)
        if ctx.direct:
            for p in params:
                _use(p)
        ctx.params = params  # long-lived leaves: kept by reference (their .grad is the target)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        params, n = ctx.params, ctx.n
        ws, bs = params[:n], params[n:]
        dy = dy.contiguous()
        cin = x.shape[-1]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv1d(dy, packed_cat(ws, dy.dtype, mode=1), None, cin)
        gw, gb, c0 = [None] * n, [None] * n, 0
        if ctx.direct and n > 4 and x.is_cuda:
            # many layers over few rows (the DiffNet's 20 step projections on B rows): ONE weight-gradient launch with the
            # concatenated output channels and one multi-tensor add of each layer's row block into its own gradient -- and ONE
            # fork of the side stream: 20 forks were 20 event records on the main stream with no kernel between them, 117 us
            # of bubbles on the step's critical path (profiles/r06_experiments.md)
            with wgrad_stream(x, dy, torch_ops=True):
                dwc, dbc = ops.conv1d_wgrad(x, dy, cin, sum(ctx.couts), 1, 1, 0)
                dwl, dbl = dwc.view(sum(ctx.couts), -1).split(ctx.couts), dbc.split(ctx.couts)
                torch._foreach_add_([w.grad.view(co, -1) for w, co in zip(ws, ctx.couts)] + [b.grad for b in bs], list(dwl) + list(dbl))
            for p in params:
                _done(p)
            return (dx, None, *gw, *gb)
        with wgrad_stream(*((x, dy) if ctx.direct else ())):  # (one fork for all layers)
            for i, (w, b, co) in enumerate(zip(ws, bs, ctx.couts)):
                dw, db = ops.conv1d_wgrad(x, dy[:, :, c0 : c0 + co], cin, co, 1, 1, 0,
                                          dw_out=w.grad if ctx.direct else None, db_out=b.grad if ctx.direct else None)
                c0 += co
                if not ctx.direct:
                    gw[i], gb[i] = dw.view_as(w), db
        if ctx.direct:
            for p in params:
                _done(p)
        return (dx, None, *gw, *gb)


def linear_fused(x, layers):
    """layers: nn.Linear modules sharing the input x (B, T, Cin) -> (B, T, sum Cout_i)."""
    ws = [l.weight for l in layers]
    bs = [l.bias for l in layers]
    return FusedLinearFn.apply(x, len(ws), *ws, *bs)


def linear(x, w, b=None, **kw):
    """nn.Linear on the last dim of a (B, T, Cin) or (N, Cin) tensor."""
    if x.dim() == 2:
        return Conv1dFn.apply(x.unsqueeze(0), w, b, None, conv_cfg(**kw)).squeeze(0)
    return Conv1dFn.apply(x, w, b, None, conv_cfg(**kw))


# ----------------------------------------------------------------------------
class LayerNormFn(Function):
    """y = drop_out(LN(drop_in(act_in(x)) + res) * gamma + beta) * mask."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, cfg):
        x = x.contiguous()
        res = res.contiguous() if res is not None else None
        fused_in = res is not None or cfg.act_in is not None or cfg.drop_in > 0
        s_in = (cfg.drop_in, next_seed()) if cfg.drop_in > 0 else (0.0, 0)
        s_out = (cfg.drop_out, next_seed()) if cfg.drop_out > 0 else (0.0, 0)
        g, b = _f32c(gamma).reshape(-1), _f32c(beta).reshape(-1)
        y, mean, rstd, xsum = ops.layernorm_fwd(x, g, b, cfg.eps, res=res, lengths=cfg.lengths, out_mask=cfg.out_mask,
                                                save_stats=True, save_sum=fused_in, act_in=cfg.act_in, drop_in=s_in,
                                                drop_out=s_out)
        ctx.cfg, ctx.s_in, ctx.s_out, ctx.has_res, ctx.fused_in = cfg, s_in, s_out, res is not None, fused_in
        ctx.gshape = gamma.shape
        ctx.direct = None
        if any(ctx.needs_input_grad) and _sink(gamma) is not None and _sink(beta) is not None:
            ctx.direct = (gamma, beta)
            _use(gamma)
            _use(beta)
        ctx.save_for_backward(x, xsum if fused_in else None, g, mean, rstd)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, xsum, g, mean, rstd = ctx.saved_tensors
        cfg = ctx.cfg
        want_dz = cfg.act_in is not None or cfg.drop_in > 0
        pg, pb = ctx.direct if ctx.direct is not None else (None, None)
        dsum, dz, dg, db = ops.layernorm_bwd(dy, xsum if ctx.fused_in else x, g, mean, rstd, cfg.lengths, cfg.out_mask,
                                             z=x if cfg.act_in is not None else None, act_in=cfg.act_in,
                                             drop_in=ctx.s_in, drop_out=ctx.s_out, want_dz=want_dz,
                                             dgamma_out=pg.grad if pg is not None else None,
                                             dbeta_out=pb.grad if pb is not None else None)
        if pg is not None:
            _done(pg)
            _done(pb)
            dg = db = None
        else:
            dg, db = dg.view(ctx.gshape), db.view(ctx.gshape)
        return (dz if want_dz else dsum), (dsum if ctx.has_res else None), dg, db, None


def layer_norm(x, gamma, beta, eps, res=None, lengths=None, out_mask=False, act_in=None, drop_in=0.0, drop_out=0.0):
    cfg = SimpleNamespace(eps=eps, lengths=lengths, out_mask=out_mask, act_in=act_in, drop_in=drop_in, drop_out=drop_out)
    return LayerNormFn.apply(x, res, gamma, beta, cfg)


# ----------------------------------------------------------------------------
# A stack of [Conv1d -> LayerNorm] layers as ONE autograd node issued by two C calls (ptpp_conv_ln_stack_fwd / _bwd):
# the variance predictors (modules/variance_adaptor.py:23-62) and the frame prior network (modules/frame_prior.py:76-89).
# Same launches, seeds and order as the chain of Conv1dFn / LayerNormFn nodes it replaces (bit-identical with
# BATCHED_WGRAD off); with it on, the layers' weight gradients are one batched launch after the stack's backward.
# ----------------------------------------------------------------------------
def _u64_table(vals):
    return (ctypes.c_uint64 * len(vals))(*vals)


class ConvLnStackFn(Function):
    @staticmethod
    def forward(ctx, x, cfg, *flat):
        n = len(flat) // 4
        ws_, bs_, gs_, be_ = flat[:n], flat[n : 2 * n], flat[2 * n : 3 * n], flat[3 * n :]
        x = x.contiguous()
        B, T, C = x.shape
        dt, dev = x.dtype, x.device
        fused_in = cfg.ln_res or cfg.act_in is not None or cfg.drop_in > 0
        seeds = []
        for _ in range(n):  # the draws of LayerNormFn.forward, in its order
            seeds += [next_seed() if cfg.drop_in > 0 else 0, next_seed() if cfg.drop_out > 0 else 0]
        x_all = torch.empty((n, B, T, C), device=dev, dtype=dt)
        z_all = torch.empty((n, B, T, C), device=dev, dtype=dt)
        sum_all = torch.empty((n, B, T, C), device=dev, dtype=dt) if fused_in else None
        stats = torch.empty((2, n, B * T), device=dev, dtype=torch.float32)
        lens = ops.i32(cfg.lengths, dev) if cfg.lengths is not None else None
        gam = [_f32_param(g).reshape(-1) for g in gs_]
        bet = [_f32_param(b).reshape(-1) for b in be_]
        bias = [_f32_param(b) for b in bs_]
        a = _lib.ConvLnFwdArgs()
        a.x0 = x.data_ptr()
        a.lengths = lens.data_ptr() if lens is not None else None
        wst = [rt_stream(w, x, C, cfg.ks, 1, cfg.conv_act) for w in ws_]  # (frame-level stacks: the row-tile conv kernel)
        rt = all(t is not None for t in wst)
        # (with the stream every conv of the stack takes the row-tile kernel -- same rule on both sides -- so the mode-0 operand
        #  is never read: it is not packed at all, which keeps the per-step batched repack at its old size)
        tabs = [_ptr_table(wst if rt else [packed(w, dt) for w in ws_]), _ptr_table(bias), _ptr_table(gam), _ptr_table(bet)]
        a.wp, a.bias, a.gamma, a.beta = [ctypes.cast(t, ctypes.c_void_p) for t in tabs]
        if rt:
            a.wstream = a.wp
        a.x_all, a.z_all = x_all.data_ptr(), z_all.data_ptr()
        a.sum_all = sum_all.data_ptr() if sum_all is not None else None
        a.mean_all, a.rstd_all = stats[0].data_ptr(), stats[1].data_ptr()
        sd = _u64_table(seeds)
        a.seeds = ctypes.cast(sd, ctypes.c_void_p)
        if not torch.cuda.is_current_stream_capturing():
            wsb = ops.workspace(dev)
            a.ws, a.ws_bytes = wsb.data_ptr(), wsb.numel()
        a.eps, a.drop_in, a.drop_out = cfg.eps, cfg.drop_in, cfg.drop_out
        a.B, a.T, a.C, a.n, a.ks = B, T, C, n, cfg.ks
        a.conv_act, a.conv_mask, a.ln_res = ops._ACT[cfg.conv_act], int(cfg.conv_mask), int(cfg.ln_res)
        a.act_in, a.out_mask, a.dtype = ops._ACT[cfg.act_in], cfg.out_mask, ops.dtype_code(dt)
        _lib.check(_lib.load().ptpp_conv_ln_stack_fwd(ctypes.byref(a), ops._stream()), "ptpp_conv_ln_stack_fwd")
        ctx.cfg, ctx.n, ctx.seeds, ctx.params = cfg, n, seeds, flat
        ctx.slabs = (x, x_all, z_all, sum_all, stats, gam)
        # (no backward will run under no_grad / eval: counting uses there would leave them pending until zero_grad)
        ctx.direct = any(ctx.needs_input_grad) and all(_sink(t) is not None for t in flat)
        if ctx.direct:
            for t in flat:
                _use(t)
        return x_all[n - 1]

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        cfg, n, flat = ctx.cfg, ctx.n, ctx.params
        x, x_all, z_all, sum_all, stats, gam = ctx.slabs
        ws_, bs_, gs_, be_ = flat[:n], flat[n : 2 * n], flat[2 * n : 3 * n], flat[3 * n :]
        B, T, C = x.shape
        dt, dev = x.dtype, x.device
        gy = gy.contiguous()
        need_gx = ctx.needs_input_grad[0]
        gx = torch.empty_like(x) if need_gx else None
        gz_all = torch.empty((n, B, T, C), device=dev, dtype=dt)
        tmp = torch.empty((4, B, T, C), device=dev, dtype=dt)
        if ctx.direct:
            tg = [[t.grad for t in grp] for grp in (ws_, bs_, gs_, be_)]
        else:
            tg = [[torch.zeros(t.shape, device=dev, dtype=torch.float32) for t in grp] for grp in (ws_, bs_, gs_, be_)]
        d = _direct
        side_h = None
        if ctx.direct and d["async"] and not torch.cuda.is_current_stream_capturing():
            if d["side_h"] is None:
                create_side_stream(dev)
            side_h = d["side_h"]
        main_h = ops._stream()
        ws_main = ops.workspace(dev)
        ws_side = ops.workspace_of(dev, side_h) if side_h is not None else ws_main
        lens = ops.i32(cfg.lengths, dev) if cfg.lengths is not None else None
        a = _lib.ConvLnBwdArgs()
        a.gy, a.x0, a.x_all, a.z_all = gy.data_ptr(), x.data_ptr(), x_all.data_ptr(), z_all.data_ptr()
        a.sum_all = sum_all.data_ptr() if sum_all is not None else None
        a.mean_all, a.rstd_all = stats[0].data_ptr(), stats[1].data_ptr()
        a.lengths = lens.data_ptr() if lens is not None else None
        wst = [rt_stream(w, gy, w.shape[1], cfg.ks, 1, None, transposed=True) for w in ws_]
        rt = all(t is not None for t in wst)
        tabs = [_ptr_table(wst if rt else [packed(w, dt, mode=1) for w in ws_]), _ptr_table(gam)] + [_ptr_table(t) for t in tg]
        a.wpt, a.gamma, a.dw, a.db, a.dgamma, a.dbeta = [ctypes.cast(t, ctypes.c_void_p) for t in tabs]
        if rt:
            a.wstream_t = a.wpt
        a.gz_all, a.tmp = gz_all.data_ptr(), tmp.data_ptr()
        a.gx = gx.data_ptr() if gx is not None else None
        sd = _u64_table(ctx.seeds)
        a.seeds = ctypes.cast(sd, ctypes.c_void_p)
        a.red_scratch, a.red_bytes = ops.reduction_scratch(dev)
        a.ws_main, a.ws_main_bytes = ws_main.data_ptr(), ws_main.numel()
        a.ws_side, a.ws_side_bytes = ws_side.data_ptr(), ws_side.numel()
        a.side_stream = side_h
        a.drop_in, a.drop_out = cfg.drop_in, cfg.drop_out
        a.B, a.T, a.C, a.n, a.ks = B, T, C, n, cfg.ks
        a.conv_act, a.conv_mask, a.ln_res = ops._ACT[cfg.conv_act], int(cfg.conv_mask), int(cfg.ln_res)
        a.act_in, a.out_mask, a.dtype = ops._ACT[cfg.act_in], cfg.out_mask, ops.dtype_code(dt)
        a.batched_wgrad = int(BATCHED_WGRAD)
        with ops.red_immediate(not ctx.direct):
            _lib.check(_lib.load().ptpp_conv_ln_stack_bwd(ctypes.byref(a), main_h), "ptpp_conv_ln_stack_bwd")
        if side_h is not None:
            d["keep"].extend((x, x_all, gz_all))
        ctx.slabs = None
        if ctx.direct:
            for t in flat:
                _done(t)
            return (gx, None) + (None,) * (4 * n)
        grads = [g.view_as(p) for grp, ps in zip(tg, (ws_, bs_, gs_, be_)) for g, p in zip(grp, ps)]
        return (gx, None, *grads)


def conv_ln_stack(x, convs, norms, ks, eps, lengths, conv_act=None, conv_mask=False, ln_res=False, act_in=None, drop_in=0.0,
                  drop_out=0.0, out_mask=0):
    """convs: nn.Conv1d modules (C -> C, kernel ks, same padding); norms: modules with gamma / beta.  out_mask: 0 | 1 (every
    layer) | 2 (last layer).  See ptpp_conv_ln_stack_fwd (include/ptpp.h)."""
    cfg = SimpleNamespace(ks=ks, eps=float(eps), lengths=lengths, conv_act=conv_act, conv_mask=conv_mask, ln_res=ln_res, act_in=act_in,
                          drop_in=float(drop_in), drop_out=float(drop_out), out_mask=int(out_mask))
    flat = [c.weight for c in convs] + [c.bias for c in convs] + [m.gamma for m in norms] + [m.beta for m in norms]
    return ConvLnStackFn.apply(x, cfg, *flat)


def conv_ln_stack_ok(x, convs):
    """The one-call stack serves device tensors whose channel count fills whole 16-byte operand chunks, convs with bias."""
    C = x.shape[-1]
    return STACK_DRIVERS and x.is_cuda and C % _kc(x.dtype) == 0 and all(c.bias is not None and c.weight.shape[0] == C and
                                                                         c.weight.shape[1] == C for c in convs)


# ----------------------------------------------------------------------------
class AttentionFn(Function):
    """Relative-position MHA core on a fused (B, T, 3C) q|k|v projection."""

    @staticmethod
    def forward(ctx, qkv, pos, bias_u, bias_v, lengths, heads, variant, drop_p=0.0):
        B, T, C3 = qkv.shape
        C = C3 // 3
        q, k, v = qkv[:, :, :C], qkv[:, :, C : 2 * C], qkv[:, :, 2 * C :]
        u, vb = _f32c(bias_u), _f32c(bias_v)
        need_bwd = any(ctx.needs_input_grad)
        seed = next_seed() if drop_p > 0 else 0
        octx, probs = ops.attention_fwd(q, k, v, pos, u, vb, lengths, heads, variant, save_probs=need_bwd, drop_p=drop_p,
                                        drop_seed=seed)
        ctx.heads, ctx.variant, ctx.lengths, ctx.drop = heads, variant, lengths, (drop_p, seed)
        ctx.ushape = bias_u.shape if bias_u is not None else None
        ctx.direct = None
        if need_bwd and variant in ("new", "legacy") and _sink(bias_u) is not None and _sink(bias_v) is not None:
            ctx.direct = (bias_u, bias_v)
            _use(bias_u)
            _use(bias_v)
        ctx.save_for_backward(qkv, pos, u, vb, probs)
        return octx

    @staticmethod
    @once_differentiable
    def backward(ctx, dctx):
        qkv, pos, u, vb, probs = ctx.saved_tensors
        B, T, C3 = qkv.shape
        C = C3 // 3
        dqkv = torch.empty_like(qkv)
        pu, pv = ctx.direct if ctx.direct is not None else (None, None)
        dpos, du, dvb = ops.attention_bwd(qkv[:, :, :C], qkv[:, :, C : 2 * C], qkv[:, :, 2 * C :], pos, u, vb, probs,
                                          dctx, ctx.lengths, ctx.heads, ctx.variant, dqkv[:, :, :C],
                                          dqkv[:, :, C : 2 * C], dqkv[:, :, 2 * C :],
                                          du_out=pu.grad if pu is not None else None,
                                          dvb_out=pv.grad if pv is not None else None, drop_p=ctx.drop[0],
                                          drop_seed=ctx.drop[1])
        if dpos is not None:
            dpos = dpos.to(pos.dtype)
            du, dvb = du.view(ctx.ushape), dvb.view(ctx.ushape)
        if pu is not None:
            _done(pu)
            _done(pv)
            du = dvb = None
        return dqkv, dpos, du, dvb, None, None, None, None


class AttentionWinFn(Function):
    """Windowed relative-position MHA core (reference modules/transformer.py:59-137) on a fused (B, T, 3C) q|k|v projection."""

    @staticmethod
    def forward(ctx, qkv, emb_k, emb_v, lengths, heads, window, drop_p=0.0):
        B, T, C3 = qkv.shape
        C = C3 // 3
        ek, ev = _f32c(emb_k), _f32c(emb_v)
        need_bwd = any(ctx.needs_input_grad)
        seed = next_seed() if drop_p > 0 else 0
        octx, probs = ops.attention_win_fwd(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], ek, ev, lengths, heads, window,
                                            save_probs=need_bwd, drop_p=drop_p, drop_seed=seed)
        ctx.heads, ctx.window, ctx.lengths, ctx.drop = heads, window, lengths, (drop_p, seed)
        ctx.save_for_backward(qkv, ek, ev, probs)
        return octx

    @staticmethod
    @once_differentiable
    def backward(ctx, dctx):
        qkv, ek, ev, probs = ctx.saved_tensors
        C = qkv.shape[2] // 3
        dqkv = torch.empty_like(qkv)
        dek, dev_ = ops.attention_win_bwd(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], ek, ev, probs, dctx, ctx.lengths, ctx.heads,
                                          ctx.window, dqkv[:, :, :C], dqkv[:, :, C:2 * C], dqkv[:, :, 2 * C:], drop_p=ctx.drop[0],
                                          drop_seed=ctx.drop[1])
        return dqkv, dek, dev_, None, None, None, None


def attention_window(qkv, emb_k, emb_v, lengths, heads, window, drop_p=0.0):
    """emb_k / emb_v: (2 * window + 1, C / heads) relative-position tables (shared by the heads)."""
    return AttentionWinFn.apply(qkv, emb_k, emb_v, lengths, heads, window, drop_p)


def attention(qkv, pos, bias_u, bias_v, lengths, heads, variant, drop_p=0.0):
    """``drop_p``: dropout on the attention probabilities (the plain / BERT variant in train mode)."""
    return AttentionFn.apply(qkv, pos, bias_u, bias_v, lengths, heads, variant, drop_p)


# ----------------------------------------------------------------------------
FUSED_L1 = not os.environ.get("PTPP_NO_FUSED_L1")  # (A/B knob: the tensor-op form of the L1 losses)


class MaskedL1MeanFn(Function):
    """sum |pred - target| * mask[row] / denom / scale: an L1 loss of the training step (reference
    models/prompttts_mdn_v2_final/model.py:126, 138-170) as ONE launch forward and one backward instead of six tensor ops and
    ~eight native autograd nodes.  Gradient for ``pred`` only (the targets are data)."""

    @staticmethod
    def forward(ctx, pred, target, mask, denom, scale):
        pred = pred.contiguous()
        target = target.contiguous()
        ctx.save_for_backward(pred, target, mask, denom)
        ctx.scale = float(scale)
        return ops.l1_masked_mean_fwd(pred, target, mask, denom, scale)

    @staticmethod
    def backward(ctx, gout):
        pred, target, mask, denom = ctx.saved_tensors
        return ops.l1_masked_mean_bwd(pred, target, mask, denom, gout.float(), ctx.scale), None, None, None, None


def masked_l1_mean(pred, target, mask, denom, scale=1.0):
    """|pred - target| summed over everything (rows weighted by ``mask`` (rows,) f32 if given), divided by ``denom`` (a 0-dim
    tensor) and ``scale``.  Device tensors: one fused launch each way (MaskedL1MeanFn); CPU tensors: the tensor expression."""
    if (FUSED_L1 and pred.is_cuda and pred.dtype in (torch.float32, torch.bfloat16) and denom.dtype == torch.float32
            and target.shape == pred.shape and not (target.requires_grad and torch.is_grad_enabled())):  # (the kernel pair
        # differentiates pred only and reads target element for element: anything else takes the tensor expression)
        return MaskedL1MeanFn.apply(pred, target.float(), None if mask is None else mask.reshape(-1).float().contiguous(),
                                    denom.reshape(()), scale)
    d = (pred.float() - target.float()).abs()
    if mask is not None:
        d = d * mask.reshape(d.shape[:-1] + (1,))
    return d.sum() / denom / scale


class LengthRegulateFn(Function):
    @staticmethod
    def forward(ctx, x, cum, Tf):
        ctx.cum, ctx.Tp = cum, x.shape[1]
        return ops.length_regulate_fwd(x.contiguous(), cum, Tf)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return ops.length_regulate_bwd(dy, ctx.cum, ctx.Tp), None, None


def length_regulate(x, durations, Tf):
    """x: (B, Tp, C); durations: (B, Tp) integer-valued (float or int) frames per
    phone -> (B, Tf, C), frame f copying its phone (rows past the total are 0)."""
    if durations.is_cuda:
        cum = ops.durations_cumsum(durations)
    else:
        cum = torch.cumsum(durations.to(torch.int64), dim=1).clamp_(max=2**31 - 1).to(torch.int32).contiguous()
    return LengthRegulateFn.apply(x, cum, int(Tf))


class PosEncFn(Function):
    @staticmethod
    def forward(ctx, x, pe, scale, drop_p):
        ctx.scale, ctx.p = scale, drop_p
        ctx.seed = next_seed() if drop_p > 0 else 0
        return ops.posenc(x.contiguous(), pe, scale, drop_p, ctx.seed)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return ops.epilogue_bwd(dy, None, None, ctx.scale, False, False, ctx.p, ctx.seed), None, None, None


def posenc(x, pe, scale, drop_p=0.0):
    return PosEncFn.apply(x, pe, scale, drop_p)


# ----------------------------------------------------------------------------
# Training-step glue as single launches (csrc/glue.hip)
# ----------------------------------------------------------------------------
class TtsLossesFn(Function):
    """Every loss of the training step (reference models/prompttts_mdn_v2_final/model.py:126-183) as ONE launch forward and ONE
    backward: masked L1 of the decoder, L1 of the log-F0 / V-UV tracks, the two mixture NLLs (log-softmax of the raw MDN head
    outputs included) and their weighted sum.  Returns (total (), comps (7,): dec, dur, cf0, vuv, style, n_frames, n_phones)."""

    @staticmethod
    def forward(ctx, pred, pv, y_dur, y_sty, noise, flen, cf0_t, vuv_t, dur, plen, sty_t, cfg):
        pred, pv, y_dur, y_sty = pred.contiguous(), pv.contiguous(), y_dur.contiguous(), y_sty.contiguous()
        total, comps, nll_dur, nll_sty = ops.tts_losses_fwd(pred, noise, flen, pv, cf0_t, vuv_t, y_dur, dur, plen, y_sty, sty_t,
                                                            cfg.G_dur, cfg.G_sty, cfg.dec_scale)
        ctx.cfg = cfg
        ctx.set_materialize_grads(False)  # (the components are logged, not differentiated: no zero-filled gradient for them)
        ctx.save_for_backward(pred, noise, flen, pv, cf0_t, vuv_t, y_dur, dur, plen, y_sty, sty_t, comps, nll_dur, nll_sty)
        return total, comps

    @staticmethod
    @once_differentiable
    def backward(ctx, g_total, g_comps):
        cfg = ctx.cfg
        gt = g_total.contiguous().float() if g_total is not None else None
        gc = g_comps.contiguous().float() if g_comps is not None else None
        dpred, dpv, dy_dur, dy_sty = ops.tts_losses_bwd(ctx.saved_tensors, gt, gc, cfg.G_dur, cfg.G_sty, cfg.dec_scale)
        return dpred, dpv, dy_dur, dy_sty, None, None, None, None, None, None, None, None


def tts_losses(pred, pv, y_dur, y_sty, noise, flen, cf0_t, vuv_t, dur, plen, sty_t, G_dur, G_sty, dec_scale):
    cfg = SimpleNamespace(G_dur=G_dur, G_sty=G_sty, dec_scale=float(dec_scale))
    return TtsLossesFn.apply(pred, pv, y_dur, y_sty, noise, flen, cf0_t, vuv_t, dur, plen, sty_t, cfg)


class MishFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.mish_fwd(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        return ops.mish_bwd(ctx.saved_tensors[0], gy)


def mish(x):
    """x * tanh(softplus(x)) (modules/denoiser.py:23-26): one launch each way for f32 device tensors."""
    if x.is_cuda and x.dtype == torch.float32:
        return MishFn.apply(x)
    return x * torch.tanh(torch.nn.functional.softplus(x))


class EmbedClFn(Function):
    """PhonemeEmbedding (layers/embedding.py:21-36): lookup, optional sqrt(C) scale, phone mask, cast -- one launch; the table
    gradient by one block per vocabulary entry in row order (bit-reproducible), straight into ``weight.grad`` when the trainer
    laid the gradients out flat."""

    @staticmethod
    def forward(ctx, ids, weight, lengths, scale, dtype, padding_idx):
        ids = ids.contiguous()
        ctx.ids, ctx.lengths, ctx.scale, ctx.padding_idx = ids, lengths, scale, padding_idx
        ctx.weight = weight
        ctx.direct = ctx.needs_input_grad[1] and _sink(weight) is not None
        if ctx.direct:
            _use(weight)
        return ops.embed_cl_fwd(ids, _f32c(weight), lengths, scale, dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        w = ctx.weight
        if ctx.direct:
            ops.embed_cl_bwd(ctx.ids, dout, ctx.lengths, ctx.scale, w.grad, ctx.padding_idx)
            _done(w)
            return None, None, None, None, None, None
        dt = torch.zeros(w.shape, device=dout.device, dtype=torch.float32)
        ops.embed_cl_bwd(ctx.ids, dout, ctx.lengths, ctx.scale, dt, ctx.padding_idx)
        return None, dt, None, None, None, None


def embed_cl(ids, weight, lengths, scale, dtype, padding_idx):
    return EmbedClFn.apply(ids, weight, lengths, scale, dtype, padding_idx)


class ScalarEmbedAddFn(Function):
    """x + Conv1d(1 -> C, k = 1)(track) * mask (modules/variance_adaptor.py:139-146): one launch forward; backward = the
    identity for x plus one column-sum launch for (dw, db) whose finishing step is deferrable."""

    @staticmethod
    def forward(ctx, x, track, weight, bias, lengths):
        x, track = x.contiguous(), track.contiguous().float()
        ctx.track, ctx.lengths, ctx.params = track, lengths, (weight, bias)
        ctx.direct = _sink(weight) is not None and _sink(bias) is not None
        if ctx.direct:
            _use(weight)
            _use(bias)
        return ops.scalar_embed_add(x, track, _f32c(weight).reshape(-1), _f32c(bias), lengths)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        w, b = ctx.params
        dout = dout.contiguous()
        if ctx.direct:
            ops.scalar_embed_bwd(dout, ctx.track, ctx.lengths, w.grad.view(-1), b.grad)
            _done(w)
            _done(b)
            return dout, None, None, None, None
        dw = torch.zeros(w.numel(), device=dout.device, dtype=torch.float32)
        db = torch.zeros_like(dw)
        with ops.red_immediate():
            ops.scalar_embed_bwd(dout, ctx.track, ctx.lengths, dw, db)
        return dout, None, dw.view_as(w), db, None


def scalar_embed_add(x, track, emb, lengths):
    """``emb``: nn.Conv1d(1, C, 1).  The track is data (teacher forcing) or a detached prediction: no gradient flows into it
    here -- callers whose track needs one use the tensor expression."""
    return ScalarEmbedAddFn.apply(x, track, emb.weight, emb.bias, lengths)


class LinearSmallFn(Function):
    """A 1x1 Conv1d / Linear with 1-4 output channels and an output mask (the pitch / V-UV head, reference
    modules/variance_adaptor.py:52-62) as one launch forward and one backward (f32 weights and accumulation)."""

    @staticmethod
    def forward(ctx, x, weight, bias, lengths):
        x = x.contiguous()
        w = _f32c(weight).reshape(weight.shape[0], -1)
        ctx.lengths, ctx.params = lengths, (weight, bias)
        ctx.direct = any(ctx.needs_input_grad) and _sink(weight) is not None and bias is not None and _sink(bias) is not None
        if ctx.direct:
            _use(weight)
            _use(bias)
        ctx.save_for_backward(x, w)
        return ops.linear_small_fwd(x, w, _f32c(bias), lengths)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        weight, bias = ctx.params
        if ctx.direct:
            dx = ops.linear_small_bwd(x, dy, w, ctx.lengths, weight.grad.view(-1), bias.grad, want_dx=ctx.needs_input_grad[0])
            _done(weight)
            _done(bias)
            return dx, None, None, None
        dw = torch.zeros(weight.numel(), device=x.device, dtype=torch.float32)
        db = torch.zeros(weight.shape[0], device=x.device, dtype=torch.float32)
        with ops.red_immediate():
            dx = ops.linear_small_bwd(x, dy, w, ctx.lengths, dw, db, want_dx=ctx.needs_input_grad[0])
        return dx, dw.view_as(weight), (db if bias is not None else None), None


def linear_small_ok(x, weight):
    return (x.is_cuda and x.dim() == 3 and weight.shape[0] <= 4 and x.shape[-1] % 256 == 0 and x.shape[-1] <= 1024
            and x.dtype in (torch.float32, torch.bfloat16) and weight.numel() == weight.shape[0] * x.shape[-1])


def linear_small(x, weight, bias, lengths):
    return LinearSmallFn.apply(x, weight, bias, lengths)


class RejoinBranchFn(Function):
    """The JOIN of a forward branch that was issued early on its own stream (the reference encoder: mel -> style embedding).
    autograd runs ready nodes newest-first, so a branch whose nodes were created BEFORE the trunk's is differentiated LAST -- after
    the whole phoneme-encoder backward, with the main stream idle for its ~1.7 ms of launches at the end of every step
    (profiles/r06_step_tail_before.txt).  This node is created at the join instead; its backward runs the branch's own graph by a
    re-entrant backward on the branch's stream as soon as the joined value's gradient exists, beside the trunk's backward."""

    @staticmethod
    def forward(ctx, _anchor, holder, stream):
        # (the branch output travels in a holder: as a tensor ARGUMENT it would become an input edge of this node, and the outer
        #  pass would run the branch's graph a second time with an undefined gradient)
        ctx.inner, ctx.stream = holder[0], stream
        return holder[0].detach()

    @staticmethod
    def backward(ctx, g):
        inner, st = ctx.inner, ctx.stream
        ctx.inner = None
        if st is None:
            torch.autograd.backward(inner, g)
            return None, None, None
        st.wait_stream(torch.cuda.current_stream())
        g.record_stream(st)
        with ops.unpinned(), torch.cuda.stream(st):  # (the caller stream of the re-entrant pass: its end-of-pass join stays here)
            torch.autograd.backward(inner, g)
        return None, None, None


_rejoin_anchor = {}


def rejoin_branch(inner, stream=None):
    """``inner``: a branch's output (its autograd graph hangs off it).  Returns a tensor with the same values whose gradient
    is propagated into the branch when it arrives (see RejoinBranchFn)."""
    if not (inner.requires_grad and torch.is_grad_enabled()):
        return inner
    key = str(inner.device)
    a = _rejoin_anchor.get(key)
    if a is None:  # an empty leaf that requires grad: makes autograd record the node (its own gradient is None)
        a = _rejoin_anchor[key] = torch.empty(0, device=inner.device, requires_grad=True)
    return RejoinBranchFn.apply(a, (inner,), stream)


class L2NormFn(Function):
    @staticmethod
    def forward(ctx, x, eps):
        y, n = ops.l2norm_fwd(x.contiguous(), eps)
        ctx.eps = eps
        ctx.save_for_backward(y, n)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        y, n = ctx.saved_tensors
        return ops.l2norm_bwd(y, n, gy, ctx.eps), None


def l2_normalize_channels(e, eps=1e-12):
    """F.normalize(e, dim=1) of a (B, C, 1) f32 device tensor as one launch each way."""
    if e.is_cuda and e.dtype == torch.float32 and e.dim() == 3 and e.shape[2] == 1:
        return L2NormFn.apply(e.reshape(e.shape[0], e.shape[1]), eps).unsqueeze(-1)
    return torch.nn.functional.normalize(e, dim=1, eps=eps)


class BcastAddRowsFn(Function):
    @staticmethod
    def forward(ctx, x, e):
        ctx.need_e = ctx.needs_input_grad[1]
        return ops.bcast_add_rows(x.contiguous(), e.contiguous())

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return dy, (ops.rows_sum(dy) if ctx.need_e else None)


def bcast_add_rows(x, e):
    """x (B, T, C) + e (B, C) f32 on every row (model.py:111), e rounded to x's dtype first like ``e.to(x.dtype)``."""
    return BcastAddRowsFn.apply(x, e)


# ----------------------------------------------------------------------------
# DiffNet residual stack (modules/denoiser.py:69-83,136-140) with a hand-written
# backward: per layer  conv(k3,dilated, +cond slice) -> gate -> 1x1 conv -> post.
# ----------------------------------------------------------------------------
def diffnet_fused_gate(dtype):
    """Inference in bf16 / f16 fuses the DiffNet gate into the dilated conv's epilogue (ptpp.h, PTPP_ACT_GATE; f16: inside the
    one-launch layer only, csrc/diffnet_layer.hip)."""
    return dtype in (torch.bfloat16, torch.float16)


_gate_perm_cache = {}


def _gate_perm(c2, device):
    """Row order [4 gate | 4 filter] per group of 8 for a 2C-channel (gate | filter) tensor."""
    key = (c2, str(device))
    p = _gate_perm_cache.get(key)
    if p is None:
        C = c2 // 2
        k = torch.arange(C // 4, device=device)[:, None] * 4 + torch.arange(4, device=device)[None, :]  # (C/4, 4)
        p = _gate_perm_cache[key] = torch.cat([k, k + C], dim=1).reshape(-1)
    return p


FUSE_GATE_SAVE = not os.environ.get("PTPP_NO_FUSED_GATE_SAVE")  # training: gate in the dilated conv's epilogue, a kept


def diffnet_gate_save(dtype, C, cuda):
    """Training forward in bf16: the DiffNet gate runs in the dilated conv's epilogue and the pre-activation is kept for the
    backward (ptpp_conv1d_gate_fwd_save) -- one launch per layer less than conv + gate_fwd, bit-identical to them."""
    return FUSE_GATE_SAVE and cuda and dtype == torch.bfloat16 and ops.conv1d_gate_fwd_save_supported(C, C, dtype)


def gate_biases(dil_bs, cond_bs):
    """Dilated-conv and conditioner biases of all layers in the gate-interleaved order: ONE cat + ONE gather per step
    (the packed weights get that order from the pack cache, mode 2).  Returns (per-layer dil biases, concatenated cond bias)."""
    L, c2 = len(dil_bs), dil_bs[0].shape[0]
    dev = dil_bs[0].device
    key = ("gateb", L, c2, str(dev))
    idx = _gate_perm_cache.get(key)
    if idx is None:
        perm = _gate_perm(c2, dev)
        idx = _gate_perm_cache[key] = (torch.arange(2 * L, device=dev)[:, None] * c2 + perm[None, :]).reshape(-1)
    allb = bias_cat(list(dil_bs) + list(cond_bs))[idx]
    return [allb[l * c2:(l + 1) * c2] for l in range(L)], allb[L * c2:]


def diffnet_cond_all(cond, cond_ws, cond_bs, gate_perm=False, bias_perm=None):
    """All layers' conditioner projections as ONE GEMM: (B,T,Cc) -> (B,T,L*2C).  ``gate_perm``: each
    layer's 2C channels in the interleaved order of the fused gate epilogue (inference; training with ``bias_perm`` =
    the already permuted concatenated bias of ``gate_biases``: the weights then come from the pack cache, mode 2)."""
    cond_ws, cond_bs = list(cond_ws), list(cond_bs)
    rows = sum(cw.shape[0] for cw in cond_ws)
    if bias_perm is not None:
        return ops.conv1d(cond, packed_cat(cond_ws, cond.dtype, mode=2), bias_perm, rows), cond_ws
    if not gate_perm:
        return ops.conv1d(cond, packed_cat(cond_ws, cond.dtype), bias_cat(cond_bs), rows), cond_ws
    perm = _gate_perm(cond_ws[0].shape[0], cond.device)
    wp = _cat_cached(cond_ws, ("wg", cond.dtype), lambda: ops.pack_conv_weight(
        torch.cat([w.detach().reshape(w.shape[0], -1)[perm] for w in cond_ws], dim=0), cond.dtype))
    bp = _cat_cached(cond_bs, "bg", lambda: torch.cat([b.detach().float()[perm] for b in cond_bs], dim=0))
    return ops.conv1d(cond, wp, bp, rows), cond_ws


FUSE_DIFFNET_POST = not os.environ.get("PTPP_NO_FUSED_POST")  # (tests compare the fused launch with the two-kernel path)


# the weight gradients of a stack's layers as one batched launch per shape after the stack's backward (no split-K partials,
# bit-reproducible) instead of one launch (+ reduce) per layer inside it -- only inside the one-call drivers
BATCHED_WGRAD = not os.environ.get("PTPP_NO_BATCHED_WGRAD")
STACK_DRIVERS = not os.environ.get("PTPP_NO_STACK_DRIVERS")  # (tests compare the one-call drivers with the per-launch path)


def _ptr_table(tensors):
    """HOST array of device pointers (an argument of the whole-unit drivers, include/ptpp.h)."""
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


def _f32_param(t):
    """A parameter the kernels read as f32: the tensor itself when it already is contiguous f32 (the usual case)."""
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.detach().float().contiguous()


DIFFNET_LAYER_KERNEL = not os.environ.get("PTPP_NO_DIFFNET_LAYER_KERNEL")  # (tests compare with the two-launch path)


DIFFNET_FOLD_COND = not os.environ.get("PTPP_NO_DIFFNET_FOLD_COND")  # training: the conditioner projection inside the layer launch


def diffnet_wstream(weights, dil_wp, out_wp, dt, cond_ws=None):
    """The operand stream of csrc/diffnet_layer.hip for all layers ((L, 1 MiB) bytes; ptpp_diffnet_pack_wstream): every
    layer's dilated-conv (mode 2) and output-projection operands re-laid as the 64 LDS stage images its kernel consumes,
    cached per version of the 2 L weights (one gather launch per optimiser step in training, one per checkpoint otherwise)."""
    lib = _lib.load()
    C = weights[0][2].shape[1]
    if not (DIFFNET_LAYER_KERNEL and lib.ptpp_diffnet_layer_supported(C, ops.dtype_code(dt))):
        return None
    L = len(weights)

    def make():
        cwp = None
        if cond_ws is not None:  # (2C, 256[, 1]) each -> one mode-2 operand of all layers, layer l at row l * 2C
            allc = packed_cat(cond_ws, dt, mode=2)
            per = allc.shape[0] // L
            cwp = [allc[l * per:(l + 1) * per] for l in range(L)]
        return ops.diffnet_pack_wstream(dil_wp, out_wp, C, cond_wps=cwp)

    srcs = [w[0] for w in weights] + [w[2] for w in weights] + (list(cond_ws) if cond_ws is not None else [])
    if not all(isinstance(t, torch.nn.Parameter) for t in srcs):
        return make()
    return _cat_cached(srcs, ("dnws", dt, cond_ws is not None), make)


def _diffnet_stack_forward_driver(h0, cond_all, dsteps, weights, lengths, cycle, save, gate_b=None, scaled=False, condx=None,
                                  cond_ws=None, yin0=None):
    """The whole residual stack in ONE C call (ptpp_diffnet_stack_fwd): the same launches in the same order as the loop
    of ``diffnet_stack_forward`` below (bit-identical), without ~60 Python -> C round trips and ~80 allocations.
    Returns (skip f32, (yin_all, a_all, g_all) slabs of all layers when ``save``); with ``scaled`` the first item is
    (skip / sqrt(L)).to(dtype) instead (modules/denoiser.py:150) -- written by the last layer's launch where the one-launch
    layer serves, the same two roundings as the expression."""
    L = len(weights)
    B, T, C = h0.shape
    dt, dev = h0.dtype, h0.device
    fused = (not save) and lengths is None and diffnet_fused_gate(dt)
    gsave = save and gate_b is not None  # (cond_all is in the gate-interleaved order then)
    n_slabs = L if save else 2
    skip = torch.empty((B, T, C), device=dev, dtype=torch.float32)
    ds = dsteps.transpose(0, 1).contiguous()  # (L, B, C)
    yin_all = torch.empty((n_slabs, B, T, C), device=dev, dtype=dt)
    g_all = torch.empty((n_slabs, B, T, C), device=dev, dtype=dt)
    a_all = None if fused else torch.empty((n_slabs, B, T, 2 * C), device=dev, dtype=dt)
    xb = torch.empty((2, B, T, C), device=dev, dtype=dt)
    o_buf = None if ops.conv1d_diffnet_post_supported(C, C, dt) else torch.empty((B, T, 2 * C), device=dev, dtype=dt)
    if gsave:
        dil_wp = [packed(dw, dt, mode=2) for dw, _, _, _ in weights]
        dil_b = gate_b
    elif fused:
        perm = _gate_perm(2 * C, dev)
        dil_wp = [_cat_cached([dw], ("wg1", dt), lambda dw=dw: ops.pack_conv_weight(dw.detach()[perm], dt)) for dw, _, _, _ in weights]
        dil_b = [_cat_cached([db], "bg1", lambda db=db: db.detach().float()[perm].contiguous()) for _, db, _, _ in weights]
    else:
        dil_wp = [packed(dw, dt) for dw, _, _, _ in weights]
        dil_b = [_f32_param(db) for _, db, _, _ in weights]
    out_wp = [packed(ow, dt) for _, _, ow, _ in weights]
    out_b = [_f32_param(ob) for _, _, _, ob in weights]
    lens = ops.i32(lengths, dev) if lengths is not None else None
    assert h0.is_contiguous() and ds.dtype == torch.float32
    assert condx is not None or (cond_all.is_contiguous() and cond_all.shape[2] == L * 2 * C)
    a = _lib.DiffNetFwdArgs()
    a.h0, a.dsteps, a.skip = h0.data_ptr(), ds.data_ptr(), skip.data_ptr()
    a.cond_all = cond_all.data_ptr() if cond_all is not None else None
    a.lengths = lens.data_ptr() if lens is not None else None
    tabs = [_ptr_table(t) for t in (dil_wp, dil_b, out_wp, out_b)]
    a.dil_wp, a.dil_b, a.out_wp, a.out_b = [ctypes.cast(t, ctypes.c_void_p) for t in tabs]
    a.yin_all, a.g_all = yin_all.data_ptr(), g_all.data_ptr()
    a.a_all = a_all.data_ptr() if a_all is not None else None
    a.x_buf0, a.x_buf1 = xb[0].data_ptr(), xb[1].data_ptr()
    a.o_buf = o_buf.data_ptr() if o_buf is not None else None
    a.B, a.T, a.C, a.L, a.cycle, a.n_slabs, a.fused_gate, a.dtype = B, T, C, L, cycle, n_slabs, 2 if gsave else int(fused), ops.dtype_code(dt)
    wstream = diffnet_wstream(weights, dil_wp, out_wp, dt, cond_ws if condx is not None else None) if (gsave or fused) and cycle <= 4 else None
    a.wstream = wstream.data_ptr() if wstream is not None else None
    if condx is not None:  # the layers project the conditioner input themselves (gate_b = dilated-conv + conditioner biases)
        assert wstream is not None and condx.stride(2) == 1 and condx.shape[2] == 256 and condx.dtype == dt
        a.condx, a.ldcx = condx.data_ptr(), condx.stride(1)
    if yin0 is not None:  # layer 0's input already formed (ops.sampler_head)
        assert yin0.is_contiguous() and yin0.shape == h0.shape and yin0.dtype == dt
        a.yin0 = yin0.data_ptr()
    sc = None
    if scaled and wstream is not None:
        sc = torch.empty((B, T, C), device=dev, dtype=dt)
        a.skip_scaled, a.skip_scale = sc.data_ptr(), 1.0 / math.sqrt(L)
    _lib.check(_lib.load().ptpp_diffnet_stack_fwd(ctypes.byref(a), ops._stream()), "ptpp_diffnet_stack_fwd")
    if scaled:
        skip = sc if sc is not None else (skip * (1.0 / math.sqrt(L))).to(dt)
    return skip, ((yin_all, a_all, g_all) if save else None)


def diffnet_fold_cond_ok(h0, cond, cycle):
    """Training forward: can every layer project the conditioner input inside its own launch (csrc/diffnet_layer.hip, COND
    instantiation) instead of reading a slice of the (B, T, L * 2C) tensor of all layers' projections?  Saves that tensor's
    GEMM (0.46 ms at the bench shape), its 0.6 GB write and the 0.6 GB the layers read back per forward."""
    return (DIFFNET_FOLD_COND and DIFFNET_LAYER_KERNEL and STACK_DRIVERS and h0.is_cuda and h0.is_contiguous() and cycle <= 4
            and cond.dim() == 3 and cond.shape[2] == 256 and cond.stride(2) == 1 and cond.dtype == h0.dtype
            and diffnet_gate_save(h0.dtype, h0.shape[2], True) and ops.diffnet_layer_supported(h0.shape[2], h0.dtype))


def diffnet_stack_forward(h0, cond_all, dsteps, weights, lengths, cycle, save, gate_b=None, scaled=False, yin0=None):
    """weights: per layer (dil_w, dil_b, out_w, out_b).  Returns (skip_sum f32, saved).  ``gate_b``: the per-layer dilated-conv
    biases in the gate-interleaved order (``gate_biases``) -- training with the gate fused into the conv and the
    pre-activation kept; ``cond_all`` is in that order too.  ``scaled``: return (skip_sum / sqrt(L)).to(dtype) instead."""
    if STACK_DRIVERS and h0.is_cuda and h0.is_contiguous() and cond_all.is_contiguous():
        return _diffnet_stack_forward_driver(h0, cond_all, dsteps, weights, lengths, cycle, save, gate_b, scaled, yin0=yin0)
    assert yin0 is None, "a precomputed first-layer input is a feature of the one-call driver"
    if scaled:
        skip, saved = diffnet_stack_forward(h0, cond_all, dsteps, weights, lengths, cycle, save, gate_b)
        return (skip * (1.0 / math.sqrt(len(weights)))).to(h0.dtype), saved
    L = len(weights)
    B, T, C = h0.shape
    skip = torch.empty((B, T, C), device=h0.device, dtype=torch.float32)
    x = h0
    ds = dsteps.transpose(0, 1).contiguous()  # (L, B, C): one copy, then ds[l] are contiguous views
    _, yin = ops.diffnet_post_fwd(None, h0, None, ds[0], init=True)
    saved = []
    fused = (not save) and lengths is None and diffnet_fused_gate(h0.dtype)  # cond_all is in gate order then
    perm = _gate_perm(2 * C, h0.device) if fused else None
    fuse_post = FUSE_DIFFNET_POST and h0.is_cuda and ops.conv1d_diffnet_post_supported(C, C, h0.dtype) and h0.is_contiguous()
    for l, (dw, db, ow, ob) in enumerate(weights):
        d = 2 ** (l % cycle)
        if save and gate_b is not None:
            g = torch.empty((B, T, C), device=h0.device, dtype=h0.dtype)
            a = torch.empty((B, T, 2 * C), device=h0.device, dtype=h0.dtype)
            ops.conv1d_gate_fwd_save(yin, packed(dw, h0.dtype, mode=2), gate_b[l], C, 3, d, d,
                                     cond_all[:, :, l * 2 * C : (l + 1) * 2 * C], g, a, lengths=lengths)
        elif fused:
            wp = _cat_cached([dw], ("wg1", h0.dtype), lambda dw=dw: ops.pack_conv_weight(dw.detach()[perm], h0.dtype))
            bp = _cat_cached([db], "bg1", lambda db=db: db.detach().float()[perm].contiguous())
            g = torch.empty((B, T, C), device=h0.device, dtype=h0.dtype)
            ops.conv1d(yin, wp, bp, 2 * C, ks=3, dil=d, pad=d, act="gate", res=cond_all[:, :, l * 2 * C : (l + 1) * 2 * C],
                       out=g)
            a = None
        else:
            # (ragged batch: output past an utterance's end only meets the masked output projection -- masked here too, so
            #  that the row tiles past the end skip their K loops)
            a = ops.conv1d(yin, packed(dw, h0.dtype), _f32c(db), 2 * C, ks=3, dil=d, pad=d,
                           res=cond_all[:, :, l * 2 * C : (l + 1) * 2 * C], lengths=lengths, out_mask=lengths is not None)
            g = ops.gate_fwd(a)
        if save:
            saved.append((yin, a, g))
        nxt = ds[l + 1] if l + 1 < L else None
        if fuse_post:  # output projection + residual / skip / next-input update in one launch (o is never stored)
            x, yin = ops.conv1d_diffnet_post(g, packed(ow, h0.dtype), _f32c(ob), x, skip, nxt, init=(l == 0), lengths=lengths,
                                             out_mask=lengths is not None)
        else:
            o = ops.conv1d(g, packed(ow, h0.dtype), _f32c(ob), 2 * C, lengths=lengths, out_mask=lengths is not None)
            x, yin = ops.diffnet_post_fwd(o, x, skip, nxt, init=(l == 0))
    return skip, saved


class DiffNetStackFn(Function):
    @staticmethod
    def forward(ctx, h0, cond, dsteps, lengths, cycle, *flat):
        L = len(flat) // 6
        ws = [flat[6 * l : 6 * l + 6] for l in range(L)]  # dil_w, dil_b, cond_w, cond_b, out_w, out_b
        gate_b = None
        if diffnet_fold_cond_ok(h0, cond, cycle):
            # no (B, T, L * 2C) tensor of conditioner projections: each layer's launch projects `cond` itself (f32 accumulation
            # with the dilated conv, one rounding); its bias joins the dilated conv's
            gate_b, cond_b = gate_biases([w[1] for w in ws], [w[3] for w in ws])
            c2 = ws[0][1].shape[0]
            allb = torch.cat(gate_b) + cond_b
            wc = [w[2] for w in ws]
            skip, saved = _diffnet_stack_forward_driver(h0, None, dsteps, [(w[0], w[1], w[4], w[5]) for w in ws], lengths, cycle, True,
                                                        [allb[l * c2:(l + 1) * c2] for l in range(L)], True, condx=cond, cond_ws=wc)
        else:
            if diffnet_gate_save(h0.dtype, h0.shape[2], h0.is_cuda) and h0.is_contiguous():
                gate_b, cond_b = gate_biases([w[1] for w in ws], [w[3] for w in ws])
                cond_all, wc = diffnet_cond_all(cond, [w[2] for w in ws], [w[3] for w in ws], bias_perm=cond_b)
            else:
                cond_all, wc = diffnet_cond_all(cond, [w[2] for w in ws], [w[3] for w in ws])
            skip, saved = diffnet_stack_forward(h0, cond_all, dsteps, [(w[0], w[1], w[4], w[5]) for w in ws], lengths,
                                                cycle, save=True, gate_b=gate_b, scaled=True)
        ctx.L, ctx.cycle, ctx.lengths, ctx.saved, ctx.ws, ctx.wc = L, cycle, lengths, saved, ws, wc
        ctx.direct = any(ctx.needs_input_grad) and all(_sink(t) is not None for t in flat)
        if ctx.direct:
            for t in flat:
                _use(t)
        ctx.save_for_backward(cond)
        return skip

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        (cond,) = ctx.saved_tensors
        L, ws = ctx.L, ctx.ws
        B, T, C = gout.shape
        dt = gout.dtype
        if gout.is_cuda and dt != torch.float32:  # (one launch: float(gout) * s rounded to dt, the arithmetic of the tensor-op chain)
            gS = ops.epilogue_bwd(gout.contiguous(), None, None, 1.0 / math.sqrt(L), False, False, 0.0, 0)
        else:
            gS = (gout.float() * (1.0 / math.sqrt(L))).to(dt).contiguous()
        # every layer's gx is kept ((L + 1, B, T, C): 0.3 GB at the bench shape) so that the per-utterance column sums
        # of all layers are ONE launch after the loop instead of one (plus its memset) per layer
        gx_all = torch.empty((L + 1, B, T, C), device=gout.device, dtype=dt)
        if isinstance(ctx.saved, tuple):  # slabs of the one-call forward: the loop below as ONE C call
            S, dcond_all, grads = _diffnet_backward_driver(ctx, gS, gx_all)
            return DiffNetStackFn._backward_tail(ctx, cond, gx_all[0], S, dcond_all, grads)
        gx = gx_all[L].zero_()
        # per-layer column sums of gx land in rows of one buffer; the step-embedding gradients
        # dd[:, l] = S[l] - S[l+1] / sqrt(2) are formed after the loop in two launches (not 3 per layer)
        S = torch.zeros((L + 1, B, C), device=gout.device, dtype=torch.float32)
        dcond_all = torch.empty((B, T, L * 2 * C), device=gout.device, dtype=dt)
        grads = [None] * (6 * L)
        r2 = 1.0 / math.sqrt(2.0)
        fuse_gbwd = FUSE_DIFFNET_POST and gout.is_cuda and ops.conv1d_gate_bwd_supported(C, 2 * C, dt)
        colpart = None  # (as the driver: S from the data-gradient convs' epilogues where every layer takes the row-tile kernel)
        if gout.is_cuda and C == 256 and ops.conv1d_rt_colpart_ok(2 * C, 3, [2 ** (l % ctx.cycle) for l in range(L)], B, T) and \
                all(rt_stream(w[0], dcond_all[:, :, : 2 * C], C, 3, 2 ** (l % ctx.cycle), None, transposed=True) is not None
                    for l, w in enumerate(ws)):
            colpart = torch.empty((L, B, (T + 31) // 32, C), device=gout.device, dtype=torch.float32)
        for l in reversed(range(L)):
            yin, a, g = ctx.saved[l]
            dil_w, _, _, _, out_w, _ = ws[l]
            d = 2 ** (l % ctx.cycle)
            do = ops.diffnet_post_bwd(gx, gS, ctx.lengths)
            tg = [t.grad if ctx.direct else None for t in ws[l]]
            with wgrad_stream(*((g, do) if ctx.direct else ())):
                dwo, dbo = ops.conv1d_wgrad(g, do, C, 2 * C, 1, 1, 0, dw_out=tg[4], db_out=tg[5])
            if fuse_gbwd:  # the projection's data gradient with the gate backward in its epilogue (dg is never stored)
                da = ops.conv1d_gate_bwd(do, packed(out_w, dt, mode=1), a, dcond_all[:, :, l * 2 * C : (l + 1) * 2 * C],
                                         lengths=ctx.lengths)
            else:
                dg = ops.conv1d(do, packed(out_w, dt, mode=1), None, C, lengths=ctx.lengths, in_mask=ctx.lengths is not None)
                da = ops.gate_bwd(a, dg, dcond_all[:, :, l * 2 * C : (l + 1) * 2 * C])
            with wgrad_stream(*((yin, da) if ctx.direct else ())):
                dwd, dbd = ops.conv1d_wgrad(yin, da, C, 2 * C, 3, d, d, dw_out=tg[0], db_out=tg[1])
            # (do / da are zero past an utterance's end: the input mask is exact and lets those row tiles skip their K loops)
            wst = rt_stream(dil_w, da, C, 3, d, None, transposed=True)
            gx = ops.conv1d(da, packed(dil_w, dt, mode=1) if wst is None else None, None, C, ks=3, dil=d, pad=d, res=gx, res_scale=r2,
                            out=gx_all[l], lengths=ctx.lengths, in_mask=ctx.lengths is not None, wstream=wst,
                            colpart=colpart[l] if colpart is not None else None)
            if ctx.direct:
                for i in (0, 1, 4, 5):
                    _done(ws[l][i])
            else:
                grads[6 * l + 0], grads[6 * l + 1] = dwd.view_as(dil_w), dbd
                grads[6 * l + 4], grads[6 * l + 5] = dwo.view_as(out_w), dbo
            ctx.saved[l] = None
        if colpart is not None:
            ops.colsum_batch(colpart.view(L * B, -1, C), out=S[:L].view(L * B, C))
        else:
            ops.colsum_batch(gx_all[:L].view(L * B, T, C), out=S[:L].view(L * B, C))
        return DiffNetStackFn._backward_tail(ctx, cond, gx, S, dcond_all, grads)

    @staticmethod
    def _backward_tail(ctx, cond, gx, S, dcond_all, grads):
        L, ws = ctx.L, ctx.ws
        C = gx.shape[-1]
        dt = gx.dtype
        r2 = 1.0 / math.sqrt(2.0)
        dd = torch.sub(S[:L], S[1:], alpha=r2).transpose(0, 1)  # (B, L, C)
        dcond = None
        if ctx.needs_input_grad[1]:
            # K = L * 2C = 10 240: the longest contraction of the step.  On the row-tile engine (operand stream of the
            # concatenated projections, pack mode 4) where it qualifies: 555 -> ~150 us at the bench shape
            if ops.conv1d_rt_ok(dcond_all, cond.shape[-1], 1, 1, None) and all(isinstance(w, torch.nn.Parameter) for w in ctx.wc):
                dcond = ops.conv1d(dcond_all, None, None, cond.shape[-1], wstream=packed_cat(ctx.wc, dt, mode=4))
            else:
                dcond = ops.conv1d(dcond_all, packed_cat(ctx.wc, dt, mode=1), None, cond.shape[-1])
        if ctx.direct:
            # the fused conditioner GEMM's weight gradient in ONE launch (Cout = L*2C: 2.7x the throughput
            # of L per-layer launches), then one multi-tensor add of each layer's row block into its own
            # gradient buffer
            with wgrad_stream(cond, dcond_all, torch_ops=True):
                dwc, dbc = ops.conv1d_wgrad(cond, dcond_all, cond.shape[-1], L * 2 * C, 1, 1, 0)
                dwl = dwc.view(L, 2 * C, -1)
                dbl = dbc.view(L, 2 * C)
                torch._foreach_add_([ws[l][2].grad.view(2 * C, -1) for l in range(L)] + [ws[l][3].grad for l in range(L)],
                                    [dwl[l] for l in range(L)] + [dbl[l] for l in range(L)])
            for l in range(L):
                _done(ws[l][2])
                _done(ws[l][3])
        else:
            dwc, dbc = ops.conv1d_wgrad(cond, dcond_all, cond.shape[-1], L * 2 * C, 1, 1, 0)
            for l in range(L):
                grads[6 * l + 2] = dwc[l * 2 * C : (l + 1) * 2 * C].view_as(ws[l][2])
                grads[6 * l + 3] = dbc[l * 2 * C : (l + 1) * 2 * C]
        return (gx, dcond, dd, None, None, *grads)


def _diffnet_backward_driver(ctx, gS, gx_all):
    """ptpp_diffnet_stack_bwd: per layer post_bwd -> output-projection weight gradient (side stream) -> its data gradient
    with the gate backward fused -> dilated conv weight gradient (side stream) -> dilated conv data gradient, then the
    column sums of all layers; the launches and their order are those of the per-launch loop (bit-identical)."""
    L, ws = ctx.L, ctx.ws
    _, B, T, C = gx_all.shape
    dt, dev = gS.dtype, gS.device
    yin_all, a_all, g_all = ctx.saved
    S = torch.zeros((L + 1, B, C), device=dev, dtype=torch.float32)
    dcond_all = torch.empty((B, T, L * 2 * C), device=dev, dtype=dt)
    do_all = torch.empty((L, B, T, 2 * C), device=dev, dtype=dt)
    dg_buf = None if ops.conv1d_gate_bwd_supported(C, 2 * C, dt) else torch.empty((B, T, C), device=dev, dtype=dt)
    if ctx.direct:
        tg = [[w[i].grad for w in ws] for i in (0, 1, 4, 5)]
    else:
        tg = [[torch.zeros(w[i].shape, device=dev, dtype=torch.float32) for w in ws] for i in (0, 1, 4, 5)]
    d = _direct
    side_h = None
    if ctx.direct and d["async"] and not torch.cuda.is_current_stream_capturing():
        if d["side_h"] is None:
            create_side_stream(dev)
        side_h = d["side_h"]
    main_h = ops._stream()
    ws_main = ops.workspace(dev)
    ws_side = ops.workspace_of(dev, side_h) if side_h is not None else ws_main
    lens = ops.i32(ctx.lengths, dev) if ctx.lengths is not None else None
    a = _lib.DiffNetBwdArgs()
    a.gS, a.yin_all, a.a_all, a.g_all = gS.data_ptr(), yin_all.data_ptr(), a_all.data_ptr(), g_all.data_ptr()
    a.lengths = lens.data_ptr() if lens is not None else None
    wst = [rt_stream(w[0], dcond_all[:, :, :2 * C], C, 3, 2 ** (l % ctx.cycle), None, transposed=True) for l, w in enumerate(ws)]
    rt = all(t is not None for t in wst)  # the dilated conv's data gradient on the row-tile kernel (then mode 1 is never read)
    tabs = [_ptr_table(wst if rt else [packed(w[0], dt, mode=1) for w in ws]), _ptr_table([packed(w[4], dt, mode=1) for w in ws])] + \
           [_ptr_table(t) for t in tg]
    a.dil_wpt, a.out_wpt, a.dw_dil, a.db_dil, a.dw_out, a.db_out = [ctypes.cast(t, ctypes.c_void_p) for t in tabs]
    if rt:
        a.dil_wst = a.dil_wpt
    if ops.conv1d_rt_gate_bwd_ok(do_all[0], C) and all(isinstance(w[4], torch.nn.Parameter) for w in ws):
        # the output projections as operand streams too: the fused gate backward on the row-tile engine
        owst = _ptr_table([packed(w[4], dt, mode=4) for w in ws])
        tabs.append(owst)  # (kept alive with the other tables until the call returns)
        a.out_wst = ctypes.cast(owst, ctypes.c_void_p)
    a.gx_all, a.do_all, a.dcond_all, a.S = gx_all.data_ptr(), do_all.data_ptr(), dcond_all.data_ptr(), S.data_ptr()
    colpart = None
    if rt and C == 256 and ops.conv1d_rt_colpart_ok(2 * C, 3, [2 ** (l % ctx.cycle) for l in range(L)], B, T):
        # the column sums of every layer's input gradient (S) from the data-gradient convs' epilogues: no 307 MB pass over gx_all
        colpart = torch.empty((L, B, (T + 31) // 32, C), device=dev, dtype=torch.float32)
        a.colpart = colpart.data_ptr()
    a.dg_buf = dg_buf.data_ptr() if dg_buf is not None else None
    a.ws_main, a.ws_main_bytes = ws_main.data_ptr(), ws_main.numel()
    a.ws_side, a.ws_side_bytes = ws_side.data_ptr(), ws_side.numel()
    a.side_stream = side_h
    a.B, a.T, a.C, a.L, a.cycle, a.dtype = B, T, C, L, ctx.cycle, ops.dtype_code(dt)
    a.batched_wgrad = int(BATCHED_WGRAD)
    _lib.check(_lib.load().ptpp_diffnet_stack_bwd(ctypes.byref(a), main_h), "ptpp_diffnet_stack_bwd")
    if side_h is not None:  # the side stream still reads these: held until the streams are joined (sync_wgrad_stream)
        d["keep"].extend((yin_all, a_all, g_all, do_all, dcond_all))
    ctx.saved = None
    grads = [None] * (6 * L)
    if ctx.direct:
        for l in range(L):
            for i in (0, 1, 4, 5):
                _done(ws[l][i])
    else:
        for l in range(L):
            grads[6 * l + 0], grads[6 * l + 1], grads[6 * l + 4], grads[6 * l + 5] = tg[0][l], tg[1][l], tg[2][l], tg[3][l]
    return S, dcond_all, grads


def diffnet_stack(h0, cond, dsteps, lengths, cycle, layer_params):
    """layer_params: list of (dil_w, dil_b, cond_w, cond_b, out_w, out_b)."""
    flat = [t for lp in layer_params for t in lp]
    return DiffNetStackFn.apply(h0, cond, dsteps, lengths, cycle, *flat)
