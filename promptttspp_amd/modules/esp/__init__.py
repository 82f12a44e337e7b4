"""Conformer phoneme encoder on HIP kernels.

Drop-in for ``promptttspp.modules.esp.ConformerEncoder`` (reference:
modules/esp/__init__.py:11-65, conformer/encoder.py:60-282,
conformer/encoder_layer.py:15-162, conformer/convolution.py:13-85,
transformer/attention.py, transformer/embedding.py:220-331,
transformer/multi_layer_conv.py:12-67): same constructor kwargs and state-dict
keys (Appendix A of SURVEY.md), re-designed for MI355X:

* activations stay channels-last (B, T, C) -- no (B,C,T) round trips around the
  k=9 feed-forward convs, which run as MFMA implicit GEMMs with ReLU / mask /
  dropout / 0.5-scaled residual fused into the epilogue;
* masks are per-utterance lengths (int32), never (B,T,T) tensors;
* q/k/v are ONE fused (C -> 3C) projection; the rel-shift is an index
  computation inside the attention kernel (no pad/view tensors);
* LayerNorm (eps 1e-12) is the wavefront-shuffle kernel.

Supported configuration = the one the reference's YAMLs use
(``normalize_before``, macaron FFN pair, conv1d position-wise layers, CNN
module, relative positions "new" (training) and "legacy" (demo)).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import functional as PF
from ... import nn_ops as NO
from ...config import compute_dtype

LN_EPS = 1e-12  # ESPnet LayerNorm (transformer/layer_norm.py:21)


def sinusoid_table(positions, d):
    """(n,) real positions -> (n, d) interleaved sin/cos encodings (f32)."""
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    ang = positions.to(torch.float32)[:, None] * div[None, :]
    pe = torch.zeros(positions.shape[0], d)
    pe[:, 0::2] = torch.sin(ang)
    pe[:, 1::2] = torch.cos(ang)
    return pe


# Master tables: the encodings of positions N-1 ... -(N-1) (descending) and 0 ... N-1 are computed ONCE
# per (d, device, dtype) -- on the CPU, exactly like the per-length tables before -- and every length
# takes a contiguous row slice.  (Measured: building a table per new sequence length cost 20-50 ms of
# host time per table; with token-bucket batching almost every step has new lengths.)
_MASTER_N = 8192
_master_cache = {}


def position_rows(first, count, descending, d, device, dtype=torch.float32):
    """Rows encoding positions first, first-1, ... (descending) or first, first+1, ... (ascending)."""
    last = first - (count - 1) if descending else first + (count - 1)
    if max(abs(first), abs(last)) >= _MASTER_N or (not descending and min(first, last) < 0):
        step = -1 if descending else 1
        return sinusoid_table(torch.arange(first, first + step * count, step), d).to(device=device, dtype=dtype)
    key = (d, str(device), dtype, descending)
    m = _master_cache.get(key)
    if m is None:
        pos = torch.arange(_MASTER_N - 1, -_MASTER_N, -1) if descending else torch.arange(_MASTER_N)
        m = _master_cache[key] = sinusoid_table(pos, d).to(device=device, dtype=dtype).contiguous()
    i0 = (_MASTER_N - 1) - first if descending else first
    return m[i0 : i0 + count]


class _LN(nn.LayerNorm):
    """Parameter holder with ESPnet's key names (weight/bias) and eps."""

    def __init__(self, n):
        super().__init__(n, eps=LN_EPS)

    def cl(self, x, **kw):
        return PF.layer_norm(x, self.weight, self.bias, LN_EPS, **kw)


class MultiLayeredConv1d(nn.Module):
    def __init__(self, in_chans, hidden_chans, kernel_size, dropout_rate):
        super().__init__()
        self.ks = kernel_size
        self.w_1 = nn.Conv1d(in_chans, hidden_chans, kernel_size, padding=(kernel_size - 1) // 2)
        self.w_2 = nn.Conv1d(hidden_chans, in_chans, kernel_size, padding=(kernel_size - 1) // 2)
        self.dropout_rate = dropout_rate

    def cl(self, x, lengths, res, scale, outer_drop):
        """res + scale * drop_outer(mask * w_2(drop(mask * relu(w_1(mask * x)))))"""
        p = self.dropout_rate if self.training else 0.0
        pad = (self.ks - 1) // 2
        h = PF.conv1d(x, self.w_1.weight, self.w_1.bias, ks=self.ks, pad=pad, act="relu", lengths=lengths,
                      in_mask=True, out_mask=True, drop_p=p)
        return PF.conv1d(h, self.w_2.weight, self.w_2.bias, res=res, ks=self.ks, pad=pad, lengths=lengths,
                         out_mask=True, out_scale=scale, drop_p=outer_drop)


class RelPositionMultiHeadedAttention(nn.Module):
    def __init__(self, n_head, n_feat, dropout_rate, variant):
        super().__init__()
        assert n_feat % n_head == 0
        self.h, self.d_k, self.variant = n_head, n_feat // n_head, variant
        self.pos_bias_u = nn.Parameter(torch.empty(n_head, self.d_k))
        self.pos_bias_v = nn.Parameter(torch.empty(n_head, self.d_k))
        nn.init.xavier_uniform_(self.pos_bias_u)
        nn.init.xavier_uniform_(self.pos_bias_v)
        self.linear_q = nn.Linear(n_feat, n_feat)
        self.linear_k = nn.Linear(n_feat, n_feat)
        self.linear_v = nn.Linear(n_feat, n_feat)
        self.linear_out = nn.Linear(n_feat, n_feat)
        self.linear_pos = nn.Linear(n_feat, n_feat, bias=False)
        assert dropout_rate == 0.0, "attention dropout is 0 in every reference config"

    def cl(self, x, pos_emb, lengths, res, out_drop):
        qkv = PF.linear_fused(x, (self.linear_q, self.linear_k, self.linear_v))
        pp = PF.linear(pos_emb.unsqueeze(0), self.linear_pos.weight)[0]
        ctx = PF.attention(qkv, pp, self.pos_bias_u, self.pos_bias_v, lengths, self.h, self.variant)
        # x = residual + dropout(att * mask)
        return PF.conv1d(ctx, self.linear_out.weight, self.linear_out.bias, res=res, lengths=lengths, out_mask=True,
                         drop_p=out_drop)


class ConvolutionModule(nn.Module):
    def __init__(self, channels, kernel_size, bias=True):
        super().__init__()
        assert (kernel_size - 1) % 2 == 0
        self.pointwise_conv1 = nn.Conv1d(channels, 2 * channels, 1, bias=bias)
        self.depthwise_conv = nn.Conv1d(channels, channels, kernel_size, padding=(kernel_size - 1) // 2,
                                        groups=channels, bias=bias)
        self.norm = nn.BatchNorm1d(channels)
        self.pointwise_conv2 = nn.Conv1d(channels, channels, 1, bias=bias)

    def cl(self, x, lengths, mask_bt1, res, out_drop):
        h = PF.conv1d(x, self.pointwise_conv1.weight, self.pointwise_conv1.bias, lengths=lengths, out_mask=True)
        # GLU -> depthwise k=7 (masked) -> BatchNorm1d (train mode: batch statistics over
        # ALL rows, padded ones included, like the reference) -> Swish: HIP kernels (bn_dw.hip)
        h = NO.dwconv1d(NO.glu(h), self.depthwise_conv.weight, self.depthwise_conv.bias, lengths)
        h = NO.batch_norm_act(h, self.norm, act="swish")
        # x = residual + dropout(mask * pw2(h)) * mask
        return PF.conv1d(h, self.pointwise_conv2.weight, self.pointwise_conv2.bias, res=res, lengths=lengths,
                         out_mask=True, drop_p=out_drop)


class EncoderLayer(nn.Module):
    def __init__(self, size, self_attn, feed_forward, feed_forward_macaron, conv_module, dropout_rate):
        super().__init__()
        self.self_attn = self_attn
        self.feed_forward = feed_forward
        self.feed_forward_macaron = feed_forward_macaron
        self.conv_module = conv_module
        self.norm_ff = _LN(size)
        self.norm_mha = _LN(size)
        self.norm_ff_macaron = _LN(size)
        self.norm_conv = _LN(size)
        self.norm_final = _LN(size)
        self.dropout_rate = dropout_rate

    def cl(self, x, pos_emb, lengths, mask_bt1, counted=False):
        """``counted``: the caller has already advanced the BatchNorm step counter (one multi-tensor launch for all blocks)."""
        p = self.dropout_rate if self.training else 0.0
        if _block_driver_ok(self, x):
            from types import SimpleNamespace

            bn = self.conv_module.norm
            if bn.training and bn.num_batches_tracked is not None and not counted:
                bn.num_batches_tracked.add_(1)
            cfg = SimpleNamespace(lengths=lengths, heads=self.self_attn.h, variant=self.self_attn.variant, p=float(p),
                                  p_ffn=float(self.feed_forward.dropout_rate if self.training else 0.0), bn=bn, training=bn.training)
            return ConformerBlockFn.apply(x, pos_emb, cfg, *_block_params(self))
        x = self.feed_forward_macaron.cl(self.norm_ff_macaron.cl(x), lengths, x, 0.5, p)
        x = self.self_attn.cl(self.norm_mha.cl(x), pos_emb, lengths, x, p)
        x = self.conv_module.cl(self.norm_conv.cl(x), lengths, mask_bt1, x, p)
        x = self.feed_forward.cl(self.norm_ff.cl(x), lengths, x, 0.5, p)
        return self.norm_final.cl(x, lengths=lengths, out_mask=True)


# ----------------------------------------------------------------------------------------------------------------------
# One EncoderLayer as ONE autograd node issued by two C calls (ptpp_conformer_block_fwd / _bwd, include/ptpp.h): the ~25
# forward and ~45 backward launches of ``EncoderLayer.cl`` in the same order with the same arguments and dropout seeds
# (bit-identical, tests/test_stack_drivers.py), without their Python -> C round trips, per-launch allocations and ~20
# autograd nodes per block.
# ----------------------------------------------------------------------------------------------------------------------
def _block_params(layer):
    """The block's parameters in the order of ``ConformerBlockFn``'s flat argument list.  The (module, name) pairs are resolved
    once per layer object; every call re-checks that each module on the paths is still the same child of its parent (20 dict reads)
    and then reads the CURRENT Parameter objects from the modules' ``_parameters`` tables -- 57 dict reads instead of ~75
    ``nn.Module.__getattr__`` calls per call, four calls per direction and step."""
    spec = layer.__dict__.get("_ptpp_param_spec")
    if spec is not None:
        for d, n, o in spec[0]:
            if d[n] is not o:
                spec = None
                break
    if spec is None:
        a, cm, ffm, ff = layer.self_attn, layer.conv_module, layer.feed_forward_macaron, layer.feed_forward
        norms = (layer.norm_ff_macaron, layer.norm_mha, layer.norm_conv, layer.norm_ff, layer.norm_final)
        pairs = [(n, "weight") for n in norms] + [(n, "bias") for n in norms] + \
                [(ffm.w_1, "weight"), (ffm.w_1, "bias"), (ffm.w_2, "weight"), (ffm.w_2, "bias"),
                 (ff.w_1, "weight"), (ff.w_1, "bias"), (ff.w_2, "weight"), (ff.w_2, "bias"),
                 (a.linear_q, "weight"), (a.linear_q, "bias"), (a.linear_k, "weight"), (a.linear_k, "bias"),
                 (a.linear_v, "weight"), (a.linear_v, "bias"), (a.linear_pos, "weight"), (a.linear_out, "weight"),
                 (a.linear_out, "bias"), (a, "pos_bias_u"), (a, "pos_bias_v"),
                 (cm.pointwise_conv1, "weight"), (cm.pointwise_conv1, "bias"), (cm.pointwise_conv2, "weight"),
                 (cm.pointwise_conv2, "bias"), (cm.depthwise_conv, "weight"), (cm.depthwise_conv, "bias"),
                 (cm.norm, "weight"), (cm.norm, "bias")]
        checks = [(layer._modules, n, layer._modules[n]) for n in ("self_attn", "conv_module", "feed_forward_macaron", "feed_forward",
                                                                  "norm_ff_macaron", "norm_mha", "norm_conv", "norm_ff", "norm_final")]
        for parent, names in ((ffm, ("w_1", "w_2")), (ff, ("w_1", "w_2")),
                              (a, ("linear_q", "linear_k", "linear_v", "linear_pos", "linear_out")),
                              (cm, ("pointwise_conv1", "pointwise_conv2", "depthwise_conv", "norm"))):
            checks += [(parent._modules, n, parent._modules[n]) for n in names]
        spec = (checks, [(m._parameters, n) for m, n in pairs])
        layer.__dict__["_ptpp_param_spec"] = spec
    return [d[n] for d, n in spec[1]]


_N_LN, _I_FF, _I_ATT, _I_CM = 10, 10, 18, 29   # offsets into the flat list: norms, feed-forward pair, attention, conv module
_VARIANT = {"new": 0, "legacy": 1}


def _fill_weights(w, P, dt, bn, training, ffn_streams=False):
    """ptpp_conformer_weights from the flat parameter list ``P`` (packed operands / f32 parameters).  ``ffn_streams``: the four
    feed-forward convs run from operand streams (pack mode 3 / 4, ``ffn_ws``): their mode-0 operands are then never read and are
    NOT requested -- every cached operand is re-packed after every optimiser step, and these four are 23 % of the model's packed
    elements each way (tools/bench_pack.py).  (Called once per block and direction: one positional struct construction instead of
    ~40 attribute stores.)"""
    from ... import _lib

    f32, pk = PF._f32_param, PF.packed
    C = P[33].shape[0]
    none4 = (None, None, None, None)
    ffn = none4 if ffn_streams else (pk(P[10], dt), pk(P[12], dt), pk(P[14], dt), pk(P[16], dt))
    if training:
        has = bn.running_mean is not None and bn.running_var is not None
        stats = (bn.running_mean if has else None, bn.running_var if has else None, None, None)
    else:
        stats = (None, None, bn.running_mean.float().contiguous(), torch.rsqrt(bn.running_var.float() + bn.eps).contiguous())
    ts = [f32(P[0]), f32(P[1]), f32(P[2]), f32(P[3]), f32(P[4]), f32(P[5]), f32(P[6]), f32(P[7]), f32(P[8]), f32(P[9]),
          ffn[0], f32(P[11]), ffn[1], f32(P[13]), ffn[2], f32(P[15]), ffn[3], f32(P[17]),
          PF.packed_cat((P[18], P[20], P[22]), dt), PF.bias_cat((P[19], P[21], P[23])), pk(P[24], dt), pk(P[25], dt), f32(P[26]),
          f32(P[27]), f32(P[28]), pk(P[29], dt), f32(P[30]), pk(P[31], dt), f32(P[32]), f32(P[33]).reshape(C, -1), f32(P[34]),
          f32(P[35]), f32(P[36]), *stats]
    w.__init__(*[None if t is None else t.data_ptr() for t in ts])
    return [t for t in ts if t is not None]


class ConformerBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos, cfg, *P):
        import ctypes

        from ... import _lib, ops

        x = x.contiguous()
        pos = pos.contiguous()
        B, T, C = x.shape
        dt, dev = x.dtype, x.device
        F_, H, L = P[10].shape[0], cfg.heads, pos.shape[0]
        dcode = ops.dtype_code(dt)
        lib = _lib.load()
        need_bwd = any(ctx.needs_input_grad)
        p_ffn, p = cfg.p_ffn, cfg.p
        seeds = [PF.next_seed() if q > 0 else 0 for q in (p_ffn, p, p, p, p_ffn, p)]  # the draws of the six Conv1dFn.forward calls
        slab = torch.empty(lib.ptpp_conformer_block_slab_bytes(B, T, C, F_, H, L, dcode), device=dev, dtype=torch.uint8)
        y = torch.empty_like(x)
        a = _lib.ConformerFwdArgs()
        # the feed-forward convs on the row-tile engine where they qualify (operand streams, pack mode 3): ops.conv1d's rule
        hF = x.new_empty((1, 1, F_))
        streams = ops.conv1d_rt_ex_ok(x, F_, P[10].shape[2], 1, "relu") and ops.conv1d_rt_ex_ok(hF, C, P[10].shape[2], 1, None) and \
            all(isinstance(P[i], torch.nn.Parameter) for i in (10, 12, 14, 16))
        keep = _fill_weights(a.w, P, dt, cfg.bn, cfg.training, ffn_streams=streams)
        lens = ops.i32(cfg.lengths, dev)
        a.x, a.y, a.pos_emb, a.lengths = x.data_ptr(), y.data_ptr(), pos.data_ptr(), lens.data_ptr()
        a.slab, a.slab_bytes = slab.data_ptr(), slab.numel()
        if not torch.cuda.is_current_stream_capturing():
            ws = ops.workspace(dev)
            a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
        a.red_scratch, a.red_bytes = ops.reduction_scratch(dev)
        sd = (ctypes.c_uint64 * 6)(*seeds)
        a.seeds = ctypes.cast(sd, ctypes.c_void_p)
        a.p_ffn, a.p_drop = p_ffn, p
        a.bn_momentum, a.bn_eps = float(cfg.bn.momentum if cfg.bn.momentum is not None else 0.1), float(cfg.bn.eps)
        a.B, a.T, a.C, a.F, a.H, a.L = B, T, C, F_, H, L
        a.ks_ffn, a.ks_dw, a.variant = P[10].shape[2], P[33].shape[-1], _VARIANT[cfg.variant]
        a.bn_train, a.save, a.dtype = int(cfg.training), int(need_bwd), dcode
        if streams:
            for k, i in enumerate((10, 12, 14, 16)):
                t = PF.packed(P[i], dt, mode=3)
                keep.append(t)
                a.ffn_ws[k] = t.data_ptr()
        _lib.check(lib.ptpp_conformer_block_fwd(ctypes.byref(a), ops._stream()), "ptpp_conformer_block_fwd")
        if need_bwd:
            ctx.cfg, ctx.P, ctx.seeds, ctx.dims = cfg, P, seeds, (B, T, C, F_, H, L)
            ctx.tensors = (x, pos, lens, slab)
            ctx.w, ctx.w_keep = a.w, keep  # the forward operands: the backward reads the same ones (no second lookup)
            ctx.direct = [PF._sink(t) is not None for t in P]
            # everything accumulates in place when the trainer allows it (the depthwise kernel adds with atomics: straight into
            # p.grad instead of into a zero-filled buffer that autograd then adds); the BatchNorm parameter gradients' kernel
            # OVERWRITES a (2C) buffer, which the backward then adds to both p.grad with one multi-tensor launch
            ctx.bn_direct = ctx.direct[35] and ctx.direct[36]
            for i in (35, 36):
                ctx.direct[i] = False
            for i, t in enumerate(P):
                if ctx.direct[i] or (ctx.bn_direct and i in (35, 36)):
                    PF._use(t)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        import ctypes

        from ... import _lib, ops

        cfg, P = ctx.cfg, ctx.P
        B, T, C, F_, H, L = ctx.dims
        x, pos, lens, slab = ctx.tensors
        dt, dev = x.dtype, x.device
        dcode = ops.dtype_code(dt)
        lib = _lib.load()
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        scratch = torch.empty(lib.ptpp_conformer_block_bwd_scratch_bytes(B, T, C, F_, H, L, dcode), device=dev, dtype=torch.uint8)
        tg = [t.grad if d else (None if i in (35, 36) else torch.zeros(t.shape, device=dev, dtype=torch.float32))
              for i, (t, d) in enumerate(zip(P, ctx.direct))]
        bn_sums = torch.empty(2 * C, device=dev, dtype=torch.float32)
        a = _lib.ConformerBwdArgs()
        a.w = ctx.w
        a.gy, a.gx, a.x, a.pos_emb, a.lengths = gy.data_ptr(), gx.data_ptr(), x.data_ptr(), pos.data_ptr(), lens.data_ptr()
        hF = gy.new_empty((1, 1, F_))
        streams = ops.conv1d_rt_ex_ok(gy, F_, P[10].shape[2], 1, None) and ops.conv1d_rt_ex_ok(hF, C, P[10].shape[2], 1, None) and \
            all(isinstance(P[i], torch.nn.Parameter) for i in (10, 12, 14, 16))
        tr = [PF.packed_cat((P[18], P[20], P[22]), dt, mode=1)] + [PF.packed(P[i], dt, mode=1) for i in (25, 29, 31)]
        (a.qkv_wt, a.out_wt, a.pw1_wt, a.pw2_wt) = [t.data_ptr() for t in tr]
        if streams:  # the feed-forward data gradients' operand streams (pack mode 4); their mode-1 operands are not requested
            for k, i in enumerate((10, 12, 14, 16)):
                t = PF.packed(P[i], dt, mode=4)
                tr.append(t)
                a.ffn_wts[k] = t.data_ptr()
        else:
            tf = [PF.packed(P[i], dt, mode=1) for i in (10, 12, 14, 16)]
            (a.ffm_w1t, a.ffm_w2t, a.ff_w1t, a.ff_w2t) = [t.data_ptr() for t in tf]
            tr += tf
        # (the accumulation targets in the field order of ptpp_conformer_grads = the order of the flat parameter list)
        a.g.__init__(*[t.data_ptr() for t in tg[:35]], bn_sums.data_ptr())
        a.slab, a.scratch, a.scratch_bytes = slab.data_ptr(), scratch.data_ptr(), scratch.numel()
        d = PF._direct
        side_h = None
        if any(ctx.direct) and d["async"] and not torch.cuda.is_current_stream_capturing():
            if d["side_h"] is None:
                PF.create_side_stream(dev)
            side_h = d["side_h"]
        main_h = ops._stream()
        ws_main = ops.workspace(dev)
        ws_side = ops.workspace_of(dev, side_h) if side_h is not None else ws_main
        a.ws_main, a.ws_main_bytes = ws_main.data_ptr(), ws_main.numel()
        a.ws_side, a.ws_side_bytes = ws_side.data_ptr(), ws_side.numel()
        a.red_scratch, a.red_bytes = ops.reduction_scratch(dev)
        a.side_stream = side_h
        a.side_stream2 = PF.second_wgrad_stream(dev) if side_h is not None else None
        if ctx.bn_direct:  # the BatchNorm parameter gradients are added by the finishing launch of their sums
            a.bn_dgamma, a.bn_dbeta = P[35].grad.data_ptr(), P[36].grad.data_ptr()
        sd = (ctypes.c_uint64 * 6)(*ctx.seeds)
        a.seeds = ctypes.cast(sd, ctypes.c_void_p)
        a.p_ffn, a.p_drop = cfg.p_ffn, cfg.p
        a.B, a.T, a.C, a.F, a.H, a.L = B, T, C, F_, H, L
        a.ks_ffn, a.ks_dw, a.variant = P[10].shape[2], P[33].shape[-1], _VARIANT[cfg.variant]
        a.bn_train, a.dtype = int(cfg.training), dcode
        with ops.red_immediate(not all(ctx.direct[i] for i in (*range(10), 27, 28))):  # LayerNorm / position-bias targets
            _lib.check(lib.ptpp_conformer_block_bwd(ctypes.byref(a), main_h), "ptpp_conformer_block_bwd")
        if side_h is not None:  # the side stream still reads these: held until the streams are joined
            d["keep"].extend((x, pos, slab, scratch))
        ctx.tensors = None
        grads = []
        if ctx.bn_direct and not a.bn_dgamma:
            torch._foreach_add_([P[35].grad, P[36].grad], [bn_sums[C:].view_as(P[35]), bn_sums[:C].view_as(P[36])])
        for i, (t, dr) in enumerate(zip(P, ctx.direct)):
            if dr or (ctx.bn_direct and i in (35, 36)):
                PF._done(t)
                grads.append(None)
            elif i == 35:
                grads.append(bn_sums[C:].view_as(t))   # dgamma
            elif i == 36:
                grads.append(bn_sums[:C].view_as(t))   # dbeta
            else:
                grads.append(tg[i].view_as(t))
        return (gx, None, None, *grads)


def _block_driver_ok(layer, x):
    cm = layer.conv_module
    return (PF.STACK_DRIVERS and x.is_cuda and x.shape[-1] % 8 == 0 and cm.depthwise_conv.bias is not None and
            cm.pointwise_conv1.bias is not None and layer.self_attn.variant in _VARIANT and
            layer.feed_forward.ks == layer.feed_forward_macaron.ks and
            layer.feed_forward.dropout_rate == layer.feed_forward_macaron.dropout_rate)


class Encoder(nn.Module):
    def __init__(self, attention_dim, attention_heads, linear_units, num_blocks, dropout_rate,
                 positional_dropout_rate, attention_dropout_rate, positionwise_conv_kernel_size, cnn_module_kernel,
                 variant):
        super().__init__()
        self.variant = variant
        self.attention_dim = attention_dim
        self.positional_dropout_rate = positional_dropout_rate
        self.encoders = nn.Sequential(*[
            EncoderLayer(
                attention_dim,
                RelPositionMultiHeadedAttention(attention_heads, attention_dim, attention_dropout_rate, variant),
                MultiLayeredConv1d(attention_dim, linear_units, positionwise_conv_kernel_size, dropout_rate),
                MultiLayeredConv1d(attention_dim, linear_units, positionwise_conv_kernel_size, dropout_rate),
                ConvolutionModule(attention_dim, cnn_module_kernel),
                dropout_rate,
            )
            for _ in range(num_blocks)
        ])
        self.after_norm = _LN(attention_dim)
        self._pos_cache = {}

    def pos_table(self, T, device, dtype):
        """rows handed to linear_pos.  new (embedding.py:263-331): relative positions
        T-1 ... -(T-1).  legacy (embedding.py:220-257): the table is built once for
        5000 positions in reverse order and sliced, so row k encodes position 4999-k."""
        if self.variant == "new":
            return position_rows(T - 1, 2 * T - 1, True, self.attention_dim, device, dtype)
        return position_rows(max(T, 5000) - 1, T, True, self.attention_dim, device, dtype)

    def cl(self, x, lengths, mask_bt1):
        p = self.positional_dropout_rate if self.training else 0.0
        x = PF.posenc(x, None, math.sqrt(self.attention_dim), p)
        pos = self.pos_table(x.shape[1], x.device, x.dtype)
        if p > 0:
            pos = PF.posenc(pos.unsqueeze(0), None, 1.0, p)[0]
        # the BatchNorm step counters of all blocks in one multi-tensor launch (they feed nothing in the step)
        drv = [l for l in self.encoders if _block_driver_ok(l, x)]
        tracked = [l.conv_module.norm.num_batches_tracked for l in drv
                   if l.conv_module.norm.training and l.conv_module.norm.num_batches_tracked is not None]
        if len(tracked) > 1 and len(drv) == len(self.encoders):
            torch._foreach_add_(tracked, 1)
            counted = True
        else:
            counted = False
        for layer in self.encoders:
            x = layer.cl(x, pos, lengths, mask_bt1, counted=counted) if counted else layer.cl(x, pos, lengths, mask_bt1)
        return self.after_norm.cl(x, lengths=lengths, out_mask=True)


class ConformerEncoder(nn.Module):
    def __init__(self, idim=8, attention_dim=8, return_mask=False, rel_pos_type=None, attention_heads=4,
                 linear_units=2048, num_blocks=6, dropout_rate=0.1, positional_dropout_rate=0.1,
                 attention_dropout_rate=0.0, normalize_before=True, concat_after=False,
                 positionwise_layer_type="linear", positionwise_conv_kernel_size=1, macaron_style=False,
                 pos_enc_layer_type="abs_pos", selfattention_layer_type="selfattn", activation_type="swish",
                 use_cnn_module=False, zero_triu=False, cnn_module_kernel=31, **unsupported):
        super().__init__()
        self._out_dim = attention_dim
        self.return_mask = return_mask
        if rel_pos_type is None or rel_pos_type == "legacy":
            variant = "legacy"
        elif rel_pos_type == "new":
            variant = "new"
        else:
            raise ValueError(f"Unknown relative positional encoding type: {rel_pos_type}")
        hot = (idim == attention_dim and normalize_before and not concat_after and macaron_style and use_cnn_module
               and positionwise_layer_type == "conv1d" and pos_enc_layer_type == "rel_pos"
               and selfattention_layer_type == "rel_selfattn" and activation_type == "swish" and not zero_triu
               and not unsupported)
        if not hot:
            raise NotImplementedError(
                "promptttspp_amd implements the Conformer configuration of prompttts_mdn_v2_wo_erg_final(_demo).yaml "
                "(idim == attention_dim, macaron conv1d FFN, rel_pos attention, CNN module, swish); "
                f"unsupported extras: {sorted(unsupported)}")
        self.encoder = Encoder(attention_dim, attention_heads, linear_units, num_blocks, dropout_rate,
                               positional_dropout_rate, attention_dropout_rate, positionwise_conv_kernel_size,
                               cnn_module_kernel, variant)

    @property
    def out_dim(self):
        return self._out_dim

    def forward_cl(self, x, lengths, mask_bt1):
        """x: (B, T, C) channels-last compute dtype, already masked."""
        return self.encoder.cl(x, lengths, mask_bt1)

    def forward(self, emb, input_lens=None):
        """Reference signature: emb (B, T, idim) float, lengths (B,) -> (B, T, C)."""
        B, T, _ = emb.shape
        if input_lens is None:
            input_lens = torch.full((B,), T, device=emb.device, dtype=torch.long)
        lengths = input_lens.to(device=emb.device, dtype=torch.int32)
        mask = (torch.arange(T, device=emb.device)[None, :] < lengths[:, None]).unsqueeze(-1)
        y = self.forward_cl(emb.to(compute_dtype()).contiguous(), lengths, mask.float()).float()
        if self.return_mask:
            return y, mask.to(y.dtype)
        return y
