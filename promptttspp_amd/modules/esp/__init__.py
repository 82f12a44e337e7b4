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

    def cl(self, x, pos_emb, lengths, mask_bt1):
        p = self.dropout_rate if self.training else 0.0
        x = self.feed_forward_macaron.cl(self.norm_ff_macaron.cl(x), lengths, x, 0.5, p)
        x = self.self_attn.cl(self.norm_mha.cl(x), pos_emb, lengths, x, p)
        x = self.conv_module.cl(self.norm_conv.cl(x), lengths, mask_bt1, x, p)
        x = self.feed_forward.cl(self.norm_ff.cl(x), lengths, x, 0.5, p)
        return self.norm_final.cl(x, lengths=lengths, out_mask=True)


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
        for layer in self.encoders:
            x = layer.cl(x, pos, lengths, mask_bt1)
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
