"""Global-style-token style encoder (reference: promptttspp/modules/style_encoder.py:21-171)."""
import math

import torch
import torch.nn as nn

from .. import functional as PF
import torch.nn.functional as F

from .reference_encoder import ReferenceEncoder


class MultiHeadedAttention(nn.Module):
    """One query (the reference embedding) over the style tokens.  NB the
    reference scales scores by 1/sqrt(d_k * h), not 1/sqrt(d_k) (style_encoder.py:164)."""

    def __init__(self, q_dim, k_dim, v_dim, n_head, n_feat, dropout_rate=0.0):
        super().__init__()
        assert n_feat % n_head == 0
        self.d_k, self.h = n_feat // n_head, n_head
        self.linear_q = nn.Linear(q_dim, n_feat)
        self.linear_k = nn.Linear(k_dim, n_feat)
        self.linear_v = nn.Linear(v_dim, n_feat)
        self.linear_out = nn.Linear(n_feat, n_feat)
        self.dropout = nn.Dropout(p=dropout_rate)

    def forward(self, ref_emb, gst_emb):
        """ref_emb (B,1,q_dim), gst_emb (N,k_dim) -> (B,1,n_feat).  10 tokens x 1
        query per utterance: ~0.2 MFLOP, torch ops."""
        B = ref_emb.shape[0]
        lin = lambda m, x: PF.linear(x, m.weight, m.bias)  # noqa: E731  (HIP GEMM, exact f32 for f32 inputs)
        q = lin(self.linear_q, ref_emb.float()).view(B, self.h, 1, self.d_k)
        k = lin(self.linear_k, gst_emb.float().unsqueeze(0))[0].view(-1, self.h, self.d_k).transpose(0, 1)  # (h, N, dk)
        v = lin(self.linear_v, gst_emb.float().unsqueeze(0))[0].view(-1, self.h, self.d_k).transpose(0, 1)
        # 1 query x N tokens per head: broadcast products instead of batched library GEMMs
        score = F.softmax((q * k.unsqueeze(0)).sum(-1) / math.sqrt(self.d_k * self.h), dim=-1)  # (B, h, N)
        o = (self.dropout(score).unsqueeze(-1) * v.unsqueeze(0)).sum(2).reshape(B, 1, -1)
        return lin(self.linear_out, o)


class StyleTokenLayer(nn.Module):
    def __init__(self, ref_embed_dim=128, gst_tokens=10, gst_token_dim=256, gst_heads=4, dropout_rate=0.0):
        super().__init__()
        self.gst_embs = nn.Parameter(torch.randn(gst_tokens, gst_token_dim // gst_heads))
        self.mha = MultiHeadedAttention(ref_embed_dim, gst_token_dim // gst_heads, gst_token_dim // gst_heads,
                                        gst_heads, gst_token_dim, dropout_rate)

    def forward(self, ref_embs):
        """(B, ref_embed_dim, 1) -> (B, gst_token_dim)"""
        return self.mha(ref_embs.transpose(-1, -2), torch.tanh(self.gst_embs)).squeeze(1)


class StyleEncoder(nn.Module):
    def __init__(self, idim=80, gst_tokens=10, gst_token_dim=256, gst_heads=4, conv_layers=6,
                 conv_chans_list=(32, 32, 64, 64, 128, 128), conv_kernel_size=3, conv_stride=2, gru_layers=1,
                 gru_units=128):
        super().__init__()
        self.ref_enc = ReferenceEncoder(idim=idim, conv_layers=conv_layers, conv_chans_list=conv_chans_list,
                                        conv_kernel_size=conv_kernel_size, conv_stride=conv_stride,
                                        gru_layers=gru_layers, gru_units=gru_units)
        self.stl = StyleTokenLayer(ref_embed_dim=gru_units, gst_tokens=gst_tokens, gst_token_dim=gst_token_dim,
                                   gst_heads=gst_heads)

    def forward(self, speech, in_lens=None):
        """speech (B, 80, T) -> style embedding (B, token_dim, 1)."""
        return self.stl(self.ref_enc(speech, in_lens)).unsqueeze(-1)
