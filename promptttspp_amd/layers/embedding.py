"""Phoneme embedding (reference: promptttspp/layers/embedding.py:21-36)."""
import math

import torch.nn as nn


class PhonemeEmbedding(nn.Module):
    def __init__(self, num_vocab, channels, do_scale=True, init_normal=True):
        super().__init__()
        self.emb = nn.Embedding(num_vocab, channels, padding_idx=0)
        if init_normal:
            nn.init.normal_(self.emb.weight, 0.0, channels**-0.5)
        self.scale = math.sqrt(channels)
        self.do_scale = do_scale

    def forward_cl(self, ids, mask_bt1, dtype):
        """ids (B,T) int64, mask (B,T,1) -> (B,T,C) channels-last in `dtype`.
        A 90-row table lookup: left to torch's gather (not on the roofline)."""
        x = self.emb(ids)
        if self.do_scale:
            x = x * self.scale
        return (x * mask_bt1.to(x.dtype)).to(dtype)

    def forward(self, x, mask):
        """Reference signature: ids (B,T), mask (B,1,T) -> (B,C,T)."""
        y = self.emb(x)
        if self.do_scale:
            y = y * self.scale
        return y.transpose(-1, -2) * mask
