"""Absolute sinusoidal positional encoding used by the frame-prior network
(reference: promptttspp/modules/embedding.py:35-92): y = dropout(x*sqrt(d) + pe[t])."""
import math

import torch
import torch.nn as nn

from .. import functional as PF
from .esp import sinusoid_table


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout_rate, max_len=5000, reverse=False):
        super().__init__()
        self.d_model, self.reverse, self.dropout_rate = d_model, reverse, dropout_rate
        self.xscale = math.sqrt(d_model)
        self._cache = {}

    def table(self, T, device):
        key = (T, str(device))
        t = self._cache.get(key)
        if t is None:
            pos = torch.arange(T - 1, -1, -1) if self.reverse else torch.arange(T)
            t = sinusoid_table(pos, self.d_model).to(device).contiguous()
            if len(self._cache) > 64:
                self._cache.clear()
            self._cache[key] = t
        return t

    def forward_cl(self, x):
        """x: (B, T, C) channels-last."""
        p = self.dropout_rate if self.training else 0.0
        return PF.posenc(x, self.table(x.shape[1], x.device), self.xscale, p)
