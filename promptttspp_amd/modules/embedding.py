"""Absolute sinusoidal positional encoding used by the frame-prior network
(reference: promptttspp/modules/embedding.py:35-92): y = dropout(x*sqrt(d) + pe[t])."""
import math

import torch
import torch.nn as nn

from .. import functional as PF
from .esp import position_rows


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout_rate, max_len=5000, reverse=False):
        super().__init__()
        self.d_model, self.reverse, self.dropout_rate = d_model, reverse, dropout_rate
        self.xscale = math.sqrt(d_model)
        self._cache = {}

    def table(self, T, device):
        if self.reverse:
            return position_rows(T - 1, T, True, self.d_model, device)
        return position_rows(0, T, False, self.d_model, device)

    def forward_cl(self, x):
        """x: (B, T, C) channels-last."""
        p = self.dropout_rate if self.training else 0.0
        return PF.posenc(x, self.table(x.shape[1], x.device), self.xscale, p)
