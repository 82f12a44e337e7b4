"""Channel-dim LayerNorm of (B, C, T) tensors (reference: promptttspp/layers/norm.py:19-32):
parameters ``gamma``/``beta`` shaped (1, C, 1).  Runs the fused HIP LayerNorm on
channels-last rows."""
import torch
import torch.nn as nn

from .. import functional as PF
from .. import ops
from ..config import compute_dtype


class LayerNorm(nn.Module):
    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.gamma = nn.Parameter(torch.ones(1, channels, 1))
        self.beta = nn.Parameter(torch.zeros(1, channels, 1))

    def forward_cl(self, x, **kw):
        return PF.layer_norm(x, self.gamma, self.beta, self.eps, **kw)

    def forward(self, x):
        y = self.forward_cl(ops.bct_to_btc(x, compute_dtype()))
        return ops.btc_to_bct(y)
