"""Frame-prior network (reference: promptttspp/modules/frame_prior.py:22-92):
x*mask -> x*sqrt(C) + sinusoid -> LN -> 6 x [ x = LN(x + dropout(gelu(conv_k17(x*mask)))) ] -> *mask.

Per layer: one MFMA implicit-GEMM conv launch (input mask fused, K = 17*256)
and one fused LayerNorm launch that applies GELU + dropout to the conv output
and adds the residual before normalising -- the reference runs ~12 kernels and
4 (B,C,T) temporaries per layer."""
import torch
import torch.nn as nn

from .. import functional as PF
from .. import ops
from ..config import compute_dtype
from .embedding import PositionalEncoding


class LayerNorm(nn.Module):
    """Parameter holder: gamma/beta shaped (C,), eps 1e-5 (frame_prior.py:22-34)."""

    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.channels, self.eps = channels, eps
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))

    def forward_cl(self, x, **kw):
        return PF.layer_norm(x, self.gamma, self.beta, self.eps, **kw)

    def forward(self, x):
        return ops.btc_to_bct(self.forward_cl(ops.bct_to_btc(x, compute_dtype())))


class FramePriorNetwork(nn.Module):
    def __init__(self, out_channels, hidden_channels, n_layers, kernel_size, p_dropout, pos_enc_p_dropout=0.1,
                 use_pos_enc=True, use_rel=False):
        super().__init__()
        if use_rel:
            raise NotImplementedError("use_rel=True is not used by any reference config")
        self.out_channels, self.hidden_channels = out_channels, hidden_channels
        self.n_layers, self.kernel_size, self.p_dropout = n_layers, kernel_size, p_dropout
        self.use_pos_enc = use_pos_enc
        if use_pos_enc:
            self.embed = PositionalEncoding(hidden_channels, pos_enc_p_dropout)
            self.norm_emb = LayerNorm(hidden_channels)
        self.convs = nn.ModuleList()
        self.norms = nn.ModuleList()
        for _ in range(n_layers):
            self.convs.append(nn.Conv1d(hidden_channels, hidden_channels, kernel_size, padding=kernel_size // 2))
            self.norms.append(LayerNorm(hidden_channels))

    def forward_cl(self, x, lengths):
        """x: (B, T, C) channels-last with rows t >= lengths[b] already zero
        (the length regulator writes zeros there) -> (B, T, C), masked."""
        p = self.p_dropout if self.training else 0.0
        if self.use_pos_enc:
            x = self.norm_emb.forward_cl(self.embed.forward_cl(x))
        last = self.n_layers - 1
        if self.n_layers > 0 and PF.conv_ln_stack_ok(x, self.convs) and len({n.eps for n in self.norms}) == 1:
            # all layers as one autograd node issued by two C calls (functional.ConvLnStackFn)
            return PF.conv_ln_stack(x, list(self.convs), list(self.norms), self.kernel_size, self.norms[0].eps, lengths, conv_mask=True,
                                    ln_res=True, act_in="gelu", drop_in=p, out_mask=2)
        for i, (conv, norm) in enumerate(zip(self.convs, self.norms)):
            z = PF.conv1d(x, conv.weight, conv.bias, ks=self.kernel_size, pad=self.kernel_size // 2, lengths=lengths,
                          in_mask=True)
            x = norm.forward_cl(z, res=x, act_in="gelu", drop_in=p, lengths=lengths, out_mask=(i == last))
        return x

    def forward(self, x, x_mask):
        """Reference signature: x (B,C,T), x_mask (B,1,T) float."""
        lengths = x_mask.sum(dim=(1, 2)).to(torch.int32)
        y = self.forward_cl(ops.bct_to_btc(x * x_mask, compute_dtype()), lengths)
        return ops.btc_to_bct(y)
