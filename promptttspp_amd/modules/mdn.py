"""Mixture-density heads and losses (reference: promptttspp/modules/mdn.py:37-257).

The heads are three Linear layers fused into ONE f32 MFMA GEMM launch
(``mdn_disable_amp``: the MDN island is always float32).  The loss algebra acts
on (B, T, G, D) tensors with G*D <= 2560 per row -- a few kB per utterance -- and
is expressed with torch tensor ops in f32 (no kernel of its own: it is far off
the roofline-relevant path)."""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as PF

FUSED_NLL = not os.environ.get("PTPP_NO_FUSED_MDN")  # (tests compare the fused launches with the tensor-op path)


class MDNLayer(nn.Module):
    def __init__(self, in_dim, out_dim, num_gaussians=30, dim_wise=False):
        super().__init__()
        self.in_dim, self.out_dim, self.num_gaussians, self.dim_wise = in_dim, out_dim, num_gaussians, dim_wise
        self.log_pi = nn.Linear(in_dim, out_dim * num_gaussians if dim_wise else num_gaussians)
        self.log_sigma = nn.Linear(in_dim, out_dim * num_gaussians)
        self.mu = nn.Linear(in_dim, out_dim * num_gaussians)

    def raw(self, minibatch):
        """(B, T, D_in) -> the three heads' outputs side by side, (B, T, n_pi + 2 G D) f32: [pi logits | log_sigma | mu].
        The fused training loss (functional.tts_losses) applies the log-softmax itself."""
        x = minibatch.float().contiguous()
        # the three heads read the same input: ONE GEMM over their concatenated rows (the packed operand is
        # cached on the three Parameters, their gradients go straight into each layer's own buffer)
        return PF.linear_fused(x, [self.log_pi, self.log_sigma, self.mu])

    def forward(self, minibatch):
        """(B, T, D_in) -> log_pi (B,T,G[,D]), log_sigma (B,T,G,D), mu (B,T,G,D); f32."""
        B, T, _ = minibatch.shape
        G, D = self.num_gaussians, self.out_dim
        y = self.raw(minibatch)
        n_pi = self.log_pi.weight.shape[0]
        log_pi, log_sigma, mu = y[..., :n_pi], y[..., n_pi : n_pi + G * D], y[..., n_pi + G * D :]
        if self.dim_wise:
            log_pi = F.log_softmax(log_pi.reshape(B, T, G, D), dim=2)
        else:
            log_pi = F.log_softmax(log_pi, dim=2)
        return log_pi, log_sigma.reshape(B, T, G, D), mu.reshape(B, T, G, D)


class _MdnNllFn(torch.autograd.Function):
    """The dimension-wise NLL as one launch forward and one backward (ptpp_mdn_nll_fwd / _bwd) instead of ~35 + ~70
    tensor ops; same clamps, same masking (masked positions: +inf, zero gradients)."""

    @staticmethod
    def forward(ctx, log_pi, log_sigma, mu, target, mask, lp_min, ls_min):
        from .. import ops

        log_pi, log_sigma, mu = log_pi.contiguous(), log_sigma.contiguous(), mu.contiguous()
        target = target.contiguous()
        m8 = mask.contiguous().view(torch.uint8) if mask is not None else None
        loss = ops.mdn_nll_fwd(log_pi, log_sigma, mu, target, m8, lp_min, ls_min)
        ctx.save_for_backward(log_pi, log_sigma, mu, target, loss)
        ctx.m8, ctx.mins = m8, (lp_min, ls_min)
        return loss

    @staticmethod
    def backward(ctx, gout):
        from .. import ops

        log_pi, log_sigma, mu, target, loss = ctx.saved_tensors
        dlp, dls, dmu = ops.mdn_nll_bwd(log_pi, log_sigma, mu, target, ctx.m8, loss, gout.contiguous().float(), *ctx.mins)
        return dlp, dls, dmu, None, None, None, None


def mdn_loss(log_pi, log_sigma, mu, target, log_pi_min=-7.0, log_sigma_min=-7.0, reduce=True, mask=None):
    """Negative log-likelihood of `target` (B,T,D) under the mixture: clamp log_sigma
    / log_pi from below, clamp the centred target to +-5 sigma, Gaussian log-density
    + log weight, -logsumexp over components (mdn.py:81-175)."""
    dim_wise = log_pi.dim() == 4
    if (dim_wise and mu.is_cuda and FUSED_NLL and log_pi.dtype == log_sigma.dtype == mu.dtype == torch.float32
            and target.dtype == torch.float32 and target.shape == mu.shape[:2] + mu.shape[3:]
            and (mask is None or (mask.dtype == torch.bool and mask.numel() == mu.shape[0] * mu.shape[1]))):
        loss = _MdnNllFn.apply(log_pi, log_sigma, mu, target, None if mask is None else mask.reshape(mu.shape[:2]),
                               log_pi_min, log_sigma_min)
        return loss.mean(dim=1) if reduce else loss
    log_sigma = log_sigma.clamp(min=log_sigma_min)
    log_pi = log_pi.clamp(min=log_pi_min)
    sigma = torch.exp(log_sigma)
    d = target.unsqueeze(2) - mu
    d = torch.maximum(torch.minimum(d, 5 * sigma), -5 * sigma)
    log_prob = -0.5 * (d / sigma) ** 2 - log_sigma - 0.5 * math.log(2 * math.pi)
    if dim_wise:
        ll = log_prob + log_pi
    else:
        ll = log_prob.sum(dim=3) + log_pi
    if mask is not None:
        m = ~mask.unsqueeze(-1) if ll.dim() == 4 else ~mask
        ll = ll.masked_fill(m.expand_as(ll), -float("inf"))
    loss = -torch.logsumexp(ll, dim=2)
    return loss.mean(dim=1) if reduce else loss


def _select(log_pi, log_sigma, mu, idx):
    """gather the chosen component per (b,t[,d])"""
    if log_pi.dim() == 4:
        idx = idx.unsqueeze(2)  # (B,T,1,D)
    else:
        idx = idx[:, :, None, None].expand(-1, -1, 1, mu.shape[3])
    return torch.exp(log_sigma.gather(2, idx).squeeze(2)), mu.gather(2, idx).squeeze(2)


def mdn_get_most_probable_sigma_and_mu(log_pi, log_sigma, mu):
    """(sigma, mu) of the largest-weight component (mdn.py:178-223); the arg-max
    index is the integer the parity tests pin bit-exactly."""
    return _select(log_pi, log_sigma, mu, log_pi.argmax(dim=2))


def mdn_sample_sigma_and_mu(log_pi, log_sigma, mu):
    """(sigma, mu) of a component drawn from the mixture weights (mdn.py:226-257)."""
    if log_pi.dim() == 4:
        probs = log_pi.exp().permute(0, 1, 3, 2)  # (B,T,D,G)
    else:
        probs = log_pi.exp()
    idx = torch.distributions.Categorical(probs=probs).sample()
    return _select(log_pi, log_sigma, mu, idx)
