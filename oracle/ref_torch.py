"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain PyTorch fp32, functional)
of the PromptTTS++ hot path, written from SURVEY.md section 8 / Appendix C.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg
may import this file; the product (``promptttspp_amd``) never does.

Parity status: PINNED.  ``oracle/gen_golden.py`` (run in the build container,
where /root/reference is importable) executes the reference modules on seeded
inputs / synthetic weights (oracle/fill.py) and stores their outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function here
against those vectors.  The reference has no tests or golden vectors of its
own, so these captured outputs are the pin.

Every function takes a flat state dict ``sd`` (reference key names, Appendix A)
and a key prefix.  Tensors follow the REFERENCE layouts ((B, C, T) for conv
stacks, (B, T, C) for the Conformer).  Citations are reference file:line.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# masks / alignment  (promptttspp/utils/model.py:30-47)
# --------------------------------------------------------------------------


def sequence_mask(length, max_length=None):
    """(B,) lengths -> (B, max) bool, True where t < length."""
    if max_length is None:
        max_length = int(length.max())
    return torch.arange(int(max_length), dtype=length.dtype)[None, :] < length[:, None]


def generate_path(duration, mask):
    """Monotonic 0/1 alignment: frame f belongs to phone p iff
    cum[p-1] <= f < cum[p]  (utils/model.py:37-47).  duration: (B, Tp) float or
    int; mask: (B, Tp, Tf).  Returns mask.dtype."""
    cum = torch.cumsum(duration, dim=1)
    lo = F.pad(cum, (1, 0))[:, :-1]
    f = torch.arange(mask.shape[2], dtype=cum.dtype)[None, None, :]
    inside = (f < cum[:, :, None]) & ~(f < lo[:, :, None])
    return inside.to(mask.dtype) * mask


def frame_to_phone_index(duration_int, n_frames):
    """Integer form of generate_path: index of the phone owning each frame
    (-1 past the end).  duration_int: (B, Tp) int64."""
    cum = torch.cumsum(duration_int, dim=1)
    f = torch.arange(n_frames)[None, :, None]
    idx = (f >= cum[:, None, :]).sum(-1)
    return torch.where(idx < duration_int.shape[1], idx, torch.full_like(idx, -1))


# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _conv(sd, p, x, **kw):
    return F.conv1d(x, sd[p + ".weight"], sd.get(p + ".bias"), **kw)


def layer_norm_last(x, w, b, eps):
    """LayerNorm over the last dim, biased variance."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def layer_norm_c(x, gamma, beta, eps=1e-5):
    """LayerNorm over dim 1 of (B, C, T) (layers/norm.py:19-32 and
    modules/frame_prior.py:22-34; gamma/beta (1,C,1) or (C,))."""
    return layer_norm_last(x.transpose(1, 2), gamma.reshape(-1), beta.reshape(-1), eps).transpose(1, 2)


def sinusoid(pos, d):
    """rows: sin/cos interleaved encodings of real positions ``pos`` (n,)."""
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(pos.shape[0], d)
    ang = pos.float()[:, None] * div[None, :]
    pe[:, 0::2] = torch.sin(ang)
    pe[:, 1::2] = torch.cos(ang)
    return pe


# --------------------------------------------------------------------------
# anti-aliased snake  (promptttspp/layers/activations.py:22-138)
# --------------------------------------------------------------------------


def aa_snake(x, log_alpha, f_up, f_dn):
    """x: (B, C, T); log_alpha: (C,); f_up/f_dn: (12,) taps.

    Polyphase restatement of replicate-pad(5) -> conv_transpose(stride 2) x2 ->
    crop(15,15) -> snake -> replicate-pad(5,6) -> conv(stride 2):
      u[2q]   = 2 sum_a x[q-3+a] f[11-2a],  u[2q+1] = 2 sum_a x[q-2+a] f[10-2a]
      y[t]    = sum_j s[2t + j - 5] f[j]      (indices clamped = replicate pad)
    """
    B, C, T = x.shape
    f_up, f_dn = f_up.reshape(-1), f_dn.reshape(-1)
    win = F.pad(x, (3, 3), mode="replicate").unfold(-1, 7, 1)  # (B,C,T,7): x[q-3+i]
    even = (win[..., 0:6] * f_up[[11, 9, 7, 5, 3, 1]]).sum(-1)
    odd = (win[..., 1:7] * f_up[[10, 8, 6, 4, 2, 0]]).sum(-1)
    u = 2.0 * torch.stack([even, odd], dim=-1).reshape(B, C, 2 * T)
    ea = torch.exp(log_alpha).reshape(1, C, 1)
    s = u + (1.0 / (ea + 1e-9)) * torch.sin(u * ea) ** 2
    sw = F.pad(s, (5, 6), mode="replicate").unfold(-1, 12, 2)  # (B,C,T,12)
    return (sw * f_dn).sum(-1)


# --------------------------------------------------------------------------
# BigVGAN  (promptttspp/vocoders/bigvgan.py:21-131)
# --------------------------------------------------------------------------


def _wn(sd, p):
    """weight-norm fold: w = g * v / ||v|| over all dims but 0."""
    if p + ".weight_g" in sd:
        v, g = sd[p + ".weight_v"], sd[p + ".weight_g"]
        return g * v / v.flatten(1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
    return sd[p + ".weight"]


def _act(sd, p, x):
    return aa_snake(x, sd[p + ".act.alpha"].reshape(-1), sd[p + ".up.filter"], sd[p + ".down.lowpass.filter"])


def amp_layer(sd, p, x, ks, dil):
    y = _act(sd, p + ".act1", x)
    y = F.conv1d(y, _wn(sd, p + ".conv1"), sd[p + ".conv1.bias"], padding=(ks * dil - dil) // 2, dilation=dil)
    y = _act(sd, p + ".act2", y)
    y = F.conv1d(y, _wn(sd, p + ".conv2"), sd[p + ".conv2.bias"], padding=ks // 2)
    return x + y


def bigvgan(sd, x, rates=(6, 5, 4, 2), res_ks=(3, 7, 11), res_dil=((1, 3, 5),) * 3, source=None, p=""):
    """x: (B, 80, T) -> (B, 1, T*prod(rates)).  ``source``: optional harmonic
    source (B, 1, T*prod(rates)) for the F0-aware variant (bigvgan_f0.py:98-115)."""
    h = F.conv1d(x, _wn(sd, p + "conv_pre"), sd[p + "conv_pre.bias"], padding=3)
    for s, u in enumerate(rates):
        h = F.conv_transpose1d(
            h, _wn(sd, f"{p}upsamples.{s}"), sd[f"{p}upsamples.{s}.bias"], stride=u, padding=u // 2 + u % 2,
            output_padding=u % 2,
        )
        if source is not None:
            stride = int(np.prod(rates[s + 1 :])) if s + 1 < len(rates) else 1
            nc = f"{p}noise_convs.{s}"
            if s + 1 < len(rates):
                h = h + F.conv1d(source, sd[nc + ".weight"], sd[nc + ".bias"], stride=stride, padding=stride // 2)
            else:
                h = h + F.conv1d(source, sd[nc + ".weight"], sd[nc + ".bias"])
        acc = 0
        for b, ks in enumerate(res_ks):
            y = h
            for l, d in enumerate(res_dil[b]):
                y = amp_layer(sd, f"{p}mrfs.{s}.{b}.layers.{l}", y, ks, d)
            acc = acc + y
        h = acc / len(res_ks)
    h = _act(sd, p + "act_post", h)
    h = F.conv1d(h, _wn(sd, p + "conv_post"), sd[p + "conv_post.bias"], padding=3)
    return torch.tanh(h)


# --------------------------------------------------------------------------
# NSF harmonic source  (promptttspp/vocoders/nsf.py:49-148, 193-206)
# --------------------------------------------------------------------------


def nsf_source(sd, f0, rand_ini, noise, sampling_rate=24000, harmonic_num=8, sine_amp=0.1, noise_std=0.003,
               upsample=240, p="m_source."):
    """f0: (B, 1, Tf) Hz (0 = unvoiced); rand_ini: (B, harmonic_num+1) initial
    phases in [0,1) (column 0 is forced to 0); noise: (B, Tf*upsample,
    harmonic_num+1) standard normal.  Returns (B, 1, Tf*upsample)."""
    f0u = f0.repeat_interleave(upsample, dim=-1).transpose(1, 2)  # nearest upsample, (B, L, 1)
    harm = torch.arange(1, harmonic_num + 2, dtype=torch.float32)
    fb = f0u * harm  # (B, L, H)
    rad = (fb / sampling_rate) % 1
    ini = rand_ini.clone()
    ini[:, 0] = 0
    rad[:, 0, :] = rad[:, 0, :] + ini
    wrapped = torch.cumsum(rad, 1) % 1
    over = (wrapped[:, 1:, :] - wrapped[:, :-1, :]) < 0
    shift = torch.zeros_like(rad)
    shift[:, 1:, :] = over * -1.0
    sines = torch.sin(torch.cumsum(rad + shift, dim=1) * 2 * np.pi) * sine_amp
    uv = (f0u > 0).float()
    namp = uv * noise_std + (1 - uv) * sine_amp / 3
    sines = sines * uv + namp * noise
    merged = torch.tanh(F.linear(sines, sd[p + "l_linear.weight"], sd[p + "l_linear.bias"]))
    return merged.transpose(1, 2)
