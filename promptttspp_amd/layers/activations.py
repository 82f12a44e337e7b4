"""Anti-aliased Snake activation (reference: promptttspp/layers/activations.py:22-138).

Same module tree / buffers as the reference (``act.alpha`` (1,C,1) in the log
domain, ``up.filter`` and ``down.lowpass.filter`` (1,1,12)), but the whole
pad -> x2 up-FIR -> snake -> low-pass/decimate chain runs as ONE fused HIP
kernel (``ptpp_aa_snake_fwd``) on channels-last tensors.
"""
import math

import torch
import torch.nn as nn

from .. import ops


def kaiser_sinc_filter1d(cutoff, half_width, kernel_size):
    """Kaiser-windowed sinc low-pass, normalised to unit DC gain -> (1, 1, K).

    Window design rule (Kaiser's attenuation formula, as used by the reference
    at activations.py:47-72): A = 2.285 (K/2 - 1) pi (4 hw) + 7.95;
    beta = 0.1102 (A - 8.7) for A > 50, the 0.5842/0.07886 form for
    21 <= A <= 50, else 0."""
    half = kernel_size // 2
    atten = 2.285 * (half - 1) * math.pi * (4 * half_width) + 7.95
    if atten > 50.0:
        beta = 0.1102 * (atten - 8.7)
    elif atten >= 21.0:
        beta = 0.5842 * (atten - 21) ** 0.4 + 0.07886 * (atten - 21.0)
    else:
        beta = 0.0
    win = torch.kaiser_window(kernel_size, periodic=False, beta=beta)
    if kernel_size % 2 == 0:
        t = torch.arange(-half, half) + 0.5
    else:
        t = torch.arange(kernel_size) - half
    if cutoff == 0:
        taps = torch.zeros(kernel_size)
    else:
        taps = 2 * cutoff * win * torch.sinc(2 * cutoff * t)
        taps = taps / taps.sum()
    return taps.view(1, 1, kernel_size)


class _FilterHolder(nn.Module):
    """Holds a `filter` buffer under the reference's key names."""

    def __init__(self, ratio, kernel_size):
        super().__init__()
        self.ratio = ratio
        self.kernel_size = kernel_size
        self.register_buffer(
            "filter", kaiser_sinc_filter1d(cutoff=0.5 / ratio, half_width=0.6 / ratio, kernel_size=kernel_size)
        )


class UpSample1d(_FilterHolder):
    def __init__(self, ratio=2, kernel_size=None):
        super().__init__(ratio, int(6 * ratio // 2) * 2 if kernel_size is None else kernel_size)


class LowPassFilter1d(_FilterHolder):
    def __init__(self, ratio=2, kernel_size=12):
        super().__init__(ratio, kernel_size)


class DownSample1d(nn.Module):
    def __init__(self, ratio=2, kernel_size=None):
        super().__init__()
        self.ratio = ratio
        self.kernel_size = int(6 * ratio // 2) * 2 if kernel_size is None else kernel_size
        self.lowpass = LowPassFilter1d(ratio, self.kernel_size)


class Snake(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.alpha = nn.Parameter(torch.zeros(1, channels, 1))


class AntiAliasActivation(nn.Module):
    """y = down(snake(up(x))).  ``forward`` keeps the reference's (B, C, T)
    signature; ``forward_cl`` is the channels-last entry the vocoder uses."""

    def __init__(self, channels):
        super().__init__()
        self.channels = channels
        self.up = UpSample1d(2, 12)
        self.act = Snake(channels)
        self.down = DownSample1d(2, 12)
        self._taps = None  # host copies of the two 12-tap filters (kernel arguments)

    def _load_from_state_dict(self, *args, **kwargs):
        self._taps = None
        return super()._load_from_state_dict(*args, **kwargs)

    def taps(self):
        if self._taps is None:
            assert self.up.ratio == 2 and self.up.filter.numel() == 12 and self.down.lowpass.filter.numel() == 12
            self._taps = (ops._taps(self.up.filter), ops._taps(self.down.lowpass.filter))
        return self._taps

    def forward_cl(self, x, out=None):
        up, dn = self.taps()
        return ops.aa_snake(x, self.act.alpha.detach().reshape(-1).float().contiguous(), up, dn, out=out)

    def forward(self, x, dtype=torch.float32):
        y = self.forward_cl(ops.bct_to_btc(x, dtype))
        return ops.btc_to_bct(y)
