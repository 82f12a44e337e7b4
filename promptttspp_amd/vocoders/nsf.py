"""Harmonic-plus-noise source of the F0-aware vocoder (reference:
promptttspp/vocoders/nsf.py:31-206): sine generator with `harmonic_num` overtones,
phase by cumulative sum of f0/sr with wrap compensation, U/V-gated noise, then a
Linear(harmonics+1 -> 1) + tanh merge.

The source is (B, Tf*240, 9) -- 0.1 % of the vocoder's arithmetic -- and is a chain of
elementwise ops plus two prefix scans over time; it stays on torch tensor ops in float32
(the scans are torch.cumsum over the innermost dimension of a (B, dim, L) copy).  Same RNG draw order as the reference (rand for the initial
phases, randn_like for the additive noise, one unused randn_like in SourceModuleHnNSF),
so seeding reproduces the reference stream on the same device."""
import numpy as np
import torch
import torch.nn as nn


FUSED_SOURCE = __import__("os").environ.get("PTPP_NSF_FUSED", "1") != "0"  # the one-launch source on device tensors (csrc/nsf.hip)


class SineGen(nn.Module):
    def __init__(self, samp_rate, harmonic_num=0, sine_amp=0.1, noise_std=0.003, voiced_threshold=0, flag_for_pulse=False):
        super().__init__()
        if flag_for_pulse:
            raise NotImplementedError("flag_for_pulse is not used by the reference vocoders")
        self.sine_amp, self.noise_std = sine_amp, noise_std
        self.harmonic_num, self.dim = harmonic_num, harmonic_num + 1
        self.sampling_rate, self.voiced_threshold = samp_rate, voiced_threshold

    def _f02uv(self, f0):
        return (f0 > self.voiced_threshold).to(f0.dtype)

    def _f02sine(self, f0_values):
        """(B, L, dim) instantaneous frequencies -> sines; the integer part of the phase is
        dropped ((x-1)*2pi == x*2pi) by subtracting 1 wherever the wrapped cumsum decreases."""
        rad = (f0_values / self.sampling_rate) % 1
        rand_ini = torch.rand(f0_values.shape[0], f0_values.shape[2], device=f0_values.device)
        rand_ini[:, 0] = 0
        rad[:, 0, :] = rad[:, 0, :] + rand_ini
        # the two prefix scans run with time as the INNERMOST dimension: torch's outer-dimension scan of the
        # (B, L, dim) layout took 24 ms per call at 32 x 141 600 x 9 (48 of the 247 ms of the config-5 app path,
        # profiles/r02b_app_path.md); the transposed copies cost two 160 MB passes
        rad = rad.transpose(1, 2).contiguous()  # (B, dim, L)
        wrapped = torch.cumsum(rad, 2) % 1
        shift = torch.zeros_like(rad)
        shift[:, :, 1:] = ((wrapped[:, :, 1:] - wrapped[:, :, :-1]) < 0) * -1.0
        return torch.sin(torch.cumsum(rad + shift, dim=2) * 2 * np.pi).transpose(1, 2)

    @torch.no_grad()
    def forward(self, f0):
        """f0: (B, L, 1) Hz, 0 for unvoiced -> (sine_waves (B,L,dim), uv, noise)."""
        harm = torch.arange(1, self.dim + 1, device=f0.device, dtype=f0.dtype)
        sine = self._f02sine(f0 * harm) * self.sine_amp
        uv = self._f02uv(f0)
        noise = (uv * self.noise_std + (1 - uv) * self.sine_amp / 3) * torch.randn_like(sine)
        return sine * uv + noise, uv, noise


class SourceModuleHnNSF(nn.Module):
    def __init__(self, sampling_rate, harmonic_num=0, sine_amp=0.1, add_noise_std=0.003, voiced_threshod=0):
        super().__init__()
        self.sine_amp, self.noise_std = sine_amp, add_noise_std
        self.l_sin_gen = SineGen(sampling_rate, harmonic_num, sine_amp, add_noise_std, voiced_threshod)
        self.l_linear = nn.Linear(harmonic_num + 1, 1)
        self.l_tanh = nn.Tanh()

    def forward(self, x):
        if FUSED_SOURCE and not (torch.is_grad_enabled() and any(p.requires_grad for p in self.l_linear.parameters())):  # (no backward)
            from .. import ops

            g = self.l_sin_gen
            if x.dim() == 3 and x.shape[2] == 1 and ops.nsf_source_ok(x, g.dim):
                # ONE launch (csrc/nsf.hip) instead of ~20 elementwise passes and two scans over (B, L, dim); the same three RNG
                # draws in the same order and shapes as the reference (rand: initial phases, randn_like: the additive noise of
                # the sines, randn_like: SourceModuleHnNSF's unused noise)
                B, L, _ = x.shape
                rand_ini = torch.rand(B, g.dim, device=x.device)
                rand_ini[:, 0] = 0
                nz = torch.randn_like(torch.empty((B, L, g.dim), device=x.device, dtype=torch.float32))
                merged = ops.nsf_source(x.reshape(B, L), rand_ini, nz, self.l_linear.weight, float(self.l_linear.bias.detach()),
                                        g.sampling_rate, g.sine_amp, g.noise_std, g.voiced_threshold)
                uv = g._f02uv(x)
                noise = torch.randn_like(uv) * self.sine_amp / 3
                return merged.unsqueeze(-1), noise, uv
        sine_wavs, uv, _ = self.l_sin_gen(x)
        sine_merge = self.l_tanh(self.l_linear(sine_wavs))
        noise = torch.randn_like(uv) * self.sine_amp / 3  # drawn (and unused) like the reference
        return sine_merge, noise, uv
