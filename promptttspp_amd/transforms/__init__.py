"""Mel front-end ``MelSpectrogramTransform`` (reference: promptttspp/transforms/__init__.py, a
subclass of ``torchaudio.transforms.MelSpectrogram``).  torchaudio is not vendored in the reference
and absent from this image, so this restates its published algorithm with torch ops: Hann-windowed
STFT (center, reflect padding), |.|^power, slaney mel filterbank with slaney area normalisation,
log(clamp(1e-5)).  PARITY UNPINNED: no torchaudio here to generate golden vectors against; it sits
on the data side, outside the model hot path."""
import math

import torch
import torch.nn as nn


def _hz_to_mel_slaney(f):
    f = float(f)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return min_log_mel + math.log(f / min_log_hz) / logstep if f >= min_log_hz else f / f_sp


def _mel_to_hz_slaney(m):
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    hz = m * f_sp
    log_t = m >= min_log_mel
    hz[log_t] = min_log_hz * torch.exp(logstep * (m[log_t] - min_log_mel))
    return hz


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm="slaney", mel_scale="slaney"):
    """(n_freqs, n_mels) triangular filterbank (torchaudio.functional.melscale_fbanks)."""
    assert mel_scale == "slaney", "only the slaney scale of the reference config is restated"
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_mel_slaney(f_min), _hz_to_mel_slaney(f_max), n_mels + 2)
    f_pts = _mel_to_hz_slaney(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.minimum(down, up), min=0)
    if norm == "slaney":
        fb = fb * (2.0 / (f_pts[2 : n_mels + 2] - f_pts[:n_mels])).unsqueeze(0)
    return fb


class MelSpectrogramTransform(nn.Module):
    def __init__(self, sample_rate=24000, n_fft=512, win_length=None, hop_length=None, f_min=0.0, f_max=None,
                 n_mels=80, power=2.0, center=True, pad_mode="reflect", norm=None, mel_scale="htk", **_unused):
        super().__init__()
        self.sample_rate, self.n_fft = sample_rate, n_fft
        self.win_length = win_length or n_fft
        self.hop_length = hop_length or self.win_length // 2
        self.power, self.center, self.pad_mode, self.n_mels = power, center, pad_mode, n_mels
        self.register_buffer("window", torch.hann_window(self.win_length), persistent=False)
        self.register_buffer("fb", melscale_fbanks(n_fft // 2 + 1, f_min, f_max or sample_rate / 2, n_mels, sample_rate,
                                                   norm, mel_scale), persistent=False)

    def to_spec(self, wav):
        shape = wav.shape
        spec = torch.stft(wav.reshape(-1, shape[-1]), self.n_fft, self.hop_length, self.win_length, self.window,
                          center=self.center, pad_mode=self.pad_mode, return_complex=True).abs()
        if self.power != 1:
            spec = spec.pow(self.power)
        return spec.reshape(shape[:-1] + spec.shape[-2:])

    def spec_to_mel(self, spec):
        mel = torch.matmul(spec.transpose(-1, -2), self.fb).transpose(-1, -2)
        return mel.clamp_min(1e-5).log()

    def to_mel(self, wav):
        return self.spec_to_mel(self.to_spec(wav))

    def forward(self, wav):
        return self.to_mel(wav)
