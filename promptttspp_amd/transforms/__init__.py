"""Mel front-end ``MelSpectrogramTransform`` (reference: promptttspp/transforms/mel.py:18-34, a subclass of
``torchaudio.transforms.MelSpectrogram``; conf/transforms/mel.yaml: 24 kHz, n_fft 512, win 480, hop 240, 80 slaney
mels 63..12000 Hz) -- the step immediately before the hot path (app.py:93-96,
egs/proposed/bin/compute_mel.py:59-68).

torchaudio is not vendored in the reference and absent from this image: its published algorithm is restated --
Hann-windowed STFT (center, reflect padding), |.|^power, slaney filterbank with slaney area normalisation,
log(clamp(1e-5)).  On a ROCm device the two contractions run on the package's GEMM in exact f32: the windowed DFT of
all frames is ONE (frames x n_fft) x (n_fft x 2*bins) product against a cached [window*cos | -window*sin] basis, the
filterbank a second product; CPU tensors go through torch.stft.  Parity: pinned against an independent numpy
restatement (oracle/ref_torch.py::mel_spectrogram_np, itself checked against torch.stft); NOT pinned against
torchaudio, which cannot be run here."""
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

    def _basis(self, device):
        """(2 * PB, n_fft) f32 rows [w*cos(2 pi k n / N) | -w*sin(...)], PB = bins padded to a multiple of 8, with the
        window centred in the n_fft frame like torch.stft; packed once per device for the GEMM kernel."""
        key = str(device)
        ent = getattr(self, "_basis_cache", {}).get(key)
        if ent is None:
            from .. import ops

            N, nb = self.n_fft, self.n_fft // 2 + 1
            pb = (nb + 7) // 8 * 8
            w = torch.zeros(N, dtype=torch.float64)
            off = (N - self.win_length) // 2
            w[off : off + self.win_length] = self.window.double().cpu()
            ang = 2.0 * math.pi * torch.arange(nb, dtype=torch.float64)[:, None] * torch.arange(N, dtype=torch.float64)[None, :] / N
            basis = torch.zeros(2 * pb, N, dtype=torch.float64)
            basis[:nb] = torch.cos(ang) * w
            basis[pb : pb + nb] = -torch.sin(ang) * w
            fb = torch.zeros(self.n_mels, pb, dtype=torch.float32)
            fb[:, :nb] = self.fb.t().float().cpu()
            ent = (ops.pack_conv_weight(basis.float().to(device), torch.float32), ops.pack_conv_weight(fb.to(device), torch.float32), pb)
            if not hasattr(self, "_basis_cache"):
                self._basis_cache = {}
            self._basis_cache[key] = ent
        return ent

    def _power_spec_hip(self, wav):
        """(..., L) device tensor -> (rows, frames, PB) power (or magnitude) spectrum, channels-last, via the GEMM."""
        from .. import ops

        x = wav.reshape(-1, wav.shape[-1]).float()
        if self.center:
            x = torch.nn.functional.pad(x.unsqueeze(1), (self.n_fft // 2, self.n_fft // 2), mode=self.pad_mode).squeeze(1)
        frames = x.unfold(-1, self.n_fft, self.hop_length).contiguous()            # (rows, F, n_fft)
        basis, _, pb = self._basis(wav.device)
        ri = ops.conv1d(frames, basis, None, 2 * pb)                                 # (rows, F, [re | im])
        spec = ri[..., :pb] ** 2 + ri[..., pb:] ** 2
        if self.power == 1:
            spec = spec.sqrt()
        elif self.power != 2:
            spec = spec.pow(self.power / 2.0)
        return spec

    def to_spec(self, wav):
        if wav.is_cuda:
            nb = self.n_fft // 2 + 1
            spec = self._power_spec_hip(wav)[..., :nb].transpose(-1, -2)
            return spec.reshape(wav.shape[:-1] + spec.shape[-2:])
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
        if wav.is_cuda:  # spectrum and filterbank both on the GEMM, no (bins, frames) transposes in between
            from .. import ops

            spec = self._power_spec_hip(wav).contiguous()
            mel = ops.conv1d(spec, self._basis(wav.device)[1], None, self.n_mels)    # (rows, F, n_mels)
            mel = mel.clamp_min(1e-5).log().transpose(-1, -2)
            return mel.reshape(wav.shape[:-1] + mel.shape[-2:])
        return self.spec_to_mel(self.to_spec(wav))

    def forward(self, wav):
        return self.to_mel(wav)
