"""F0-aware BigVGAN (reference: promptttspp/vocoders/bigvgan_f0.py:25-123): BigVGAN whose
every upsampling stage adds a strided Conv1d of a harmonic source signal.  Reuses the HIP
generator of bigvgan.py; each `noise_conv` (Conv1d 1 -> C, kernel 2s, stride s) is a
strided window view of the source times a (k x C) matrix, i.e. one small GEMM launch
writing channels-last output that is added to the stage input."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import remove_weight_norm, weight_norm

from .. import functional as PF
from .. import ops
from ..layers.activations import AntiAliasActivation
from .bigvgan import AMPBlock, BigVGAN
from .nsf import SourceModuleHnNSF


class F0AwareBigVGAN(BigVGAN):
    def __init__(self, sampling_rate, harmonic_num, in_channel, upsample_initial_channel, upsample_rates,
                 upsample_kernel_sizes, resblock_kernel_sizes, resblock_dilations):
        nn.Module.__init__(self)
        self.num_kernels = len(resblock_kernel_sizes)
        self.upsample_rates = list(upsample_rates)
        self.f0_up = nn.Upsample(scale_factor=int(np.prod(upsample_rates)))
        self.m_source = SourceModuleHnNSF(sampling_rate=sampling_rate, harmonic_num=harmonic_num)
        self.noise_convs = nn.ModuleList()
        self.conv_pre = weight_norm(nn.Conv1d(in_channel, upsample_initial_channel, kernel_size=7, stride=1, padding=3))
        self.upsamples = nn.ModuleList()
        for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            self.upsamples.append(weight_norm(nn.ConvTranspose1d(
                upsample_initial_channel // (2**i), upsample_initial_channel // (2 ** (i + 1)), kernel_size=k, stride=u,
                padding=u // 2 + u % 2, output_padding=u % 2)))
            cout = upsample_initial_channel // (2 ** (i + 1))
            if i + 1 < len(upsample_rates):
                s = int(np.prod(upsample_rates[i + 1:]))
                self.noise_convs.append(nn.Conv1d(1, cout, kernel_size=s * 2, stride=s, padding=s // 2))
            else:
                self.noise_convs.append(nn.Conv1d(1, cout, 1))
        self.mrfs = nn.ModuleList()
        for i in range(len(self.upsamples)):
            channel = upsample_initial_channel // (2 ** (i + 1))
            self.mrfs.append(nn.ModuleList([AMPBlock(channel, kernel_size=k, dilations=d)
                                            for k, d in zip(resblock_kernel_sizes, resblock_dilations)]))
        self.act_post = AntiAliasActivation(channel)
        self.conv_post = weight_norm(nn.Conv1d(channel, 1, kernel_size=7, stride=1, padding=3))
        from ..config import compute_dtype

        self.compute_dtype = compute_dtype()
        self._packed = None
        self._packed_key = None
        self.parallel_blocks = True
        self.fuse_amp_layers = True
        self.fuse_wide_layers = True
        self.wide_streams = False
        self._streams = None

    def _source_term(self, s, h, src):
        """noise_convs[s](har_source) as channels-last (B, T_s, C): strided windows x matrix."""
        conv = self.noise_convs[s]
        k, st, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        win = F.pad(src, (pad, pad)).unfold(-1, k, st)  # (B, T_s, k) view of (B, L)
        win = win[:, : h.shape[1]].to(h.dtype).contiguous()
        return PF.linear(win, conv.weight.reshape(conv.out_channels, k), conv.bias)

    @torch.no_grad()
    def forward(self, x, f0):
        """x: (B, in_channel, T) mel, f0: (B, 1, T) Hz (0 = unvoiced) -> (B, 1, T*hop)."""
        har, _, _ = self.m_source(self.f0_up(f0).transpose(-1, -2))  # (B, L, 1)
        src = har.squeeze(-1).float()

        def hook(s, h):
            return h + self._source_term(s, h, src)

        h, pk = self._generate(x, source_hook=hook)
        return self._post(h, pk).unsqueeze(1)
