"""GST reference encoder (reference: promptttspp/modules/reference_encoder.py:21-124):
6 x (Conv2d 3x3 stride 2, no bias -> BatchNorm2d -> ReLU) on (B,1,T,80), then a GRU
whose last valid hidden state is the reference embedding.

MI355X design: the spectrogram stays channels-last (B, T, F, C); each Conv2d is an
im2col gather + ONE MFMA GEMM launch (K = 9*Cin, the same implicit-GEMM kernel as
every other dense layer), BatchNorm2d(+ReLU) is the fused batch-statistics kernel
pair of bn_dw.hip, and the GRU input projection for all (<= ~12) steps is one GEMM.
Nothing here goes through MIOpen, whose per-shape solver search cannot cope with
token-bucket batches that change shape every step.

Differences from the reference's op sequence that do not change results: the GRU
runs on the padded batch and freezes each sequence's state after its last valid
step (the packed-sequence semantics), which removes the reference's lengths->CPU
copy + pack_padded_sequence host sync (reference_encoder.py:118-121).
"""
import torch
import torch.nn as nn

from .. import functional as PF
from .. import nn_ops as NO
from ..config import compute_dtype


class ReferenceEncoder(nn.Module):
    def __init__(self, idim=80, conv_layers=6, conv_chans_list=(32, 32, 64, 64, 128, 128), conv_kernel_size=3,
                 conv_stride=2, gru_layers=1, gru_units=128):
        super().__init__()
        assert conv_kernel_size % 2 == 1, "kernel size must be odd."
        assert len(conv_chans_list) == conv_layers
        if conv_kernel_size != 3 or conv_stride != 2 or gru_layers != 1:
            raise NotImplementedError("promptttspp_amd implements the reference config: 3x3 stride-2 convs, 1 GRU layer")
        self.conv_stride, self.conv_layers = conv_stride, conv_layers
        self.chans = tuple(conv_chans_list)
        padding = (conv_kernel_size - 1) // 2
        convs = []
        for i in range(conv_layers):
            cin = 1 if i == 0 else conv_chans_list[i - 1]
            convs += [nn.Conv2d(cin, conv_chans_list[i], conv_kernel_size, stride=conv_stride, padding=padding, bias=False),
                      nn.BatchNorm2d(conv_chans_list[i]), nn.ReLU(inplace=True)]
        self.convs = nn.Sequential(*convs)  # parameter holders (reference key names convs.{0,1,3,4,...})
        f = idim
        for _ in range(conv_layers):
            f = (f - conv_kernel_size + 2 * padding) // conv_stride + 1
        self.gru = nn.GRU(f * conv_chans_list[-1], gru_units, gru_layers, batch_first=True)

    def forward(self, speech, in_lens=None):
        """speech (B, idim, T) float -> (B, gru_units, 1) float32."""
        B, _, T = speech.shape
        if self.training and PF.STACK_DRIVERS and speech.is_cuda and all(c % 8 == 0 for c in self.chans):
            x = NO.refenc_convs(speech.transpose(1, 2).to(compute_dtype()), [self.convs[3 * i] for i in range(self.conv_layers)],
                                [self.convs[3 * i + 1] for i in range(self.conv_layers)])
        else:
            x = speech.transpose(1, 2).unsqueeze(-1).to(compute_dtype()).contiguous()  # (B, T, F, 1)
            for i in range(self.conv_layers):
                x = NO.conv2d_3x3s2(x, self.convs[3 * i].weight)
                x = NO.batch_norm_act(x, self.convs[3 * i + 1], act="relu")
        # reference flattens (B, C, T', F') -> (B, T', C*F'): channel-major features
        hs = x.permute(0, 1, 3, 2).reshape(B, x.shape[1], -1)
        if in_lens is None:
            lens = torch.full((B,), hs.shape[1], device=speech.device, dtype=torch.long)
        else:
            lens = torch.ceil(in_lens.to(speech.device).float() / (self.conv_stride**self.conv_layers)).long().clamp(min=1)
        g = self.gru
        ref = NO.gru_last_state(hs.contiguous(), g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0, lens)
        return ref.unsqueeze(-1)
