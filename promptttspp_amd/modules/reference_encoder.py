"""GST reference encoder (reference: promptttspp/modules/reference_encoder.py:21-124):
6 x (Conv2d 3x3 stride 2, no bias -> BatchNorm2d -> ReLU) on (B,1,T,80), then a GRU
whose last valid hidden state is the reference embedding.

Round-1 state: this 3 MFLOP/frame side branch still runs on PyTorch-ROCm library
ops (MIOpen conv/BN, rocBLAS GRU) -- it is the next module to move onto the HIP
conv kernel (a 3x3 stride-2 Conv2d over (T, F) is a conv1d over T with
K = 3*F_in*C_in).  Two deliberate differences from the reference's op sequence,
neither changing results: the GRU runs on the padded batch and the state at
step len-1 is gathered (the GRU is causal), which removes the reference's
lengths->CPU copy + pack_padded_sequence host sync (reference_encoder.py:118-121).
"""
import torch
import torch.nn as nn

from ..config import compute_dtype


class ReferenceEncoder(nn.Module):
    def __init__(self, idim=80, conv_layers=6, conv_chans_list=(32, 32, 64, 64, 128, 128), conv_kernel_size=3,
                 conv_stride=2, gru_layers=1, gru_units=128):
        super().__init__()
        assert conv_kernel_size % 2 == 1, "kernel size must be odd."
        assert len(conv_chans_list) == conv_layers
        self.conv_stride, self.conv_layers = conv_stride, conv_layers
        padding = (conv_kernel_size - 1) // 2
        convs = []
        for i in range(conv_layers):
            cin = 1 if i == 0 else conv_chans_list[i - 1]
            convs += [nn.Conv2d(cin, conv_chans_list[i], conv_kernel_size, stride=conv_stride, padding=padding, bias=False),
                      nn.BatchNorm2d(conv_chans_list[i]), nn.ReLU(inplace=True)]
        self.convs = nn.Sequential(*convs)
        f = idim
        for _ in range(conv_layers):
            f = (f - conv_kernel_size + 2 * padding) // conv_stride + 1
        self.gru = nn.GRU(f * conv_chans_list[-1], gru_units, gru_layers, batch_first=True)

    def forward(self, speech, in_lens=None):
        """speech (B, idim, T) float -> (B, gru_units, 1) float32."""
        B = speech.size(0)
        amp = compute_dtype() == torch.bfloat16 and speech.is_cuda
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            hs = self.convs(speech.transpose(1, 2).unsqueeze(1)).transpose(1, 2)  # (B, T', C', F')
            hs = hs.contiguous().view(B, hs.size(1), -1)
            self.gru.flatten_parameters()
            out, h_last = self.gru(hs)
        if in_lens is None:
            ref = h_last[-1]
        else:
            lens = torch.ceil(in_lens.to(speech.device).float() / (self.conv_stride**self.conv_layers)).long().clamp(min=1)
            ref = out.gather(1, (lens - 1).view(B, 1, 1).expand(-1, 1, out.size(-1))).squeeze(1)
        return ref.float().unsqueeze(-1)
