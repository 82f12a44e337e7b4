"""Phoneme embedding (reference: promptttspp/layers/embedding.py:21-36)."""
import math

import torch
import torch.nn as nn


class PhonemeEmbedding(nn.Module):
    def __init__(self, num_vocab, channels, do_scale=True, init_normal=True):
        super().__init__()
        self.emb = nn.Embedding(num_vocab, channels, padding_idx=0)
        if init_normal:
            nn.init.normal_(self.emb.weight, 0.0, channels**-0.5)
        self.scale = math.sqrt(channels)
        self.do_scale = do_scale

    def forward_cl(self, ids, mask_bt1, dtype, lengths=None):
        """ids (B,T) int64, mask (B,T,1) (or, on the GPU, int32 ``lengths`` (B,)) -> (B,T,C) channels-last in `dtype`:
        one launch each way on the GPU (ptpp_embed_cl_fwd / _bwd), torch's gather on the CPU."""
        if lengths is not None and ids.is_cuda and ids.dtype == torch.int64 and self.emb.weight.shape[1] % 4 == 0:
            from .. import functional as PF

            return PF.embed_cl(ids, self.emb.weight, lengths, self.scale if self.do_scale else None, dtype, self.emb.padding_idx)
        if mask_bt1 is None:
            mask_bt1 = (torch.arange(ids.shape[1], device=ids.device)[None, :] < lengths[:, None]).unsqueeze(-1).float()
        x = self.emb(ids)
        if self.do_scale:
            x = x * self.scale
        return (x * mask_bt1.to(x.dtype)).to(dtype)

    def forward(self, x, mask):
        """Reference signature: ids (B,T), mask (B,1,T) -> (B,C,T)."""
        y = self.emb(x)
        if self.do_scale:
            y = y * self.scale
        return y.transpose(-1, -2) * mask
