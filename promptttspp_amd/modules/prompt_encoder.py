"""Prompt encoder: BERT CLS state -> 3-layer MLP (reference:
promptttspp/modules/prompt_encoder.py:22-56).

The BERT encoder itself is the third-party HuggingFace ``BertModel`` in the
reference too (un-vendored dependency).  Round-1 state: it runs here as the same
HF module on PyTorch-ROCm library ops; frozen layers run under ``no_grad``.  Only
the adaptor MLP goes through the HIP GEMM.  (A BERT encoder on the package's own
GEMM / LayerNorm / plain-attention kernels is the planned replacement:
``ptpp_attention_fwd`` already has the PLAIN variant for it.)

Offline behaviour: ``from_pretrained`` needs network/cache; when it is not
available the wrapper builds the architecture from ``BertConfig()`` (weights then
come from the model checkpoint, whose state dict contains
``prompt_encoder.bert.model.*``) and accepts pre-tokenised prompts
``(input_ids, attention_mask)`` in place of ``List[str]``.
"""
import warnings
from typing import List

import torch
import torch.nn as nn

from .. import functional as PF
from ..config import compute_dtype


class BertWrapper(nn.Module):
    def __init__(self, class_name="bert-base-uncased"):
        super().__init__()
        from transformers import BertConfig, BertModel, BertTokenizer

        try:
            self.model = BertModel.from_pretrained(class_name, local_files_only=True)
        except Exception as e:  # offline box: architecture only
            warnings.warn(f"BertModel.from_pretrained({class_name!r}) unavailable ({type(e).__name__}); "
                          "building bert-base from BertConfig() -- load weights via the model checkpoint")
            self.model = BertModel(BertConfig())
        try:
            self.tokenizer = BertTokenizer.from_pretrained(class_name, local_files_only=True)
            if len(self.tokenizer) < 1000:  # transformers may return a stub vocabulary offline
                self.tokenizer = None
        except Exception:
            self.tokenizer = None
        for p in self.model.parameters():
            p.requires_grad = False
        for p in self.model.encoder.layer[-1].attention.parameters():
            p.requires_grad = True

    def tokenize(self, prompts, device):
        if isinstance(prompts, (tuple, list)) and len(prompts) == 2 and isinstance(prompts[0], torch.Tensor):
            return prompts[0].to(device), prompts[1].to(device)
        if self.tokenizer is None:
            raise RuntimeError("no BERT vocabulary available offline: pass (input_ids, attention_mask) tensors")
        enc = self.tokenizer(list(prompts), padding=True, return_tensors="pt").to(device)
        return enc["input_ids"], enc["attention_mask"]

    def forward(self, prompts: List[str], device: torch.device) -> torch.Tensor:
        ids, am = self.tokenize(prompts, device)
        amp = compute_dtype() == torch.bfloat16 and ids.is_cuda
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            # BertModel.forward, layer by layer (bit-identical, checked on CPU): the library's
            # mask helper inspects the mask on the HOST (mask.all()), a device sync that stalls
            # the launch queue once per step.  Same modules, same state-dict keys.
            h = self.model.embeddings(input_ids=ids)
            ext = (1.0 - am[:, None, None, :].to(h.dtype)) * torch.finfo(h.dtype).min
            for layer in self.model.encoder.layer:
                out = layer(h, attention_mask=ext)
                h = out[0] if isinstance(out, tuple) else out
        return h[:, 0, :].float()


class PromptEncoder(nn.Module):
    def __init__(self, model_name, in_channels, mid_channels, out_channels):
        super().__init__()
        self.bert = BertWrapper(model_name)
        self.adaptor = nn.Sequential(
            nn.Linear(in_channels, mid_channels), nn.ReLU(inplace=True),
            nn.Linear(mid_channels, mid_channels), nn.ReLU(inplace=True),
            nn.Linear(mid_channels, out_channels),
        )

    def forward(self, prompts, device):
        """-> (B, out_channels, 1) float32"""
        x = self.bert(prompts, device)  # (B, 768) f32
        a = self.adaptor
        h = PF.linear(x, a[0].weight, a[0].bias, act="relu")
        h = PF.linear(h, a[2].weight, a[2].bias, act="relu")
        return PF.linear(h, a[4].weight, a[4].bias).unsqueeze(-1)
