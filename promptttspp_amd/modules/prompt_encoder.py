"""Prompt encoder: BERT CLS state -> 3-layer MLP (reference:
promptttspp/modules/prompt_encoder.py:22-56).

The BERT encoder itself is the third-party HuggingFace ``BertModel`` in the
reference too (un-vendored dependency).  Round-1 state: it runs here as the same
HF module on PyTorch-ROCm library ops; frozen layers run under ``no_grad``.  Only
the adaptor MLP goes through the HIP GEMM.  (A BERT encoder on the package's own
GEMM / LayerNorm / plain-attention kernels is the planned replacement:
``ptpp_attention_fwd`` already has the PLAIN variant for it.)

Pretrained weights: like the reference, ``BertModel.from_pretrained(name)`` (local cache first, then the
hub).  When neither is reachable the constructor RAISES -- training the prompt encoder on 11 frozen layers of
random weights would be silently useless -- unless random initialisation is explicitly allowed
(``PTPP_ALLOW_RANDOM_BERT=1`` or ``allow_random_init=True``): tests, benchmarks, and runs whose model
checkpoint supplies ``prompt_encoder.bert.model.*`` anyway (the trainer / app / synthesize drivers set it when a
checkpoint path is configured).  Without a vocabulary the wrapper accepts pre-tokenised prompts
``(input_ids, attention_mask)`` in place of ``List[str]``.
"""
import os
import warnings
from typing import List

import torch
import torch.nn as nn

from .. import functional as PF
from ..config import compute_dtype


class allow_random_bert:
    """``with allow_random_bert():`` -- constructors inside may fall back to a randomly initialised BERT (the caller
    loads a checkpoint that holds ``prompt_encoder.bert.model.*``, or asked for a random model explicitly)."""

    def __enter__(self):
        self.old = os.environ.get("PTPP_ALLOW_RANDOM_BERT")
        os.environ["PTPP_ALLOW_RANDOM_BERT"] = "1"

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop("PTPP_ALLOW_RANDOM_BERT", None)
        else:
            os.environ["PTPP_ALLOW_RANDOM_BERT"] = self.old
        return False


class BertWrapper(nn.Module):
    def __init__(self, class_name="bert-base-uncased", allow_random_init=None):
        super().__init__()
        from transformers import BertConfig, BertModel, BertTokenizer

        def pretrained(cls):
            try:
                return cls.from_pretrained(class_name, local_files_only=True)  # the local cache, no network round trip
            except Exception:
                return cls.from_pretrained(class_name)  # the hub, like the reference (prompt_encoder.py:25-26)

        try:
            self.model = pretrained(BertModel)
        except Exception as e:
            if allow_random_init is None:
                allow_random_init = os.environ.get("PTPP_ALLOW_RANDOM_BERT", "") not in ("", "0")
            if not allow_random_init:
                raise RuntimeError(
                    f"BertModel.from_pretrained({class_name!r}) failed ({type(e).__name__}: {e}) and random "
                    "initialisation of the frozen BERT encoder was not allowed.  Provide the pretrained weights (HF "
                    "cache / network), or -- when a model checkpoint will supply prompt_encoder.bert.model.* -- set "
                    "PTPP_ALLOW_RANDOM_BERT=1.") from e
            warnings.warn(f"BertModel.from_pretrained({class_name!r}) unavailable ({type(e).__name__}); "
                          "building bert-base from BertConfig() -- weights must come from the model checkpoint")
            self.model = BertModel(BertConfig())
        try:
            self.tokenizer = pretrained(BertTokenizer)
            if len(self.tokenizer) < 1000:  # transformers may return a stub vocabulary offline
                self.tokenizer = None
        except Exception:
            self.tokenizer = None
        for p in self.model.parameters():
            p.requires_grad = False
        for p in self.model.encoder.layer[-1].attention.parameters():
            p.requires_grad = True
        self.hip_frozen_layers = True  # False: every layer through the library modules

    def _n_frozen(self, layers):
        """Number of leading encoder layers without a trainable parameter.  Walking ~200 parameters per
        forward cost ~0.5 ms of host time per step, so the count is cached; ``train()`` / ``eval()``
        (and ``refresh_frozen_layers()`` after changing ``requires_grad`` by hand) recompute it."""
        n = getattr(self, "_n_frozen_cache", None)
        if n is None:
            n = 0
            while n < len(layers) and not any(p.requires_grad for p in layers[n].parameters()):
                n += 1
            self._n_frozen_cache = n
        return n

    def refresh_frozen_layers(self):
        self._n_frozen_cache = None

    def train(self, mode=True):
        self._n_frozen_cache = None
        return super().train(mode)

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
        layers = list(self.model.encoder.layer)
        # BertModel.forward, layer by layer (checked bit-identical on CPU): the library's mask helper
        # inspects the mask on the HOST (mask.all()), a device sync that stalls the launch queue once per
        # step.  Same modules, same state-dict keys.
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            h = self.model.embeddings(input_ids=ids)
        n_hip = 0
        if ids.is_cuda and self.hip_frozen_layers:
            # frozen layers (no gradient flows into or through them: the embeddings are frozen too) run
            # forward-only on the HIP kernels: 7 launches per layer, weights packed once
            n_hip = self._n_frozen(layers)
            if n_hip and not (torch.is_grad_enabled() and h.requires_grad):
                lengths = am.sum(dim=1).to(torch.int32)
                hc = h.detach().to(compute_dtype()).contiguous()
                for layer in layers[:n_hip]:
                    hc = self._frozen_layer(layer, hc, lengths)
                h = hc.float()
            else:
                n_hip = 0
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            ext = (1.0 - am[:, None, None, :].to(h.dtype)) * torch.finfo(h.dtype).min
            for layer in layers[n_hip:]:
                out = layer(h, attention_mask=ext)
                h = out[0] if isinstance(out, tuple) else out
        return h[:, 0, :].float()

    @torch.no_grad()
    def _frozen_layer(self, layer, h, lengths):
        """One post-LN BERT layer, forward only, on (B, T, 768) in the compute dtype (reference:
        transformers BertLayer = BertAttention(BertSelfAttention + BertSelfOutput) + BertIntermediate
        (exact GELU) + BertOutput; dropout sites as in train mode).  Keys at positions >= length are
        masked (right-padded attention masks, as the tokenizer produces)."""
        from .. import ops

        att, dt = layer.attention, h.dtype
        C = h.shape[-1]
        sa = att.self
        H = sa.num_attention_heads
        tr = self.training
        p_att = float(sa.dropout.p) if tr else 0.0
        p_hid = float(att.output.dropout.p) if tr else 0.0
        qkv_w = [sa.query.weight, sa.key.weight, sa.value.weight]
        qkv = ops.conv1d(h, PF.packed_cat(qkv_w, dt), PF.bias_cat([sa.query.bias, sa.key.bias, sa.value.bias]), 3 * C)
        ctx, _ = ops.attention_fwd(qkv[:, :, :C], qkv[:, :, C : 2 * C], qkv[:, :, 2 * C :], None, None, None, lengths, H,
                                   "plain", drop_p=p_att, drop_seed=PF.next_seed() if p_att > 0 else 0)
        a = ops.conv1d(ctx, PF.packed(att.output.dense.weight, dt), att.output.dense.bias, C, res=h, drop_p=p_hid,
                       drop_seed=PF.next_seed() if p_hid > 0 else 0)
        ln = att.output.LayerNorm
        h1 = ops.layernorm_fwd(a, ln.weight, ln.bias, ln.eps)[0]
        inter = ops.conv1d(h1, PF.packed(layer.intermediate.dense.weight, dt), layer.intermediate.dense.bias,
                           layer.intermediate.dense.out_features, act="gelu")
        o = ops.conv1d(inter, PF.packed(layer.output.dense.weight, dt), layer.output.dense.bias, C, res=h1, drop_p=p_hid,
                       drop_seed=PF.next_seed() if p_hid > 0 else 0)
        ln2 = layer.output.LayerNorm
        return ops.layernorm_fwd(o, ln2.weight, ln2.bias, ln2.eps)[0]


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
