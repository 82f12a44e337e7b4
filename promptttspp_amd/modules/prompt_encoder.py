"""Prompt encoder: BERT CLS state -> 3-layer MLP (reference:
promptttspp/modules/prompt_encoder.py:22-56).

The BERT encoder is the third-party HuggingFace ``BertModel`` in the reference
(un-vendored dependency): the module tree and state-dict keys are kept, the
arithmetic runs on this package's kernels -- embeddings, the 11 frozen layers
(forward only) and the trainable ``encoder.layer[-1].attention`` with autograd
through the attention-probability dropout (``ptpp_attention_fwd/bwd``, PLAIN
variant).  The transformers modules themselves are only called for CPU tensors.

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


_allow_random = [0]  # nesting depth of ``allow_random_bert`` (a module-level flag: nothing process-global is mutated)


class allow_random_bert:
    """``with allow_random_bert():`` -- constructors inside may fall back to a randomly initialised BERT (the caller
    loads a checkpoint that holds ``prompt_encoder.bert.model.*``, or asked for a random model explicitly).  A wrapper
    built that way carries ``random_init = True`` until a state dict with its keys is loaded (``check_bert_loaded``)."""

    def __enter__(self):
        _allow_random[0] += 1

    def __exit__(self, *exc):
        _allow_random[0] -= 1
        return False


def check_bert_loaded(model, what="the checkpoint"):
    """Raise if a BertWrapper under ``model`` still holds its random initialisation: a partial / non-strict load that did
    not supply ``prompt_encoder.bert.model.*`` would otherwise train on eleven frozen random layers without a word."""
    for name, m in model.named_modules():
        if isinstance(m, BertWrapper) and m.random_init:
            raise RuntimeError(f"{name}: BERT was randomly initialised (pretrained weights unavailable) and {what} did not "
                               "supply its weights (prompt_encoder.bert.model.*)")


class BertWrapper(nn.Module):
    def __init__(self, class_name="bert-base-uncased", allow_random_init=None):
        super().__init__()
        from transformers import BertConfig, BertModel, BertTokenizer

        def pretrained(cls):
            try:
                return cls.from_pretrained(class_name, local_files_only=True)  # the local cache, no network round trip
            except Exception:
                return cls.from_pretrained(class_name)  # the hub, like the reference (prompt_encoder.py:25-26)

        self.random_init = False  # True: weights are a random initialisation until a state dict supplies them
        try:
            self.model = pretrained(BertModel)
        except Exception as e:
            if allow_random_init is None:
                allow_random_init = _allow_random[0] > 0 or os.environ.get("PTPP_ALLOW_RANDOM_BERT", "") not in ("", "0")
            if not allow_random_init:
                raise RuntimeError(
                    f"BertModel.from_pretrained({class_name!r}) failed ({type(e).__name__}: {e}) and random "
                    "initialisation of the frozen BERT encoder was not allowed.  Provide the pretrained weights (HF "
                    "cache / network), or -- when a model checkpoint will supply prompt_encoder.bert.model.* -- set "
                    "PTPP_ALLOW_RANDOM_BERT=1.") from e
            warnings.warn(f"BertModel.from_pretrained({class_name!r}) unavailable ({type(e).__name__}); "
                          "building bert-base from BertConfig() -- weights must come from the model checkpoint")
            self.model = BertModel(BertConfig())
            self.random_init = True
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

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        if any(k.startswith(prefix + "model.encoder.layer.0.") for k in state_dict):
            self.random_init = False
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

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
        """CLS state of the last layer, (B, 768) f32.  On a ROCm device everything runs on the package's kernels:
        embeddings (gathers + fused LayerNorm/dropout), the frozen layers forward-only, the trainable layer
        (``encoder.layer[-1].attention``, modules/prompt_encoder.py:29-31) through autograd Functions with hand-written
        backward -- including the dropout on the attention probabilities, whose mask the backward regenerates."""
        ids, am = self.tokenize(prompts, device)
        if not (ids.is_cuda and self.hip_frozen_layers):
            return self._forward_library(ids, am)
        from .. import ops

        layers = list(self.model.encoder.layer)
        lengths = am.sum(dim=1).to(torch.int32)
        with torch.no_grad():  # the embeddings are frozen, and nothing upstream of them needs a gradient
            h = self._embeddings(ids)
            n_frozen = self._n_frozen(layers)
            if PF.STACK_DRIVERS and n_frozen > 0:
                h = self._frozen_layers_driver(layers[:n_frozen], h, lengths)
            else:
                for layer in layers[:n_frozen]:
                    h = self._frozen_layer(layer, h, lengths)
        for layer in layers[n_frozen:]:
            if layer is layers[-1]:
                return self._last_layer_cls(layer, h, lengths).float()
            h = self._full_layer(layer, h, lengths)
        return h[:, 0, :].float()

    def _forward_library(self, ids, am):
        """The same computation through the transformers modules (CPU tensors, or ``hip_frozen_layers = False``: the
        cross-check of the tests).  BertModel.forward, layer by layer: the library's mask helper inspects the mask on
        the HOST (mask.all()), a device sync per step; same modules, same state-dict keys."""
        amp = compute_dtype() == torch.bfloat16 and ids.is_cuda
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            h = self.model.embeddings(input_ids=ids)
            ext = (1.0 - am[:, None, None, :].to(h.dtype)) * torch.finfo(h.dtype).min
            for layer in self.model.encoder.layer:
                out = layer(h, attention_mask=ext)
                h = out[0] if isinstance(out, tuple) else out
        return h[:, 0, :].float()

    def _embeddings(self, ids):
        """BertEmbeddings (transformers): word + absolute position + token-type(0) embeddings -> LayerNorm -> dropout,
        (B, T, 768) in the compute dtype.  The three gathers are index lookups; LayerNorm + dropout is one kernel."""
        from .. import ops

        e = self.model.embeddings
        T = ids.shape[1]
        x = e.word_embeddings.weight[ids] + e.position_embeddings.weight[:T].unsqueeze(0) + e.token_type_embeddings.weight[0]
        p = float(e.dropout.p) if self.training else 0.0
        ln = e.LayerNorm
        return ops.layernorm_fwd(x.to(compute_dtype()).contiguous(), ln.weight, ln.bias, ln.eps,
                                 drop_out=(p, PF.next_seed() if p > 0 else 0))[0]

    def _attention_block(self, layer, h, lengths):
        """BertAttention with autograd: fused q|k|v projection, plain attention with probability dropout, output
        projection with hidden dropout + residual, LayerNorm."""
        att = layer.attention
        sa = att.self
        tr = self.training
        qkv = PF.linear_fused(h, [sa.query, sa.key, sa.value])
        ctx = PF.attention(qkv, None, None, None, lengths, sa.num_attention_heads, "plain",
                           drop_p=float(sa.dropout.p) if tr else 0.0)
        a = PF.conv1d(ctx, att.output.dense.weight, att.output.dense.bias, res=h,
                      drop_p=float(att.output.dropout.p) if tr else 0.0)
        ln = att.output.LayerNorm
        return PF.layer_norm(a, ln.weight, ln.bias, ln.eps)

    def _ffn_block(self, layer, h1):
        """BertIntermediate (exact GELU) + BertOutput on (B, T, 768) rows, differentiable w.r.t. its input."""
        z = PF.linear(h1, layer.intermediate.dense.weight, layer.intermediate.dense.bias)
        g = torch.nn.functional.gelu(z.float()).to(z.dtype)
        o = PF.conv1d(g, layer.output.dense.weight, layer.output.dense.bias, res=h1,
                      drop_p=float(layer.output.dropout.p) if self.training else 0.0)
        ln2 = layer.output.LayerNorm
        return PF.layer_norm(o, ln2.weight, ln2.bias, ln2.eps)

    def _full_layer(self, layer, h, lengths):
        return self._ffn_block(layer, self._attention_block(layer, h, lengths))

    def _last_layer_cls(self, layer, h, lengths):
        """The last layer.  Only its CLS row leaves the encoder (prompt_encoder.py:37), and the feed-forward half of a
        BERT layer acts on each token separately: it runs on the B CLS rows only (the same numbers and the same
        gradients as the full layer restricted to what is used, a 1/T of the work)."""
        h1 = self._attention_block(layer, h, lengths)
        return self._ffn_block(layer, h1[:, 0:1, :].contiguous())[:, 0, :]

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


    @torch.no_grad()
    def _frozen_layers_driver(self, layers, h, lengths):
        """All frozen layers in ONE C call (ptpp_encoder_layers_fwd): the launches of ``_frozen_layer`` in the same order
        with the same dropout seeds (bit-identical to the loop), without ~70 Python -> C round trips per step."""
        import ctypes

        from .. import _lib, ops

        dt, dev = h.dtype, h.device
        B, T, C = h.shape
        sa0 = layers[0].attention.self
        Fi = layers[0].intermediate.dense.out_features
        tr = self.training
        p_att = float(sa0.dropout.p) if tr else 0.0
        p_hid = float(layers[0].attention.output.dropout.p) if tr else 0.0
        # the pointer tables of the frozen layers change only when their parameters do: cached on the parameters' versions
        # and storage (132 packed-operand lookups per step otherwise, ~0.8 ms of host time)
        srcs = getattr(self, "_frozen_srcs", None)
        # identity of the layer list AND of a first / last parameter OBJECT: load_state_dict(assign=True), a parametrization or
        # `module.weight = nn.Parameter(..)` replaces Parameters without touching the layer modules
        ident = (len(layers), id(layers[0]), id(layers[-1]), id(layers[0].attention.self.query.weight), id(layers[-1].output.LayerNorm.bias))
        if srcs is None or srcs[0] != ident:
            # (the parameter OBJECTS of the frozen layers do not change between steps: walking 11 layers x 16 parameters through
            #  named_parameters() every forward was ~0.2 ms of host time per step)
            srcs = self._frozen_srcs = (ident, [t for layer in layers for t in layer.parameters()])
            self._frozen_tables = None
        srcs = srcs[1]
        stamp = (dt, len(layers), PF._pack_gen[0], tuple(t._version for t in srcs), srcs[0].data_ptr())
        cached = getattr(self, "_frozen_tables", None)
        if cached is not None and cached[0] == stamp:
            cols = cached[1]
        else:
            cols = [[] for _ in range(12)]
            for layer in layers:
                att, sa = layer.attention, layer.attention.self
                assert sa.num_attention_heads == sa0.num_attention_heads and layer.intermediate.dense.out_features == Fi
                row = (PF.packed_cat([sa.query.weight, sa.key.weight, sa.value.weight], dt),
                       PF.bias_cat([sa.query.bias, sa.key.bias, sa.value.bias]),
                       PF.packed(att.output.dense.weight, dt), att.output.dense.bias, att.output.LayerNorm.weight, att.output.LayerNorm.bias,
                       PF.packed(layer.intermediate.dense.weight, dt), layer.intermediate.dense.bias,
                       PF.packed(layer.output.dense.weight, dt), layer.output.dense.bias, layer.output.LayerNorm.weight, layer.output.LayerNorm.bias)
                for c, t in zip(cols, row):
                    assert t.is_contiguous() and (t.dtype == torch.float32 or t.dtype == dt)
                    c.append(t)
            stamp = (dt, len(layers), PF._pack_gen[0], tuple(t._version for t in srcs), srcs[0].data_ptr())  # (packing may add entries)
            self._frozen_tables = (stamp, cols)
        seeds = []
        for _ in layers:
            seeds += [PF.next_seed() if p_att > 0 else 0, PF.next_seed() if p_hid > 0 else 0, PF.next_seed() if p_hid > 0 else 0]
        h = h.contiguous()
        out = torch.empty_like(h)
        scratch = torch.empty(B * T * (7 * C + Fi), device=dev, dtype=dt)
        ws = ops.workspace(dev)
        lens = ops.i32(lengths, dev)
        a = _lib.EncoderLayersFwdArgs()
        a.h_in, a.h_out, a.lengths = h.data_ptr(), out.data_ptr(), lens.data_ptr()
        tabs = [PF._ptr_table(c) for c in cols]
        (a.qkv_wp, a.qkv_b, a.ao_wp, a.ao_b, a.ln1_g, a.ln1_b, a.i_wp, a.i_b, a.o_wp, a.o_b, a.ln2_g, a.ln2_b) = \
            [ctypes.cast(t, ctypes.c_void_p) for t in tabs]
        sd = (ctypes.c_uint64 * len(seeds))(*seeds)
        a.seeds = ctypes.cast(sd, ctypes.c_void_p)
        a.scratch, a.scratch_bytes = scratch.data_ptr(), scratch.numel() * scratch.element_size()
        if not torch.cuda.is_current_stream_capturing():  # (as ops.conv1d: no split-K scratch inside a graph capture)
            a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
        a.eps, a.p_att, a.p_hid = float(layers[0].attention.output.LayerNorm.eps), p_att, p_hid
        a.B, a.T, a.C, a.F, a.H, a.L, a.dtype = B, T, C, Fi, sa0.num_attention_heads, len(layers), ops.dtype_code(dt)
        _lib.check(_lib.load().ptpp_encoder_layers_fwd(ctypes.byref(a), ops._stream()), "ptpp_encoder_layers_fwd")
        return out


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
