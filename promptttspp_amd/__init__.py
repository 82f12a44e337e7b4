"""promptttspp_amd -- MI355X-native (gfx950) implementation of the PromptTTS++
mel-synthesis hot path (prompttts_mdn_v2 training/inference step + BigVGAN).

The package mirrors the reference's ``promptttspp`` module tree (same class
names, constructor kwargs, method signatures and state-dict keys) so Hydra
``_target_`` paths keep resolving through the thin ``promptttspp`` alias package
at the repo root; the arithmetic runs in hand-written HIP kernels behind the
C ABI of ``include/ptpp.h`` (``promptttspp_amd/csrc``).
"""
import os

# The few modules still on PyTorch-ROCm library ops (reference-encoder Conv2d stack,
# Conformer depthwise conv) see a new (batch, length) shape every step
# (token-bucket batching).  MIOpen's default find mode benchmarks every solver --
# including its naive reference kernels -- per new shape (measured: 7 s/step); FAST
# mode picks a solver from its heuristics instead.
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

# Kernel arguments in device memory instead of host-coherent memory: the step issues ~1000 short dependent launches on four
# streams and each one's argument block is fetched when the dispatch starts -- 15.5 -> 14.7 ms per training step on the same
# box.  Read by the HIP runtime when it initialises (first device call of the process); an
# entry point that imports torch first sets it itself before that import (bench.py, __graft_entry__.py, train.py, app.py).
# PTPP_NO_DEV_KERNARG=1 opts out (the variable is process-wide: it also affects other HIP users of the process).
if not os.environ.get("PTPP_NO_DEV_KERNARG"):
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    import sys as _sys

    _t = _sys.modules.get("torch")
    if _t is not None and getattr(_t, "cuda", None) is not None and _t.cuda.is_initialized() and os.environ.get("HIP_FORCE_DEV_KERNARG") == "1":
        import warnings as _w

        _w.warn("promptttspp_amd: the HIP runtime was initialised before this import, HIP_FORCE_DEV_KERNARG=1 cannot take effect any "
                "more (~0.8 ms per training step); set it in the environment or import promptttspp_amd before touching the GPU",
                stacklevel=2)

__version__ = "0.1.0"
