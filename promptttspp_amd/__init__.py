"""promptttspp_amd -- MI355X-native (gfx950) implementation of the PromptTTS++
mel-synthesis hot path (prompttts_mdn_v2 training/inference step + BigVGAN).

The package mirrors the reference's ``promptttspp`` module tree (same class
names, constructor kwargs, method signatures and state-dict keys) so Hydra
``_target_`` paths keep resolving through the thin ``promptttspp`` alias package
at the repo root; the arithmetic runs in hand-written HIP kernels behind the
C ABI of ``include/ptpp.h`` (``promptttspp_amd/csrc``).
"""
__version__ = "0.1.0"
