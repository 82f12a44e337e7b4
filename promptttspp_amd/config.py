"""Process-wide compute dtype of the HIP path (torch.bfloat16 = bf16 storage +
bf16 MFMA with f32 accumulate, the benchmark configuration; torch.float32 =
exact-f32 MFMA, the parity configuration).  MDN heads/losses, LayerNorm and
softmax statistics and all parameter gradients are f32 in both modes."""
import contextlib

import torch

_state = {"dtype": torch.bfloat16}


def compute_dtype():
    return _state["dtype"]


def set_compute_dtype(dtype):
    # float16 (round 6): INFERENCE of the acoustic model's decoder only -- the 100-step sampler (one-launch DiffNet layers,
    # sampler head, conditioner GEMM) with the conditioning path in f32 (model.f32_conditioning); BASELINE config 5's wording
    assert dtype in (torch.float32, torch.bfloat16, torch.float16), "compute dtype must be float32, bfloat16 or float16"
    _state["dtype"] = dtype


@contextlib.contextmanager
def use_dtype(dtype):
    old = _state["dtype"]
    set_compute_dtype(dtype)
    try:
        yield
    finally:
        _state["dtype"] = old
