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
    assert dtype in (torch.float32, torch.bfloat16), "compute dtype must be float32 or bfloat16"
    _state["dtype"] = dtype


@contextlib.contextmanager
def use_dtype(dtype):
    old = _state["dtype"]
    set_compute_dtype(dtype)
    try:
        yield
    finally:
        _state["dtype"] = old
