"""CPU: libptpp_hip.so loads and exports every symbol include/ptpp.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ptpp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptpp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from promptttspp_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "run `make -C promptttspp_amd/csrc` (or __graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ptpp.h but not exported"
    # and the Python binding table covers the header exactly
    assert sorted(_lib.SIGNATURES) == names


def test_binding_loads_and_reports_version():
    from promptttspp_amd import _lib

    lib = _lib.load()
    assert lib.ptpp_version() >= 1
    assert lib.ptpp_conv_cin_padded(80, _lib.BF16) == 96
    assert lib.ptpp_conv_cin_padded(256, _lib.BF16) == 256
    assert lib.ptpp_conv_cin_padded(32, _lib.F32) == 32


def test_bad_arguments_fail_loudly_without_gpu():
    import pytest
    import torch

    from promptttspp_amd import _lib, ops

    with pytest.raises(_lib.PtppError):
        ops.bct_to_btc(torch.zeros(1, 4, 4), torch.float32)  # CPU tensor: no fallback
    a = _lib.ConvArgs()
    assert _lib.load().ptpp_conv1d_fwd(ctypes.byref(a), None) == -1
    assert b"null" in _lib.load().ptpp_last_error()


def test_bert_wrapper_refuses_silent_random_init(monkeypatch):
    """Offline and without the opt-in, the prompt encoder must not come up with 11 frozen random BERT layers."""
    import pytest

    from promptttspp_amd.modules.prompt_encoder import BertWrapper, allow_random_bert

    monkeypatch.delenv("PTPP_ALLOW_RANDOM_BERT", raising=False)
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    with pytest.raises(RuntimeError, match="PTPP_ALLOW_RANDOM_BERT"):
        BertWrapper("bert-base-uncased")
    with allow_random_bert():
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert BertWrapper("bert-base-uncased").model.config.hidden_size == 768
    import os

    assert "PTPP_ALLOW_RANDOM_BERT" not in os.environ  # (the context manager is a module-level flag, not the environment)


def test_random_bert_must_be_overwritten_by_a_checkpoint(monkeypatch):
    """A BertWrapper that fell back to random weights says so until a state dict with its keys is loaded; the trainer's
    check (check_bert_loaded) refuses a partial / non-strict load that left it random."""
    import warnings

    import pytest
    import torch

    from promptttspp_amd.modules.prompt_encoder import BertWrapper, allow_random_bert, check_bert_loaded

    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    with allow_random_bert(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        holder = torch.nn.Module()
        holder.bert = BertWrapper("bert-base-uncased")
    assert holder.bert.random_init
    with pytest.raises(RuntimeError, match="randomly initialised"):
        check_bert_loaded(holder, "the test checkpoint")
    sd = holder.state_dict()
    partial = {k: v for k, v in sd.items() if "encoder.layer" not in k}       # a checkpoint without the encoder
    holder.load_state_dict(partial, strict=False)
    assert holder.bert.random_init
    holder.load_state_dict(sd)
    assert not holder.bert.random_init
    check_bert_loaded(holder)


def test_fused_adamw_state_dict_round_trips_with_torch_adamw():
    """Checkpoint interchange with the reference's torch.optim.AdamW (host logic only: no step is taken)."""
    import torch

    from promptttspp_amd.optim import FusedAdamW

    ps = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5))]
    ref = torch.optim.AdamW(ps, lr=1e-3)
    for p in ps:
        p.grad = torch.randn_like(p)
    for _ in range(3):
        ref.step()
    fused = FusedAdamW(ps, lr=1e-3)
    fused.load_state_dict(ref.state_dict())
    assert fused.param_groups[0]["step"] == 3                      # bias correction continues at step 3
    assert torch.equal(fused.state[ps[0]]["exp_avg"], ref.state[ps[0]]["exp_avg"])
    back = torch.optim.AdamW(ps, lr=1e-3)
    back.load_state_dict(fused.state_dict())                       # and the other direction
    assert float(back.state[ps[1]]["step"]) == 3.0
    back.step()                                                    # would raise KeyError('step') without the per-param entry


def test_ctypes_structures_match_the_header(tmp_path):
    """sizeof of every argument block: the C header compiled with gcc against the ctypes mirrors of _lib.py."""
    import shutil
    import subprocess

    from promptttspp_amd import _lib

    if shutil.which("gcc") is None:
        import pytest

        pytest.skip("no gcc")
    pairs = [("ptpp_conv1d_args", _lib.ConvArgs), ("ptpp_wgrad_problem", _lib.WgradProblem), ("ptpp_wgrad_gproblem", _lib.WgradGProblem), ("ptpp_diffnet_stack_fwd_args", _lib.DiffNetFwdArgs), ("ptpp_diffnet_layer_args", _lib.DiffNetLayerArgs),
             ("ptpp_diffnet_stack_bwd_args", _lib.DiffNetBwdArgs), ("ptpp_encoder_layers_fwd_args", _lib.EncoderLayersFwdArgs),
             ("ptpp_conv_ln_stack_fwd_args", _lib.ConvLnFwdArgs), ("ptpp_conv_ln_stack_bwd_args", _lib.ConvLnBwdArgs),
             ("ptpp_conformer_weights", _lib.ConformerWeights), ("ptpp_conformer_grads", _lib.ConformerGrads),
             ("ptpp_conformer_block_fwd_args", _lib.ConformerFwdArgs), ("ptpp_conformer_block_bwd_args", _lib.ConformerBwdArgs),
             ("ptpp_refenc_convs_fwd_args", _lib.RefEncConvsFwdArgs), ("ptpp_refenc_convs_bwd_args", _lib.RefEncConvsBwdArgs)]
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "%s"\nint main(void){%s return 0;}\n' % (
        os.path.join(ROOT, "include", "ptpp.h"), "".join('printf("%%zu\\n", sizeof(%s));' % n for n, _ in pairs)))
    subprocess.run(["gcc", str(src), "-o", str(tmp_path / "sz")], check=True)
    sizes = [int(v) for v in subprocess.run([str(tmp_path / "sz")], check=True, capture_output=True, text=True).stdout.split()]
    for (name, cls), size in zip(pairs, sizes):
        assert ctypes.sizeof(cls) == size, (name, ctypes.sizeof(cls), size)
