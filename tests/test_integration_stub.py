"""INTEGRATION.md level 2: the ctypes stub printed there is EXECUTED as written (extracted from the file between
the stub markers) -- the CPU test checks that it binds every symbol it names and reads the attribute names of the
reference's modules (layers/activations.py:22-44, vocoders/bigvgan.py:21-47); the GPU test runs it on a module with
the reference's attribute tree and compares with the oracle."""
import os
import re
import types

import pytest
import torch
from conftest import ROOT, rel_err


def _stub():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- stub:begin -->\s*```python\n(.*?)```\s*<!-- stub:end -->", text, flags=re.S)
    assert m, "INTEGRATION.md lost its stub markers"
    from promptttspp_amd import _lib

    os.environ["PTPP_LIB"] = _lib.LIB_PATH
    mod = types.ModuleType("reference_side_ptpp_stub")
    exec(compile(m.group(1), "INTEGRATION.md:stub", "exec"), mod.__dict__)
    return mod, m.group(1)


def test_stub_binds_and_names_the_references_attributes():
    mod, src = _stub()
    for fn in ("aa_snake_channels_last", "conv1d_channels_last", "amp_layer_channels_last", "amp_layer_fused_channels_last"):
        assert callable(getattr(mod, fn))
    # the reference's own names (activations.py:22-44, bigvgan.py:24-40), not another code base's
    for name in ("act.up.filter", "act.down.lowpass.filter", "act.act.alpha", "layer.act1", "layer.conv2", "weight_g"):
        assert name in src
    for stale in ("Activation1d", "upsample.filter", "downsample.lowpass"):
        assert stale not in open(os.path.join(ROOT, "INTEGRATION.md")).read()
    # the struct mirror in the stub has the size of the library's own binding
    import ctypes

    from promptttspp_amd import _lib

    assert ctypes.sizeof(mod.ConvArgs) == ctypes.sizeof(_lib.ConvArgs) == 112
    assert ctypes.sizeof(mod.AmpLayerArgs) == ctypes.sizeof(_lib.AmpLayerArgs)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 3e-2), (torch.float16, 4e-3)])
def test_stub_amp_layer_matches_oracle(dtype, tol):
    from oracle import ref_torch as R
    from oracle.fill import fill_state_dict
    from promptttspp.vocoders.bigvgan import AMPLayer   # same attribute tree / state-dict keys as the reference's

    mod, _ = _stub()
    dev = torch.device("cuda:0")
    layer = AMPLayer(32, 7, 3)
    fill_state_dict(layer, seed=9, overrides={"weight_g": 0.4})
    sd = {"l." + k: v.clone() for k, v in layer.state_dict().items()}
    x = torch.randn(2, 32, 200, generator=torch.Generator().manual_seed(1))
    ref = R.amp_layer(sd, "l", x, 7, 3)                                     # (B, C, T)
    layer = layer.to(dev)
    y = mod.amp_layer_channels_last(x.transpose(1, 2).contiguous().to(dev).to(dtype), layer)
    torch.cuda.synchronize()
    assert rel_err(y.float().cpu().transpose(1, 2), ref) < tol
    yf = mod.amp_layer_fused_channels_last(x.transpose(1, 2).contiguous().to(dev).to(dtype), layer)   # the one-launch form
    assert rel_err(yf.float().cpu().transpose(1, 2), ref) < tol
