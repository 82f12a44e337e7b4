"""TEST INFRASTRUCTURE ONLY (oracle/): deterministic synthetic weights.

Golden fixtures cannot carry 70 M-parameter state dicts, so both the fixture
generator (which imports the reference, oracle/gen_golden.py) and the tests fill
a module's ``state_dict`` with the SAME values from a numpy PCG64 stream keyed
by (seed, parameter name).  Only names + shapes matter, so a product module with
the reference's state-dict contract gets bit-identical weights.
"""
import hashlib

import numpy as np
import torch

# buffers that are computed by the module itself and must not be overwritten
KEEP_SUFFIXES = (
    "filter",  # anti-alias FIR taps
    "num_batches_tracked",
    "betas",
    "alphas_cumprod",
    "alphas_cumprod_prev",
    "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod",
    "log_one_minus_alphas_cumprod",
    "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod",
    "posterior_variance",
    "posterior_log_variance_clipped",
    "posterior_mean_coef1",
    "posterior_mean_coef2",
    "position_ids",
    "token_type_ids",
)


def _rng(seed, name):
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return np.random.default_rng(int.from_bytes(h[:8], "little"))


def synth_tensor(name, shape, seed, gain=1.0):
    """Value for parameter/buffer ``name`` of ``shape`` (float32 numpy)."""
    r = _rng(seed, name)
    leaf = name.rsplit(".", 1)[-1]
    shape = tuple(shape)
    n = int(np.prod(shape)) if len(shape) else 1
    if leaf == "weight_g":
        v = gain * r.uniform(0.5, 1.5, size=n)
    elif leaf == "running_var":
        v = r.uniform(0.5, 1.5, size=n)
    elif leaf == "running_mean":
        v = 0.1 * r.standard_normal(n)
    elif leaf == "gamma" or (leaf == "weight" and len(shape) == 1):
        # LayerNorm / BatchNorm scale
        v = 1.0 + 0.1 * r.standard_normal(n)
    elif leaf in ("beta",) or "bias" in leaf:
        v = 0.1 * r.standard_normal(n)
    elif leaf == "alpha":
        v = 0.3 * r.standard_normal(n)
    elif leaf == "gst_embs":
        v = r.standard_normal(n)
    elif leaf in ("pos_bias_u", "pos_bias_v"):
        v = 0.1 * r.standard_normal(n)
    else:
        # dense / conv / embedding / recurrent weights: fan-in scaling
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        v = gain * r.standard_normal(n) / np.sqrt(max(fan_in, 1))
    return v.astype(np.float32).reshape(shape)


def fill_state_dict(module, seed, gain=1.0, overrides=None, offsets=None):
    """Overwrite every (non-kept) entry of ``module.state_dict()`` in place.
    ``overrides``: {name_suffix: gain} for individual tensors (e.g. to tame the
    duration head); ``offsets``: {name_suffix: constant added}.  Returns [(name, shape)] of what was filled."""
    filled = []
    sd = module.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            if name.endswith(KEEP_SUFFIXES) or not torch.is_floating_point(t):
                continue
            g = gain
            if overrides:
                for suf, og in overrides.items():
                    if name.endswith(suf):
                        g = og
            t.copy_(torch.from_numpy(synth_tensor(name, t.shape, seed, g)))
            if offsets:
                for suf, off in offsets.items():
                    if name.endswith(suf):
                        t.add_(off)
            filled.append((name, tuple(t.shape)))
    return filled


def synth_state_dict(names_shapes, seed, gain=1.0):
    """{name: tensor} for an explicit [(name, shape)] list (oracle-side use)."""
    return {n: torch.from_numpy(synth_tensor(n, s, seed, gain)) for n, s in names_shapes}
