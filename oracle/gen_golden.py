"""TEST INFRASTRUCTURE ONLY -- golden-vector generator.

Runs ONLY in the build container, where the reference is importable from
/root/reference (it never travels to the GPU box).  For every hot-path module
 it builds the REFERENCE module from explicit kwargs, fills
it with synthetic weights (oracle/fill.py, reproducible from (seed, name)),
feeds seeded numpy inputs and stores inputs + reference outputs as small .npz
fixtures under tests/golden/.  No reference source or bytecode is copied.

    python oracle/gen_golden.py [name ...]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle.fill import fill_state_dict  # noqa: E402

GENERATORS = {}


def gen(fn):
    GENERATORS[fn.__name__] = fn
    return fn


def _ref():
    if REF not in sys.path:
        sys.path.insert(0, REF)


def _save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def _keys(module):
    return np.array([f"{k}|{','.join(map(str, v.shape))}" for k, v in module.state_dict().items()])


def rnd(seed, *shape, scale=1.0):
    return torch.from_numpy((scale * np.random.default_rng(seed).standard_normal(shape)).astype(np.float32))


# ----------------------------------------------------------------------------
@gen
def aa_snake():
    _ref()
    from promptttspp.layers.activations import AntiAliasActivation

    torch.manual_seed(0)
    C = 8
    m = AntiAliasActivation(C)
    fill_state_dict(m, seed=11)
    cases = {}
    for i, T in enumerate([1, 2, 5, 6, 7, 13, 64, 131]):
        x = rnd(100 + i, 2, C, T, scale=1.5)
        cases[f"x{T}"] = x
        cases[f"y{T}"] = m(x)
    _save("aa_snake", alpha=m.act.alpha.reshape(-1), f_up=m.up.filter.reshape(-1), f_dn=m.down.lowpass.filter.reshape(-1),
          keys=_keys(m), **cases)


@gen
def layer_norms():
    _ref()
    from promptttspp.layers.norm import LayerNorm as LNc
    from promptttspp.modules.esp.transformer.layer_norm import LayerNorm as LNesp
    from promptttspp.modules.frame_prior import LayerNorm as LNfp

    C = 256
    a, b, c = LNesp(C), LNc(C), LNfp(C)
    for i, m in enumerate((a, b, c)):
        fill_state_dict(m, seed=20 + i)
    x_btc = rnd(5, 3, 17, C, scale=2.0) + 0.5
    x_bct = rnd(6, 3, C, 17, scale=2.0) - 0.25
    _save(
        "layer_norms",
        x_btc=x_btc, x_bct=x_bct,
        esp_w=a.weight, esp_b=a.bias, esp_y=a(x_btc),
        c_gamma=b.gamma, c_beta=b.beta, c_y=b(x_bct),
        fp_gamma=c.gamma, fp_beta=c.beta, fp_y=c(x_bct),
        keys_esp=_keys(a), keys_c=_keys(b), keys_fp=_keys(c),
    )


@gen
def masks_paths():
    _ref()
    from promptttspp.utils.model import generate_path, sequence_mask, to_log_scale

    rng = np.random.default_rng(3)
    B, Tp = 3, 9
    plen = torch.tensor([9, 6, 1])
    dur = torch.from_numpy(rng.integers(1, 6, size=(B, Tp))).long()
    pmask = sequence_mask(plen, Tp)
    dur = dur * pmask
    flen = dur.sum(1)
    Tf = int(flen.max())
    fmask = sequence_mask(flen, Tf)
    path_mask = pmask[:, :, None].float() * fmask[:, None, :].float()
    path_int = generate_path(dur, path_mask.long())
    path_f = generate_path(dur.float(), path_mask)
    d2 = dur.float().clone()
    _save("masks_paths", plen=plen, dur=dur, flen=flen, pmask=pmask, fmask=fmask, path_int=path_int, path_f=path_f,
          log_dur=to_log_scale(d2))


BIGVGAN_KW = dict(in_channel=80, upsample_initial_channel=512, upsample_rates=[6, 5, 4, 2],
                  upsample_kernel_sizes=[12, 10, 8, 4], resblock_kernel_sizes=[3, 7, 11],
                  resblock_dilations=[[1, 3, 5], [1, 3, 5], [1, 3, 5]])


# weight-norm gains < 1 keep the synthetic vocoder's output off the tanh rails
VOC_GAIN = {"weight_g": 0.4}


def mel_like(seed, B, T):
    x = -5.5 + 2.1 * np.random.default_rng(seed).standard_normal((B, 80, T))
    return torch.from_numpy(np.clip(x, -11.5, 2.0).astype(np.float32))


@gen
def bigvgan():
    _ref()
    from promptttspp.vocoders import BigVGAN

    torch.manual_seed(0)
    m = BigVGAN(**BIGVGAN_KW).eval()
    fill_state_dict(m, seed=31, overrides=VOC_GAIN)
    x = mel_like(32, 2, 12)
    with torch.no_grad():
        y = m(x)
        amp = m.mrfs[3][1].layers[2]
        xa = rnd(33, 2, 32, 50)
        ya = amp(xa)
    _save("bigvgan", x=x, y=y, amp_x=xa, amp_y=ya, keys=_keys(m))


if __name__ == "__main__":
    names = sys.argv[1:] or list(GENERATORS)
    for n in names:
        GENERATORS[n]()
