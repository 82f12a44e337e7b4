"""GPU: BigVGAN product path (HIP kernels through the C ABI) against the golden
vectors captured from the reference and against the oracle."""
import pytest
import torch
import torch.nn.functional as F
from conftest import key_shapes, load_golden, rel_err
from test_oracle_golden import VOC_GAIN, vocoder_sd

from oracle import ref_torch as R

pytestmark = pytest.mark.gpu

BIGVGAN_KW = dict(in_channel=80, upsample_initial_channel=512, upsample_rates=[6, 5, 4, 2],
                  upsample_kernel_sizes=[12, 10, 8, 4], resblock_kernel_sizes=[3, 7, 11],
                  resblock_dilations=[[1, 3, 5], [1, 3, 5], [1, 3, 5]])


def build(dev, seed=31):
    from promptttspp_amd.vocoders import BigVGAN

    g = load_golden("bigvgan")
    m = BigVGAN(**BIGVGAN_KW)
    ref_keys = key_shapes(g["keys"])
    mine = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert mine == ref_keys, "state-dict contract (names, shapes, order) differs from the reference"
    sd = vocoder_sd(ref_keys, seed)
    m.load_state_dict(sd)  # strict: reference-format checkpoints load unchanged
    return m.to(dev).eval(), sd, g


def test_conv_transpose_rewrite_cpu_free(dev):
    """ConvTranspose1d == 3-tap conv with stride*Cout outputs (pure index algebra,
    checked here against torch on CPU tensors)."""
    from promptttspp_amd.vocoders.bigvgan import conv_transpose_as_conv

    for u, k in [(6, 12), (5, 10), (4, 8), (2, 4)]:
        w = torch.randn(8, 4, k)
        x = torch.randn(2, 8, 9)
        ref = F.conv_transpose1d(x, w, stride=u, padding=u // 2 + u % 2, output_padding=u % 2)
        wc, ks, pad = conv_transpose_as_conv(w, u, u // 2 + u % 2, u % 2)
        y = F.conv1d(x, wc, padding=pad)  # (B, u*Cout, T)
        y = y.view(2, u, 4, 9).permute(0, 2, 3, 1).reshape(2, 4, 9 * u)
        assert torch.allclose(y, ref, atol=1e-5)


def test_bigvgan_f32_matches_reference_golden(dev):
    m, sd, g = build(dev)
    m.set_compute_dtype(torch.float32)
    y = m(g["x"].to(dev)).cpu()
    assert y.shape == g["y"].shape
    assert rel_err(y, g["y"]) < 1e-3  # north_star tolerance: 1e-3 relative fp32
    assert rel_err(y, g["y"]) < 1e-4  # what we actually hold


def test_bigvgan_f32_ragged_length_matches_oracle(dev):
    m, sd, g = build(dev)
    m.set_compute_dtype(torch.float32)
    x = torch.clamp(-5.5 + 2.1 * torch.randn(3, 80, 29, generator=torch.Generator().manual_seed(5)), -11.5, 2.0)
    ref = R.bigvgan(sd, x)
    y = m(x.to(dev)).cpu()
    assert rel_err(y, ref) < 1e-4


def test_bigvgan_bf16_close(dev):
    m, sd, g = build(dev)
    m.set_compute_dtype(torch.bfloat16)
    y = m(g["x"].to(dev)).cpu()
    # bf16 storage through 72 convs: waveform-level agreement, not 1e-3
    assert rel_err(y, g["y"]) < 1.3e-2  # measured 8.2e-3 (round 5: x 1.5); the f16 mode holds 1.0e-3 (test_bigvgan_f16_close)
    mse = float(((y - g["y"]) ** 2).mean() / (g["y"] ** 2).mean())
    assert mse < 1e-3


def test_bigvgan_f16_close(dev):
    """PTPP_F16 (the reference's AMP dtype, BASELINE configs 4 / 5): half storage, f32 accumulation -- 3 more mantissa bits than
    bf16 through the same 72 convs."""
    m, sd, g = build(dev)
    m.set_compute_dtype(torch.float16)
    y = m(g["x"].to(dev)).cpu()
    assert torch.isfinite(y).all()
    e16 = rel_err(y, g["y"])
    m.set_compute_dtype(torch.bfloat16)
    eb = rel_err(m(g["x"].to(dev)).cpu(), g["y"])
    print(f"BigVGAN waveform max-norm error vs the reference golden: f16 {e16:.2e}, bf16 {eb:.2e}")
    assert e16 < 1.5e-2 and e16 < eb
    assert float(((y - g["y"]) ** 2).mean() / (g["y"] ** 2).mean()) < 1e-5


def test_weight_cache_follows_state_dict(dev):
    m, sd, g = build(dev)
    m.set_compute_dtype(torch.float32)
    y1 = m(g["x"].to(dev)).cpu()
    sd2 = vocoder_sd(key_shapes(g["keys"]), seed=77)
    m.load_state_dict(sd2)
    y2 = m(g["x"].to(dev)).cpu()
    assert rel_err(y2, R.bigvgan(sd2, g["x"])) < 1e-4
    assert rel_err(y1, y2) > 1e-2


def test_f0_aware_bigvgan_matches_reference_golden(dev):
    """F0AwareBigVGAN with the reference's RNG draws injected (rand -> initial phases,
    randn_like -> additive noise, randn_like -> the unused noise branch)."""
    from promptttspp_amd.vocoders import F0AwareBigVGAN

    g = load_golden("bigvgan_f0")
    m = F0AwareBigVGAN(sampling_rate=24000, harmonic_num=8, **BIGVGAN_KW)
    keys = key_shapes(g["keys"])
    assert sorted((k, tuple(v.shape)) for k, v in m.state_dict().items()) == sorted(keys)
    sd = vocoder_sd(keys, 120)
    m.load_state_dict(sd)
    m = m.to(dev).eval().set_compute_dtype(torch.float32)
    B, L = g["nz"].shape[0], g["nz"].shape[1]
    q_rand, q_like = [g["rand_ini"].clone().to(dev)], [g["nz"].to(dev), torch.zeros(B, L, 1, device=dev)]
    o_rand, o_like = torch.rand, torch.randn_like
    torch.rand = lambda *a, **k: q_rand.pop(0)
    torch.randn_like = lambda *a, **k: q_like.pop(0)
    try:
        y = m(g["x"].to(dev), g["f0"].to(dev)).cpu()
    finally:
        torch.rand, torch.randn_like = o_rand, o_like
    assert y.shape == g["y"].shape
    assert rel_err(y, g["y"]) < 1e-3


@pytest.mark.parametrize("B,L,voiced", [(3, 24000, 0.7), (2, 141600, 0.6), (5, 999, 1.0), (1, 1500, 0.0)])
def test_fused_harmonic_source_against_the_tensor_op_chain(dev, B, L, voiced):
    """ptpp_nsf_source (csrc/nsf.hip; reference vocoders/nsf.py:31-206) against the tensor-op chain of SourceModuleHnNSF with the same
    three RNG draws: config-5 length (141 600 samples = 590 frames x 240), a length below the block's thread count, all-voiced and
    all-unvoiced tracks.  The two prefix sums run in another order than torch.cumsum; a wrap of the first one detected a sample early
    or late moves the phase by a whole period, which the sine does not see: the merged source agrees within 2e-3 absolute (its range is
    (-1, 1); measured ~2e-4 at full length, dominated by the O(L) rounding walk of the second sum in both implementations) and the
    launch is bit-reproducible."""
    from promptttspp_amd.vocoders import nsf

    torch.manual_seed(11)
    m = nsf.SourceModuleHnNSF(24000, harmonic_num=8).to(dev).eval()
    g = torch.Generator().manual_seed(5)
    t = torch.arange(L) / 24000.0
    f0 = 110.0 + 60.0 * torch.sin(2 * 3.14159 * 0.7 * t)[None, :] * torch.rand(B, 1, generator=g) + 40.0 * torch.rand(B, 1, generator=g)
    seg = (torch.rand(B, (L + 2399) // 2400, generator=g) < voiced).repeat_interleave(2400, dim=1)[:, :L]
    f0 = (f0 * seg).unsqueeze(-1).to(dev)

    rand_ini = torch.rand(B, 9, generator=g).to(dev)
    nz, nz2 = torch.randn(B, L, 9, generator=g).to(dev), torch.randn(B, L, 1, generator=g).to(dev)

    def run(fused):  # (the draws are injected: the tensor-op chain draws its noise for a TRANSPOSED tensor, i.e. in another element order)
        old, o_rand, o_like = nsf.FUSED_SOURCE, torch.rand, torch.randn_like
        q_rand, q_like = [rand_ini.clone()], [nz, nz2]
        nsf.FUSED_SOURCE = fused
        torch.rand = lambda *a, **k: q_rand.pop(0)
        torch.randn_like = lambda *a, **k: q_like.pop(0)
        try:
            with torch.no_grad():
                out = m(f0)
            assert not q_rand and not q_like  # all three draws were taken, in the reference's order
            return out
        finally:
            nsf.FUSED_SOURCE, torch.rand, torch.randn_like = old, o_rand, o_like

    ref, n_ref, uv_ref = run(False)
    got, n_got, uv_got = run(True)
    again, _, _ = run(True)
    assert got.shape == ref.shape == (B, L, 1)
    assert torch.equal(got, again)
    assert torch.equal(uv_ref, uv_got) and torch.equal(n_ref, n_got)  # same draws, same order
    err = float((got - ref).abs().max())
    assert err < 2e-3, err
    assert float(ref.abs().max()) > 0.05


def test_bigvgan_bench_size_batch_independence(dev):
    """BASELINE config 4 size (64 x 1000 frames, bf16): tiles never span utterances, so every utterance of
    the batch must come out BIT-IDENTICAL to the same utterance synthesised alone (streams / tile order /
    batch index must not change the arithmetic), finite and inside (-1, 1)."""
    from oracle.fill import fill_state_dict
    from promptttspp_amd.vocoders import BigVGAN

    torch.manual_seed(3)
    m = BigVGAN(80, 512, [6, 5, 4, 2], [12, 10, 8, 4], [3, 7, 11], [[1, 3, 5]] * 3)
    fill_state_dict(m, seed=5, overrides={"weight_g": 0.4})
    m = m.to(dev).eval().set_compute_dtype(torch.bfloat16)
    x = torch.clamp(-5.5 + 2.1 * torch.randn(64, 80, 1000, device=dev), -11.5, 2.0)
    y = m(x)
    assert y.shape == (64, 1, 240000) and torch.isfinite(y).all() and float(y.abs().max()) <= 1.0
    for b in (0, 37, 63):
        assert torch.equal(m(x[b : b + 1])[0], y[b]), b
