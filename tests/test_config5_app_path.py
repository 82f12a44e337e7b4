"""BASELINE.json config 5 at FULL size on the GPU: prompt -> waveform for 32 prompts, phoneme sequences
Tp ~ U{40..120}, ``app.synthesize_batch`` (infer_batch with use_max, 100-step sampler, zero-phase low-pass
on log-F0, F0-aware BigVGAN), bf16 decoder / vocoder with the f32 MDN island -- plus the bf16 end-to-end
bounds against the reference's f32 golden vectors (reference: app.py:49-82, model.py:261-325)."""
import os
import sys

import numpy as np
import pytest
import torch
from conftest import ROOT, load_golden, rel_err

pytestmark = pytest.mark.gpu
sys.path.insert(0, ROOT)


def _vocoder(dev, dtype):
    from oracle.fill import fill_state_dict
    from promptttspp_amd.hydra_lite import compose, instantiate

    voc = instantiate(compose(os.path.join(ROOT, "egs", "proposed", "bin", "conf"), "demo", []).vocoder)
    fill_state_dict(voc, seed=5, overrides={"weight_g": 0.4})
    return voc.to(dev).eval().set_compute_dtype(dtype)


def _inputs(B, seed=0):
    r = np.random.default_rng(seed)
    tp = r.integers(40, 121, size=B)
    phon = [torch.from_numpy(np.concatenate([[1], r.integers(3, 90, size=int(n) - 2), [2]])) for n in tp]
    L = r.integers(12, 49, size=B)
    ids = torch.zeros(B, int(L.max()), dtype=torch.long)
    am = torch.zeros_like(ids)
    for b in range(B):
        ids[b, 0], ids[b, L[b] - 1] = 101, 102
        ids[b, 1 : L[b] - 1] = torch.from_numpy(r.integers(1000, 30000, size=int(L[b]) - 2))
        am[b, : L[b]] = 1
    return phon, ids, am


def test_config5_32_prompts_full_size(dev):
    import app
    import test_hip_acoustic as T
    from promptttspp_amd import config

    model, _ = T._model(dev)  # synthetic weights with the tamed duration head of the parity fixtures
    model.eval()
    phon, ids, am = _inputs(32)
    prompts = (ids.to(dev), am.to(dev))
    try:
        with config.use_dtype(torch.bfloat16):
            voc = _vocoder(dev, torch.bfloat16)
            runs = []
            for _ in range(2):
                torch.manual_seed(11)
                runs.append(app.synthesize_batch(model, voc, phon, style_prompts=prompts, noise_scale=0.5))
            (wavs, mels), (wavs2, mels2) = runs
            dur_batch = model.last_durations.cpu()
            assert len(wavs) == 32
            total = 0
            for i, (w, m) in enumerate(zip(wavs, mels)):
                assert m.shape[0] == 80 and w.shape[0] == 240 * m.shape[1] and m.shape[1] >= len(phon[i])
                assert torch.isfinite(w).all() and torch.isfinite(m).all() and float(w.abs().max()) <= 1.0
                # same seed, same batch: bit-identical per utterance (deterministic kernels end to end)
                assert torch.equal(w, wavs2[i]) and torch.equal(m, mels2[i])
                # integer frame length == sum of the integer durations of this utterance's phones
                assert int(dur_batch[i, : len(phon[i])].sum()) == m.shape[1] and int(dur_batch[i, len(phon[i]):].sum()) == 0
                total += m.shape[1]
            assert total > 32 * 40
        # the integer side of the batch does not depend on the batch: every utterance run on its own (f32 compute so
        # that the comparison is about batching, not about rounding) gives the same integer durations as in the batch
        # -- except at its last two phones: the reference adds the style embedding to PADDED phones too
        # (model.py:111), so in a padded batch the 2-layer k=3 duration predictor sees a non-zero right neighbour
        # where the single-utterance call sees the conv's zero padding (reference behaviour, kept)
        with config.use_dtype(torch.float32):
            ph = torch.zeros(32, max(len(p) for p in phon), dtype=torch.long)
            for i, p in enumerate(phon):
                ph[i, : len(p)] = p
            plen = torch.tensor([len(p) for p in phon])
            torch.manual_seed(3)
            _, flen_b = model.infer_batch(ph[:8].to(dev), plen[:8].to(dev), style_prompt=(prompts[0][:8], prompts[1][:8]),
                                          use_max=True, noise_scale=0.0, noise_fn=lambda i, shape: torch.zeros(shape, device=dev))
            dur_b = model.last_durations.cpu()
            for i in range(8):
                n = len(phon[i])
                mel1 = model.infer(phon[i][None].to(dev), style_prompt=(prompts[0][i : i + 1, : int(am[i].sum())],
                                                                          prompts[1][i : i + 1, : int(am[i].sum())]),
                                   use_max=True, noise_scale=0.0, noise_fn=lambda i, shape: torch.zeros(shape, device=dev))
                assert torch.equal(model.last_durations.cpu()[0, : n - 2], dur_b[i, : n - 2]), i   # integer: bit exact
                assert mel1.shape[-1] == int(model.last_durations.sum())
                if n == ph[:8].shape[1]:  # the longest utterance of the batch has no padding: identical throughout
                    assert torch.equal(model.last_durations.cpu()[0], dur_b[i, :n]) and mel1.shape[-1] == int(flen_b[i])
    finally:
        config.set_compute_dtype(torch.float32)


def test_bf16_infer_batch_against_reference_golden(dev):
    """bf16 compute vs the reference's f32 outputs: free-running integer durations and frame lengths are bit-exact (integer
    island); with the same frame grid the bf16 mel is compared element by element with the golden.  Measured on MI355X (profiles/r02_bf16_module_errors.txt): mel MSE 4.9e-3
    after the 100 bf16 denoiser evaluations of the sampler (the f32 mode holds < 1e-6, and north_star's 1e-3 is the
    f32 bar); the bound below is 2x the measurement.  The bf16 durations themselves move by at most one frame on a
    small fraction of the phones, 2.8 % measured (the predictor convs run in reduced precision in the reference's AMP
    mode too, variance_adaptor.py:84-95; only the MDN head is f32)."""
    import test_hip_acoustic as T
    from promptttspp_amd import config

    gi = load_golden("model_infer")
    m, _ = T._model(dev)
    m.eval()
    B = gi["phon"].shape[0]
    Tf = int(gi["new_flen_ref"].max())

    def noise_fn(i, shape):
        t = T.rnd(112, B, 80, Tf) if i < 0 else T.rnd(2000 + i, B, 80, Tf)
        return t.transpose(1, 2).contiguous().to(dev)

    dur_ref = gi["new_dur_ref"].squeeze(1)
    try:
        with config.use_dtype(torch.bfloat16):
            kw = dict(reference_mel=gi["mel"].to(dev), ref_lengths=gi["flen_in"], return_f0=True)
            # (1) free-running durations in the bf16 mode vs the reference's: the integer island (style embedding, phoneme
            # encoder, duration predictor in f32 at inference, model.py `integer_island`) makes them bit-exact (north_star:
            # "integer durations / alignments bit-exact"); round 2 ran these layers in bf16 and moved 2.8 % of the phones
            # by one frame
            _, _, _, flen_free = m.infer_batch(gi["phon"].to(dev), gi["plen"].to(dev), noise_fn=lambda i, s: torch.zeros(s, device=dev), **kw)
            assert torch.equal(m.last_durations.cpu(), dur_ref)
            assert torch.equal(flen_free.cpu().float(), gi["new_flen_ref"].float())
            m.integer_island = False
            try:
                m.infer_batch(gi["phon"].to(dev), gi["plen"].to(dev), noise_fn=lambda i, s: torch.zeros(s, device=dev), **kw)
            finally:
                m.integer_island = True
            d = (m.last_durations.cpu() - dur_ref).abs()
            print("durations with the phone-level layers in bf16 (island off): max |diff|", int(d.max()), "fraction moved",
                  float((d > 0).float().mean()))
            assert int(d.max()) <= 1 and float((d > 0).float().mean()) <= 0.15
            # (2) reference durations imposed -> same frame grid -> mel comparable element by element
            dp = m.variance_adaptor.duration_predictor
            orig = dp.infer_cl
            dp.infer_cl = lambda x, plen: torch.log(dur_ref.clamp_min(1).float()).to(dev)
            try:
                mel, cf0, vuv, flen = m.infer_batch(gi["phon"].to(dev), gi["plen"].to(dev), noise_fn=noise_fn, **kw)
            finally:
                dp.infer_cl = orig
            assert torch.equal(m.last_durations.cpu(), dur_ref)
            assert torch.equal(flen.cpu().float(), gi["new_flen_ref"].float())
            mse = float(((mel.cpu() - gi["new_mel_ref"]) ** 2).mean())
            print("bf16 mel MSE vs reference", mse, "cf0 rel err", rel_err(cf0.cpu(), gi["new_cf0_ref"]))
            # north_star's bar for the mel (MSE < 1e-3 against the reference) holds in the bf16 mode too: the conditioning path
            # runs in f32 (model.py `f32_conditioning`), the 100 denoiser evaluations in bf16
            assert mse < 1e-3, mse
            assert rel_err(cf0.cpu(), gi["new_cf0_ref"]) < 1e-4
            assert torch.equal(vuv.cpu() > 0.5, gi["new_vuv_ref"] > 0.5) if "new_vuv_ref" in gi else True
            # the conditioning path in bf16 as well (round 2's mode), for the record: mel MSE 4.9e-3, log-F0 4.1e-2
            m.f32_conditioning = False
            dp.infer_cl = lambda x, plen: torch.log(dur_ref.clamp_min(1).float()).to(dev)
            try:
                mel2, cf02, _, _ = m.infer_batch(gi["phon"].to(dev), gi["plen"].to(dev), noise_fn=noise_fn, **kw)
            finally:
                dp.infer_cl = orig
                m.f32_conditioning = True
            mse2 = float(((mel2.cpu() - gi["new_mel_ref"]) ** 2).mean())
            print("  conditioning path in bf16: mel MSE", mse2, "cf0 rel err", rel_err(cf02.cpu(), gi["new_cf0_ref"]))
            assert mse2 < 1e-2 and rel_err(cf02.cpu(), gi["new_cf0_ref"]) < 8e-2
    finally:
        config.set_compute_dtype(torch.float32)


def test_f16_infer_batch_against_reference_golden(dev):
    """BASELINE config 5 as worded ("fp16 mel decoder + fp32 MDN head"; the reference's reduced-precision mode is fp16 autocast,
    trainers/tts.py:92,203-211): compute dtype float16 = the 100-step sampler on IEEE-half storage (one-launch DiffNet layers,
    sampler head, conditioner GEMM: csrc/diffnet_layer.hip / sampler_head.hip f16 instantiations) with the text -> conditioning
    path in f32.  Same checks as the bf16 twin above: free-running integer durations bit-exact, mel MSE against the reference's
    f32 golden < 1e-3 on the reference's frame grid (f16 keeps 3 more mantissa bits than bf16: the MSE is lower)."""
    import test_hip_acoustic as T
    from promptttspp_amd import config

    gi = load_golden("model_infer")
    m, _ = T._model(dev)
    m.eval()
    B = gi["phon"].shape[0]
    Tf = int(gi["new_flen_ref"].max())

    def noise_fn(i, shape):
        t = T.rnd(112, B, 80, Tf) if i < 0 else T.rnd(2000 + i, B, 80, Tf)
        return t.transpose(1, 2).contiguous().to(dev)

    dur_ref = gi["new_dur_ref"].squeeze(1)
    kw = dict(reference_mel=gi["mel"].to(dev), ref_lengths=gi["flen_in"], return_f0=True)
    res = {}
    try:
        for dt in (torch.float16, torch.bfloat16):
            with config.use_dtype(dt):
                if dt == torch.float16:
                    _, _, _, flen_free = m.infer_batch(gi["phon"].to(dev), gi["plen"].to(dev), noise_fn=lambda i, s: torch.zeros(s, device=dev), **kw)
                    assert torch.equal(m.last_durations.cpu(), dur_ref) and torch.equal(flen_free.cpu().float(), gi["new_flen_ref"].float())
                dp = m.variance_adaptor.duration_predictor
                orig = dp.infer_cl
                dp.infer_cl = lambda x, plen: torch.log(dur_ref.clamp_min(1).float()).to(dev)
                try:
                    mel, cf0, vuv, flen = m.infer_batch(gi["phon"].to(dev), gi["plen"].to(dev), noise_fn=noise_fn, **kw)
                finally:
                    dp.infer_cl = orig
                assert torch.isfinite(mel).all()
                res[dt] = float(((mel.cpu() - gi["new_mel_ref"]) ** 2).mean())
                assert rel_err(cf0.cpu(), gi["new_cf0_ref"]) < 1e-4
        print("mel MSE vs the reference's f32 golden: f16", res[torch.float16], "bf16", res[torch.bfloat16])
        assert res[torch.float16] < 1e-3, res
        assert res[torch.float16] <= res[torch.bfloat16] * 1.5 + 1e-6, res  # (measured: well below the bf16 figure)
        # the conditioning path has no f16 build: asked for, it refuses instead of silently computing in another dtype
        m.f32_conditioning = False
        try:
            with config.use_dtype(torch.float16), pytest.raises(NotImplementedError):
                m.infer_batch(gi["phon"].to(dev), gi["plen"].to(dev), noise_fn=noise_fn, **kw)
        finally:
            m.f32_conditioning = True
    finally:
        config.set_compute_dtype(torch.float32)
