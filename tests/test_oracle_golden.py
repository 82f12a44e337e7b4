"""CPU: the oracle (oracle/ref_torch.py) against golden vectors captured from the
reference (oracle/gen_golden.py).  This is what pins the oracle."""
import torch
from conftest import key_shapes, load_golden, rel_err

from oracle import ref_torch as R
from oracle.fill import synth_state_dict, synth_tensor

VOC_GAIN = 0.4


def vocoder_sd(keys, seed):
    sd = {}
    taps = load_golden("aa_snake")["f_up"].reshape(1, 1, 12)
    for n, s in keys:
        if n.endswith("filter"):
            sd[n] = taps  # FIR taps are module-computed buffers, not synthetic
        else:
            sd[n] = torch.from_numpy(synth_tensor(n, s, seed, VOC_GAIN if n.endswith("weight_g") else 1.0))
    return sd


def test_aa_snake_matches_reference():
    g = load_golden("aa_snake")
    for T in [1, 2, 5, 6, 7, 13, 64, 131]:
        y = R.aa_snake(g[f"x{T}"], g["alpha"], g["f_up"], g["f_dn"])
        assert rel_err(y, g[f"y{T}"]) < 2e-6, T


def test_aa_filter_taps():
    g = load_golden("aa_snake")
    f = g["f_up"]
    assert torch.allclose(f, f.flip(0), atol=1e-8)  # linear phase
    assert abs(float(f.sum()) - 1.0) < 1e-6
    assert torch.equal(g["f_up"], g["f_dn"])
    # published-in-SURVEY probe values (section 4.1)
    ref6 = torch.tensor([0.0020289647, 0.0093894657, -0.0255434588, -0.0576573834, 0.1285725832, 0.4432097971])
    assert torch.allclose(f[:6], ref6, atol=2e-7)


def test_layer_norm_variants():
    g = load_golden("layer_norms")
    y = R.layer_norm_last(g["x_btc"], g["esp_w"], g["esp_b"], 1e-12)
    assert rel_err(y, g["esp_y"]) < 2e-6
    y = R.layer_norm_c(g["x_bct"], g["c_gamma"], g["c_beta"], 1e-5)
    assert rel_err(y, g["c_y"]) < 2e-6
    y = R.layer_norm_c(g["x_bct"], g["fp_gamma"], g["fp_beta"], 1e-5)
    assert rel_err(y, g["fp_y"]) < 2e-6


def test_masks_and_paths_bit_exact():
    g = load_golden("masks_paths")
    Tp, Tf = g["pmask"].shape[1], g["fmask"].shape[1]
    assert torch.equal(R.sequence_mask(g["plen"], Tp), g["pmask"].bool())
    assert torch.equal(R.sequence_mask(g["flen"], Tf), g["fmask"].bool())
    pm = g["pmask"][:, :, None].float() * g["fmask"][:, None, :].float()
    assert torch.equal(R.generate_path(g["dur"], pm.long()), g["path_int"])
    assert torch.equal(R.generate_path(g["dur"].float(), pm), g["path_f"])
    # integer gather form == dense path
    idx = R.frame_to_phone_index(g["dur"], Tf)
    dense = torch.zeros_like(g["path_int"])
    for b in range(idx.shape[0]):
        for f in range(Tf):
            if idx[b, f] >= 0 and f < int(g["flen"][b]):
                dense[b, idx[b, f], f] = 1
    assert torch.equal(dense, g["path_int"])
    # row sums of the path are the durations
    assert torch.equal(g["path_int"].sum(-1), g["dur"])


def test_bigvgan_matches_reference():
    g = load_golden("bigvgan")
    sd = vocoder_sd(key_shapes(g["keys"]), seed=31)
    y = R.bigvgan(sd, g["x"])
    assert y.shape == g["y"].shape
    assert rel_err(y, g["y"]) < 5e-5
    ya = R.amp_layer(sd, "mrfs.3.1.layers.2", g["amp_x"], 7, 5)
    assert rel_err(ya, g["amp_y"]) < 1e-5


def test_mel_front_end_restatement_against_torch_stft():
    """n1: the numpy restatement of the log-mel front-end (reference transforms/mel.py:18-34 on torchaudio's MelSpectrogram,
    conf/transforms/mel.yaml: power 1) pinned piece by piece: (1) its spectrum against torch.stft MAGNITUDES, power 1 and 2;
    (2) its slaney scale and filterbank against the published constants / properties of Slaney's Auditory-Toolbox mel
    scale (linear 200/3 Hz per mel below 1 kHz, 27 mels per factor 6.4 above; triangles of unit area between neighbouring
    band edges) -- the package derives its filterbank separately (promptttspp_amd/transforms), so (3) the package's CPU
    transform against the restatement closes the loop.  (Not a torchaudio pin: torchaudio is absent from this image.)"""
    import numpy as np

    from promptttspp.transforms import MelSpectrogramTransform

    rng = np.random.default_rng(7)
    wav = (0.3 * rng.standard_normal(24000 // 2)).astype(np.float32)
    st = torch.stft(torch.from_numpy(wav).double(), 512, 240, 480, torch.hann_window(480, dtype=torch.float64), center=True,
                    pad_mode="reflect", return_complex=True).abs()
    for power in (1.0, 2.0):
        spec, fb, fpts, hz2mel, mel2hz = R.mel_spectrogram_np(wav, power=power, parts=True)
        assert spec.shape == tuple(st.shape) == (257, 51)
        assert float((torch.from_numpy(spec) - st.pow(power)).abs().max() / st.pow(power).max()) < 1e-10   # (1) magnitudes
        ref = R.mel_spectrogram_np(wav, power=power)
        t = MelSpectrogramTransform(sample_rate=24000, n_fft=512, win_length=480, hop_length=240, f_min=63.0, f_max=12000.0,
                                    n_mels=80, norm="slaney", mel_scale="slaney", power=power)
        mel = t(torch.from_numpy(wav)[None])[0]
        assert mel.shape == ref.shape == (80, 51)
        assert float((mel.double() - torch.from_numpy(ref)).abs().max()) < 2e-4   # (3) log domain, f32 STFT vs f64
    # (2) the scale: fixed points of Slaney's definition
    assert abs(float(hz2mel(1000.0)) - 15.0) < 1e-12 and abs(float(hz2mel(200.0)) - 3.0) < 1e-12
    assert abs(float(hz2mel(6400.0)) - 42.0) < 1e-9 and abs(float(mel2hz(42.0)) - 6400.0) < 1e-6
    f = np.array([63.0, 500.0, 999.0, 1000.0, 1001.0, 4000.0, 12000.0])
    assert np.allclose(mel2hz(hz2mel(f)), f, rtol=1e-12)
    # band edges: 82 points equally spaced in mel between f_min and f_max
    assert fpts.shape == (82,) and abs(fpts[0] - 63.0) < 1e-9 and abs(fpts[-1] - 12000.0) < 1e-6
    assert np.allclose(np.diff(hz2mel(fpts)), np.diff(hz2mel(fpts))[0], rtol=1e-9)
    # filterbank: non-negative triangles, filter m supported strictly inside (f[m], f[m+2]), peak of height 2 / (f[m+2] - f[m])
    # at f[m+1] (slaney area normalisation: unit area in Hz)
    freqs = np.linspace(0, 12000, 257)
    assert fb.shape == (257, 80) and float(fb.min()) >= 0.0
    for m in (0, 1, 17, 40, 79):
        sup = freqs[fb[:, m] > 0]
        assert sup.min() > fpts[m] and sup.max() < fpts[m + 2]
        tri = np.interp(freqs, [fpts[m], fpts[m + 1], fpts[m + 2]], [0.0, 2.0 / (fpts[m + 2] - fpts[m]), 0.0], left=0.0, right=0.0)
        assert np.allclose(fb[:, m], tri, atol=1e-12)


def test_zero_state_filtfilt_restatement_against_scipy():
    """n2: forward-backward lfilter with zero initial state (what torchaudio's filtfilt computes) vs scipy.signal.lfilter
    applied twice; and that it is NOT scipy.signal.filtfilt (which pads the edges)."""
    import numpy as np
    from scipy import signal

    from promptttspp.utils.model import lowpass_filter

    b, a = signal.butter(5, [20 / 50], "lowpass")
    x = 5.2 + 0.3 * np.random.default_rng(3).standard_normal((2, 1, 300))
    y = R.filtfilt_zero_state(x, b, a)
    sp = signal.lfilter(b, a, signal.lfilter(b, a, x, axis=-1)[..., ::-1], axis=-1)[..., ::-1]
    assert np.abs(y - sp).max() < 1e-9
    assert np.abs(y - signal.filtfilt(b, a, x)).max() > 1e-2
    out = lowpass_filter(torch.from_numpy(x).float(), 100, cutoff=20)          # the package's CPU tensor path
    assert float((out.double() - torch.from_numpy(np.ascontiguousarray(y))).abs().max()) < 1e-5
    short = torch.randn(1, 1, 10)
    assert torch.equal(lowpass_filter(short, 100, cutoff=20), short)             # too short: returned unchanged (:180-182)
