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
