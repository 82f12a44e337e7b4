"""CPU: acoustic-model oracle functions against golden vectors captured from the
reference (oracle/gen_golden_am.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from conftest import key_shapes, load_golden, rel_err

from oracle import ref_torch as R
from oracle.fill import synth_tensor

TAME = {"duration_predictor.out_layer.mu.weight": 0.5, "duration_predictor.out_layer.log_sigma.weight": 0.1}
TAME_OFF = {"duration_predictor.out_layer.mu.bias": 1.3, "duration_predictor.out_layer.log_sigma.bias": -1.5}
SCHEDULE = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
            "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
            "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2")


def synth_sd(keys, seed, overrides=None, offsets=None, prefix=""):
    """state dict for [(name, shape)] with the generator's fill rules."""
    sd = {}
    sch = R.diffusion_schedule()
    for n, s in keys:
        leaf = n.rsplit(".", 1)[-1]
        if leaf in SCHEDULE:
            sd[prefix + n] = sch[leaf]
            continue
        if leaf in ("num_batches_tracked", "position_ids", "token_type_ids", "filter"):
            continue
        g = 1.0
        for suf, og in (overrides or {}).items():
            if n.endswith(suf):
                g = og
        t = torch.from_numpy(synth_tensor(n, s, seed, g))
        for suf, off in (offsets or {}).items():
            if n.endswith(suf):
                t = t + off
        sd[prefix + n] = t
    return sd


@pytest.mark.parametrize("variant", ["new", "legacy"])
def test_conformer(variant):
    g = load_golden("conformer")
    sd = synth_sd(key_shapes(g[f"keys_{variant}"]), 40, prefix="enc.")
    y = R.conformer_encoder(sd, "enc", g["x"], g["lens"], variant=variant)
    assert rel_err(y, g[f"y_{variant}"]) < 2e-5
    yt = R.conformer_encoder(sd, "enc", g["x"], g["lens"], variant=variant, train_bn=True)
    assert rel_err(yt, g[f"y_{variant}_trainbn"]) < 2e-5


def test_mdn():
    g = load_golden("mdn")
    sd = synth_sd(key_shapes(g["keys_d"]), 50, prefix="m.")
    lp, ls, mu = R.mdn_layer(sd, "m", g["x"], 4, 1)
    for a, b in ((lp, g["lp"]), (ls, g["ls"]), (mu, g["mu"])):
        assert rel_err(a, b) < 1e-5
    assert rel_err(R.mdn_loss(lp, ls, mu, g["tgt"], g["mask"].bool())[g["mask"].bool().squeeze(-1)],
                   g["loss_m"][g["mask"].bool().squeeze(-1)]) < 1e-5
    assert rel_err(R.mdn_loss(lp, ls, mu, g["tgt"]), g["loss_u"]) < 1e-5
    sg, mm = R.mdn_most_probable(lp, ls, mu)
    assert torch.equal(lp.argmax(2), g["idx"])  # integer: bit exact
    assert rel_err(sg, g["sg"]) < 1e-5 and rel_err(mm, g["mm"]) < 1e-5
    sds = synth_sd(key_shapes(g["keys_s"]), 51, prefix="s.")
    out = R.mdn_layer(sds, "s", g["xs"], 10, 256)
    assert rel_err(R.mdn_loss(*out, g["ts"]).mean(1), g["loss_s"]) < 1e-5
    assert torch.equal(out[0].argmax(2), g["idxs"])


def test_variance_adaptor():
    g = load_golden("variance_adaptor")
    sd = synth_sd(key_shapes(g["keys"]), 60, TAME, TAME_OFF, prefix="va.")
    Tp = g["x"].shape[-1]
    pm = R.sequence_mask(g["plen"], Tp).unsqueeze(1)
    fm = R.sequence_mask(g["flen"], g["cf0"].shape[-1]).unsqueeze(1).float()
    h, (lp, ls, mu), cf0p, vuvp = R.variance_adaptor_forward(sd, "va", g["x"], pm, fm, g["dur"], g["cf0"])
    assert rel_err(h, g["h"]) < 2e-5
    assert rel_err(lp, g["lp"]) < 1e-5 and rel_err(mu, g["mu"]) < 1e-5 and rel_err(ls, g["ls"]) < 1e-5
    assert rel_err(cf0p, g["cf0p"]) < 2e-5 and rel_err(vuvp, g["vuvp"]) < 2e-5
    assert rel_err(R.frame_prior(sd, "va.frame_prior_network", g["fp_x"], fm), g["fp_y"]) < 2e-5
    hi, fmi, cf0i, vuvi, dur, flen = R.variance_adaptor_infer_batch(sd, "va", g["x"], pm.long())
    assert torch.equal(dur, g["duri"])  # integer durations: bit exact
    assert torch.equal(fmi, g["fmi"])
    assert rel_err(hi, g["hi"]) < 2e-5 and rel_err(cf0i, g["cf0i"]) < 2e-5 and rel_err(vuvi, g["vuvi"]) < 2e-5


def test_variance_adaptor_energy_branch():
    """energy_predictor reads the frame-prior output before pitch_emb is added (variance_adaptor.py:139-146)."""
    g = load_golden("variance_adaptor_energy")
    sd = synth_sd(key_shapes(g["keys"]), 65, TAME, TAME_OFF, prefix="va.")
    Tp = g["x"].shape[-1]
    pm = R.sequence_mask(g["plen"], Tp).unsqueeze(1)
    fm = R.sequence_mask(g["flen"], g["cf0"].shape[-1]).unsqueeze(1).float()
    h, _, cf0p, vuvp, enp = R.variance_adaptor_forward(sd, "va", g["x"], pm, fm, g["dur"], g["cf0"], g["energy"])
    assert rel_err(h, g["h"]) < 2e-5 and rel_err(enp, g["enp"]) < 2e-5
    assert rel_err(cf0p, g["cf0p"]) < 2e-5 and rel_err(vuvp, g["vuvp"]) < 2e-5
    hi, fmi, cf0i, vuvi, _, _ = R.variance_adaptor_infer_batch(sd, "va", g["x"], pm.long())
    assert torch.equal(fmi, g["fmi"])
    assert rel_err(hi, g["hi"]) < 2e-5 and rel_err(cf0i, g["cf0i"]) < 2e-5
    h1, _, cf01, _, _, _ = R.variance_adaptor_infer_batch(sd, "va", g["x"][:1], pm[:1].long())
    assert rel_err(h1, g["h1"]) < 2e-5 and rel_err(cf01, g["cf01"]) < 2e-5


def test_style_encoder():
    g = load_golden("style_encoder")
    sd = synth_sd(key_shapes(g["keys"]), 70, prefix="se.")
    assert rel_err(R.style_encoder(sd, "se", g["mel"], g["lens"]), g["y"]) < 2e-5
    assert rel_err(R.style_encoder(sd, "se", g["mel"], g["lens"], train_bn=True), g["y_trainbn"]) < 2e-5


def test_bert_cls():
    g = load_golden("bert")
    sd = synth_sd(key_shapes(g["keys"]), 80, prefix="b.")
    assert rel_err(R.bert_cls(sd, "b.", g["ids"], g["am"]), g["cls"]) < 5e-5


def test_diffusion():
    g = load_golden("diffusion")
    sch = R.diffusion_schedule()
    for k in SCHEDULE:
        assert torch.equal(sch[k], g["buf_" + k]), k  # float64 -> float32 schedule: bit exact
    sd = synth_sd(key_shapes(g["keys"]), 90, prefix="dec.")
    nz, pred = R.diffusion_train(sd, "dec", g["cond"].transpose(1, 2), g["mel"].transpose(1, 2), g["mask"], g["t"],
                                 g["noise"])
    assert rel_err(pred.transpose(1, 2), g["pred"]) < 2e-5
    assert torch.equal(nz.transpose(1, 2), g["nz"])
    eps = R.diffnet(sd, "dec.denoise_fn", g["eps_x"], g["t"], g["cond"].transpose(1, 2), g["mask"])
    assert rel_err(eps, g["eps"]) < 2e-5
    B, _, T = g["x_init"].shape
    steps = [torch.from_numpy((np.random.default_rng(1000 + i).standard_normal((B, 80, T))).astype(np.float32))
             for i in range(100)]
    y = R.diffusion_sample(sd, "dec", g["cond"].transpose(1, 2), g["x_init"], steps)
    assert rel_err(y.transpose(1, 2), g["sampled"]) < 1e-4


def _model_sd(keys):
    return synth_sd(keys, 100, TAME, TAME_OFF)


def test_model_forward_losses():
    g = load_golden("model_forward")
    sd = _model_sd(key_shapes(g["keys"]))
    batch = (g["phon"], g["dur"], g["plen"], g["mel"], g["cf0"], g["vuv"], g["flen"], g["ids"], g["am"])
    out = R.model_forward(sd, batch, g["t"], g["noise"])
    for k in ("loss", "dec", "dur", "cf0", "vuv", "style"):
        assert abs(float(out[k]) - float(g["ev_" + k])) < 2e-5 * max(1.0, abs(float(g["ev_" + k]))), k
    tr = R.model_forward(sd, batch, g["t"], g["noise"], train_bn=True)
    for k in ("loss", "dec", "dur", "cf0", "vuv", "style"):
        assert abs(float(tr[k]) - float(g["tr_" + k])) < 2e-5 * max(1.0, abs(float(g["tr_" + k]))), k


def test_model_forward_grads():
    g = load_golden("model_forward")
    sd = _model_sd(key_shapes(g["keys"]))
    names = [k[2:] for k in g if k.startswith("g:")]
    for n in names:
        sd[n] = sd[n].clone().requires_grad_()
    batch = (g["phon"], g["dur"], g["plen"], g["mel"], g["cf0"], g["vuv"], g["flen"], g["ids"], g["am"])
    loss = R.model_forward(sd, batch, g["t"], g["noise"], train_bn=True)["loss"]
    grads = torch.autograd.grad(loss, [sd[n] for n in names], allow_unused=True)
    for n, gr in zip(names, grads):
        ref = g["g:" + n]
        if gr.numel() > 70000:
            gr = gr.flatten()[:: max(1, gr.numel() // 4096)][:4096]
        # (the BERT gradient runs through a 12-layer fp32 stack: looser)
        assert rel_err(gr, ref.reshape(gr.shape)) < (3e-3 if "bert" in n else 5e-4), n


def test_model_infer_batch_integer_and_mel():
    g = load_golden("model_infer")
    keys = key_shapes(load_golden("model_forward")["keys"])
    sd = _model_sd(keys)
    for variant in ("new", "legacy"):
        Tf = int(g[f"{variant}_flen_ref"].max())
        B = g["phon"].shape[0]
        mk = lambda base: [torch.from_numpy(np.random.default_rng(base + i).standard_normal((B, 80, Tf)).astype(np.float32))  # noqa: E731
                           for i in range(100)]
        x_init = torch.from_numpy(np.random.default_rng(112).standard_normal((B, 80, Tf)).astype(np.float32))
        mel, cf0, vuv, flen, dur = R.model_infer_batch(sd, g["phon"], g["plen"], lambda b, t: x_init, lambda b, t: mk(2000),
                                                       ref_mel=g["mel"], ref_len=g["flen_in"], variant=variant)
        assert torch.equal(dur, g[f"{variant}_dur_ref"])       # integer: bit exact
        assert torch.equal(flen, g[f"{variant}_flen_ref"])
        assert rel_err(mel, g[f"{variant}_mel_ref"]) < 2e-4
        assert rel_err(cf0, g[f"{variant}_cf0_ref"]) < 2e-5
    # prompt path
    B = g["phon"].shape[0]
    Tf2 = int(g["prompt_flen"].max())
    x2 = torch.from_numpy(np.random.default_rng(114).standard_normal((B, 80, Tf2)).astype(np.float32))
    st2 = [torch.from_numpy(np.random.default_rng(3000 + i).standard_normal((B, 80, Tf2)).astype(np.float32)) for i in range(100)]
    mel, cf0, vuv, flen, dur = R.model_infer_batch(sd, g["phon"], g["plen"], lambda b, t: x2, lambda b, t: st2, ids=g["ids"],
                                                   am=g["am"], style_noise=g["style_noise"], noise_scale=0.5)
    assert torch.equal(dur, g["prompt_dur"]) and torch.equal(flen, g["prompt_flen"])
    assert rel_err(mel, g["prompt_mel"]) < 2e-4


def single_noise(seed, Tf):
    x0 = torch.from_numpy(np.random.default_rng(seed).standard_normal((1, 80, Tf)).astype(np.float32))
    steps = [torch.from_numpy(np.random.default_rng(seed * 100 + i).standard_normal((1, 80, Tf)).astype(np.float32))
             for i in range(100)]
    return x0, steps


def test_model_infer_single_utterance():
    """infer() (model.py:198-259) on both style branches, use_max True / False (injected draw), and
    generate_style_emb (model.py:327-344)."""
    g = load_golden("model_infer_single")
    sd = _model_sd(key_shapes(load_golden("model_forward")["keys"]))
    plen = torch.tensor([g["phon"].shape[1]])
    cases = [("prompt_max", 215, dict(ids=g["ids"], am=g["am"], style_noise=g["style_noise"], noise_scale=0.5)),
             ("ref", 216, dict(ref_mel=g["mel"], ref_len=torch.tensor([g["mel"].shape[-1]]))),
             ("prompt_sample", 217, dict(ids=g["ids"], am=g["am"], style_noise=g["style_noise"], noise_scale=0.7,
                                         comp=g["comp"]))]
    for tag, seed, kw in cases:
        Tf = g[tag + "_mel"].shape[-1]
        x0, steps = single_noise(seed, Tf)
        mel, cf0, vuv, flen, dur = R.model_infer_batch(sd, g["phon"], plen, lambda b, t: x0, lambda b, t: steps, **kw)
        assert int(flen[0]) == Tf, tag                       # integer frame count: exact
        assert rel_err(mel, g[tag + "_mel"]) < 2e-4, tag
        assert rel_err(cf0, g[tag + "_cf0"]) < 2e-5 and rel_err(vuv, g[tag + "_vuv"]) < 2e-5, tag
    pe = R.style_from_prompt(sd, g["ids"], g["am"], g["style_noise"], 0.5)
    # generate_style_emb normalises the sampled embedding once more over dim 1 (model.py:337-338): idempotent
    assert rel_err(F.normalize(pe, dim=1), g["gen_prompt_emb"]) < 2e-5
    re_ = F.normalize(R.style_encoder(sd, "reference_encoder", g["mel"], torch.tensor([g["mel"].shape[-1]])), dim=1)
    assert rel_err(re_, g["gen_ref_emb"]) < 2e-5


def test_transformer_plugin():
    """modules/transformer.py: both attention flavours, with and without the per-layer conditioning g, and gradients."""
    g = load_golden("transformer")
    mask = R.sequence_mask(g["lens"], g["x"].shape[-1]).unsqueeze(1).float()
    for tag, rel in (("rel", True), ("abs", False)):
        sd = synth_sd(key_shapes(g[f"{tag}_keys"]), 510 + int(rel), {"emb_rel_k": 0.5, "emb_rel_v": 0.5}, prefix="t.")
        assert rel_err(R.transformer(sd, "t", g["x"], mask, use_rel=rel), g[f"{tag}_y"]) < 2e-5
        assert rel_err(R.transformer(sd, "t", g["x"], mask, use_rel=rel, g=g["g"]), g[f"{tag}_yg"]) < 2e-5
        names = [k[len(tag) + 3:] for k in g if k.startswith(f"{tag}_g:")]
        for n in names:
            sd["t." + n] = sd["t." + n].clone().requires_grad_()
        xx = g["x"].clone().requires_grad_()
        grads = torch.autograd.grad(R.transformer(sd, "t", xx, mask, use_rel=rel), [xx] + [sd["t." + n] for n in names], g["dy"])
        assert rel_err(grads[0], g[f"{tag}_dx"]) < 2e-5
        for n, gr in zip(names, grads[1:]):
            ref = g[f"{tag}_g:{n}"]
            if gr.numel() > 70000:
                gr = gr.flatten()[:: max(1, gr.numel() // 4096)][:4096]
            assert rel_err(gr, ref.reshape(gr.shape)) < 5e-5, n


def test_plms_sampler():
    g = load_golden("diffusion_plms")
    sd = synth_sd(key_shapes(g["keys"]), 90, prefix="dec.")
    for interval in (10, 25):
        mel = R.diffusion_sample_plms(sd, "dec", g["cond"].transpose(1, 2), g["x_init"], interval)
        assert rel_err(mel.transpose(1, 2), g[f"mel_{interval}"]) < 1e-4, interval


def test_nsf_source_and_f0_vocoder():
    from test_oracle_golden import vocoder_sd

    g = load_golden("bigvgan_f0")
    sd = vocoder_sd(key_shapes(g["keys"]), 120)
    src = R.nsf_source(sd, g["f0"], g["rand_ini"], g["nz"])
    assert rel_err(src, g["src"]) < 1e-4
    y = R.bigvgan(sd, g["x"], source=src)
    assert rel_err(y, g["y"]) < 1e-4
