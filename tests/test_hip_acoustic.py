"""GPU: the acoustic-model modules on the HIP path against golden vectors captured
from the reference (f32 compute mode, tolerance 1e-3 relative per north_star; the
asserts hold what we actually achieve, usually ~1e-5) and integer outputs bit-exact."""
import os
import warnings

import numpy as np
import pytest
import torch
from conftest import ROOT, key_shapes, load_golden, rel_err
from test_oracle_golden_am import TAME, TAME_OFF, synth_sd

from oracle import ref_torch as R

warnings.simplefilter("ignore")
pytestmark = pytest.mark.gpu
CONF = os.path.join(ROOT, "egs", "proposed", "bin", "conf", "model")


def node(name, variant="new"):
    from promptttspp_amd import hydra_lite as H

    f = "prompttts_mdn_v2_wo_erg_final" + ("_demo" if variant == "legacy" else "") + ".yaml"
    cfg = H.load_node(os.path.join(CONF, f))
    return H.instantiate(cfg[name] if name else cfg)


def load(m, keys, seed, dev, overrides=None, offsets=None):
    sd = synth_sd(keys, seed, overrides, offsets)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith(("num_batches_tracked", "position_ids", "token_type_ids")) for k in missing), missing
    return m.to(dev), sd


@pytest.fixture(autouse=True)
def f32_mode():
    from promptttspp_amd import config

    with config.use_dtype(torch.float32):
        yield


def rnd(seed, *shape, scale=1.0):
    return torch.from_numpy((scale * np.random.default_rng(seed).standard_normal(shape)).astype(np.float32))


# ---------------------------------------------------------------------------------------
def test_length_regulator_is_the_path_matmul(dev):
    from promptttspp_amd import functional as PF

    g = load_golden("masks_paths")
    B, Tp = g["dur"].shape
    Tf = g["path_int"].shape[-1]
    x = rnd(1, B, Tp, 256).requires_grad_()
    ref = (x.transpose(1, 2) @ g["path_f"]).transpose(1, 2)  # x @ path, channels-last
    dy = rnd(2, B, Tf, 256)
    (dx_ref,) = torch.autograd.grad(ref, x, dy)
    xg = x.detach().to(dev).requires_grad_()
    y = PF.length_regulate(xg, g["dur"].to(dev), Tf)
    assert torch.equal(y.cpu(), ref.detach())  # pure copy: bit exact
    y.backward(dy.to(dev))
    assert rel_err(xg.grad.cpu(), dx_ref) < 1e-6


def test_attention_variants_and_grads(dev):
    from promptttspp_amd import functional as PF

    B, T, H, C = 3, 21, 2, 256
    lens = torch.tensor([21, 13, 1])
    km = R.sequence_mask(lens, T)
    sd = {"a.pos_bias_u": 0.1 * rnd(1, H, C // H), "a.pos_bias_v": 0.1 * rnd(2, H, C // H)}
    for n in ("q", "k", "v", "out", "pos"):
        sd[f"a.linear_{n}.weight"] = torch.eye(C)
        if n != "pos":
            sd[f"a.linear_{n}.bias"] = torch.zeros(C)
    for variant in ("new", "legacy"):
        qkv = rnd(3, B, T, 3 * C, scale=0.5).requires_grad_()
        pos = R.rel_pos_emb(T, C, variant).requires_grad_()
        # oracle with identity projections: feed q,k,v through separate tensors
        def oracle(qkv_, pos_):
            q, k, v = qkv_[..., :C], qkv_[..., C:2 * C], qkv_[..., 2 * C:]
            dk = C // H
            qh = q.view(B, T, H, dk)
            kh, vh = k.view(B, T, H, dk).transpose(1, 2), v.view(B, T, H, dk).transpose(1, 2)
            pp = pos_.view(-1, H, dk).transpose(0, 1)
            qu, qv = (qh + sd["a.pos_bias_u"]).transpose(1, 2), (qh + sd["a.pos_bias_v"]).transpose(1, 2)
            bd = R._pad_view_shift(qv @ pp.transpose(-1, -2)[None], T)
            sc = (qu @ kh.transpose(-1, -2) + bd) / np.sqrt(dk)
            m2 = (km[:, :, None] & km[:, None, :])[:, None]
            a = torch.softmax(sc.masked_fill(~m2, torch.finfo(sc.dtype).min), -1).masked_fill(~m2, 0.0)
            return (a @ vh).transpose(1, 2).reshape(B, T, C)
        ref = oracle(qkv, pos)
        u = sd["a.pos_bias_u"].to(dev).requires_grad_()
        vb = sd["a.pos_bias_v"].to(dev).requires_grad_()
        qg, pg = qkv.detach().to(dev).requires_grad_(), pos.detach().to(dev).requires_grad_()
        out = PF.attention(qg, pg, u, vb, lens.to(dev).int(), H, variant)
        assert rel_err(out.cpu(), ref.detach()) < 1e-5, variant
        if True:  # both tables are trainable (round 3: the legacy pad/view shift has a hand-written backward too)
            dy = rnd(4, B, T, C)
            us, vs = sd["a.pos_bias_u"].clone().requires_grad_(), sd["a.pos_bias_v"].clone().requires_grad_()
            sd2 = dict(sd)
            sd["a.pos_bias_u"], sd["a.pos_bias_v"] = us, vs
            gq, gp, gu, gv = torch.autograd.grad(oracle(qkv, pos), (qkv, pos, us, vs), dy)
            sd.update(sd2)
            out.backward(dy.to(dev))
            assert rel_err(qg.grad.cpu(), gq) < 1e-5, variant
            assert pg.grad.shape == gp.shape and rel_err(pg.grad.cpu(), gp) < 1e-5, variant
            assert rel_err(u.grad.cpu(), gu) < 1e-5 and rel_err(vb.grad.cpu(), gv) < 1e-5, variant


@pytest.mark.parametrize("variant", ["new", "legacy"])
def test_conformer_matches_reference(variant, dev):
    g = load_golden("conformer")
    m, _ = load(node("encoder", variant), key_shapes(g[f"keys_{variant}"]), 40, dev)
    m.eval()
    with torch.no_grad():
        y = m(g["x"].to(dev), g["lens"].to(dev)).cpu()
    assert rel_err(y, g[f"y_{variant}"]) < 1e-3
    assert rel_err(y, g[f"y_{variant}"]) < 5e-5
    for mod in m.modules():  # train mode, dropout off: BatchNorm batch statistics incl. padding
        if hasattr(mod, "dropout_rate"):
            mod.dropout_rate = 0.0
        if hasattr(mod, "positional_dropout_rate"):
            mod.positional_dropout_rate = 0.0
    m.train()
    with torch.no_grad():
        yt = m(g["x"].to(dev), g["lens"].to(dev)).cpu()
    assert rel_err(yt, g[f"y_{variant}_trainbn"]) < 5e-5


def test_mdn_layer_and_losses(dev):
    from promptttspp_amd.modules.mdn import mdn_get_most_probable_sigma_and_mu, mdn_loss

    g = load_golden("mdn")
    from promptttspp_amd.modules.mdn import MDNLayer

    m, _ = load(MDNLayer(256, 1, 4, True), key_shapes(g["keys_d"]), 50, dev)
    lp, ls, mu = m(g["x"].to(dev))
    for a, b in ((lp, g["lp"]), (ls, g["ls"]), (mu, g["mu"])):
        assert rel_err(a.detach().cpu(), b) < 1e-5
    mask = g["mask"].bool().to(dev)
    lm = mdn_loss(lp, ls, mu, g["tgt"].to(dev), reduce=False, mask=mask)
    sel = g["mask"].bool().squeeze(-1)
    assert rel_err(lm.detach().cpu()[sel], g["loss_m"][sel]) < 1e-5
    assert rel_err(mdn_loss(lp, ls, mu, g["tgt"].to(dev), reduce=False).detach().cpu(), g["loss_u"]) < 1e-5
    sg, mm = mdn_get_most_probable_sigma_and_mu(lp, ls, mu)
    assert torch.equal(lp.argmax(2).cpu(), g["idx"])  # integer: bit exact
    assert rel_err(sg.detach().cpu(), g["sg"]) < 1e-5 and rel_err(mm.detach().cpu(), g["mm"]) < 1e-5
    s, _ = load(MDNLayer(256, 256, 10, True), key_shapes(g["keys_s"]), 51, dev)
    out = s(g["xs"].to(dev))
    assert rel_err(mdn_loss(*out, g["ts"].to(dev)).detach().cpu(), g["loss_s"]) < 1e-5
    assert torch.equal(out[0].argmax(2).cpu(), g["idxs"])


def test_variance_adaptor(dev):
    g = load_golden("variance_adaptor")
    m, _ = load(node("variance_adaptor"), key_shapes(g["keys"]), 60, dev, TAME, TAME_OFF)
    m.eval()
    Tp, Tf = g["x"].shape[-1], g["cf0"].shape[-1]
    pm = R.sequence_mask(g["plen"], Tp).unsqueeze(1).long().to(dev)
    fm = R.sequence_mask(g["flen"], Tf).unsqueeze(1).float().to(dev)
    with torch.no_grad():
        h, (lp, ls, mu), cf0p, vuvp, _ = m(g["x"].to(dev), pm, fm, g["dur"].to(dev), g["cf0"].to(dev), None, None)
        assert rel_err(h.cpu(), g["h"]) < 5e-5
        assert rel_err(lp.cpu(), g["lp"]) < 1e-5 and rel_err(mu.cpu(), g["mu"]) < 1e-5 and rel_err(ls.cpu(), g["ls"]) < 1e-5
        assert rel_err(cf0p.cpu(), g["cf0p"]) < 5e-5 and rel_err(vuvp.cpu(), g["vuvp"]) < 5e-5
        fp = m.frame_prior_network(g["fp_x"].to(dev), fm)
        assert rel_err(fp.cpu(), g["fp_y"]) < 5e-5
        hi, fmi, cf0i, vuvi = m.infer_batch(g["x"].to(dev), pm, return_f0=True)
        logd = m.duration_predictor.infer(g["x"].to(dev), pm)
    dur = (logd.exp().round().clamp_min(1).long() * pm).cpu()
    assert torch.equal(dur, g["duri"])  # integer durations: bit exact
    assert torch.equal(fmi.cpu(), g["fmi"])
    assert rel_err(logd.cpu(), g["logd"]) < 1e-5
    assert rel_err(hi.cpu(), g["hi"]) < 5e-5 and rel_err(cf0i.cpu(), g["cf0i"]) < 5e-5 and rel_err(vuvi.cpu(), g["vuvi"]) < 5e-5


def test_style_encoder(dev):
    g = load_golden("style_encoder")
    m, _ = load(node("reference_encoder"), key_shapes(g["keys"]), 70, dev)
    m.eval()
    with torch.no_grad():
        assert rel_err(m(g["mel"].to(dev), g["lens"].to(dev)).cpu(), g["y"]) < 1e-4
    m.train()
    with torch.no_grad():
        assert rel_err(m(g["mel"].to(dev), g["lens"].to(dev)).cpu(), g["y_trainbn"]) < 1e-4


def test_diffusion_train_and_sampler(dev):
    g = load_golden("diffusion")
    m, _ = load(node("decoder"), key_shapes(g["keys"]), 90, dev)
    m.eval()
    for k in ("betas", "posterior_mean_coef2", "sqrt_recipm1_alphas_cumprod", "posterior_log_variance_clipped"):
        assert torch.equal(getattr(m, k).cpu(), g["buf_" + k])  # schedule buffers: bit exact
    m.injected = {"t": g["t"], "noise": g["noise"]}
    with torch.no_grad():
        nz, pred = m(cond=g["cond"].to(dev), y=g["mel"].to(dev), mask=g["mask"].to(dev))
        assert torch.equal(nz.cpu(), g["nz"])
        assert rel_err(pred.cpu(), g["pred"]) < 5e-5
        eps = m.denoise_fn(g["eps_x"].to(dev), g["t"].to(dev), g["cond"].transpose(1, 2).to(dev), mask=g["mask"].to(dev))
        assert rel_err(eps.cpu(), g["eps"]) < 5e-5
        B, _, T = g["x_init"].shape
        def noise_fn(i, shape):
            if i < 0:
                return g["x_init"].transpose(1, 2).contiguous().to(dev)
            return rnd(1000 + i, B, 80, T).transpose(1, 2).contiguous().to(dev)
        y = m.inference_cl(g["cond"].to(dev), noise_fn)
    assert rel_err(y.cpu(), g["sampled"]) < 1e-3
    assert float(((y.cpu() - g["sampled"]) ** 2).mean()) < 1e-3  # mel MSE vs reference


def _model(dev, variant="new"):
    g = load_golden("model_forward")
    m, sd = load(node(None, variant), key_shapes(g["keys"]), 100, dev, TAME, TAME_OFF)
    return m, g


def _batch(g, dev):
    d = lambda k: g[k].to(dev)  # noqa: E731
    return [d("phon"), d("dur"), d("plen"), d("mel"), d("cf0"), d("vuv"), torch.zeros_like(d("cf0")), d("flen"),
            (d("ids"), d("am"))]


def test_model_forward_losses_eval(dev):
    m, g = _model(dev)
    m.eval()
    m.decoder.injected = {"t": g["t"], "noise": g["noise"]}
    dur_before = g["dur"].clone()
    batch = _batch(g, dev)
    with torch.no_grad():
        out = m(batch)
    assert torch.equal(batch[1].cpu(), dur_before)  # no in-place mutation of the caller's durations
    for k in ("loss", "dec", "dur", "cf0", "vuv", "style"):
        ref = float(g["ev_" + k])
        assert abs(float(out[k]) - ref) < 1e-3 * max(1.0, abs(ref)), k   # north_star tolerance
        assert abs(float(out[k]) - ref) < 1e-4 * max(1.0, abs(ref)), k   # held


def test_model_train_step_losses_and_grads(dev):
    m, g = _model(dev)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        for a in ("dropout_rate", "positional_dropout_rate", "p_dropout", "p"):
            if isinstance(getattr(mod, a, None), float):
                setattr(mod, a, 0.0)
    m.train()
    m.decoder.injected = {"t": g["t"], "noise": g["noise"]}
    out = m(_batch(g, dev))
    for k in ("loss", "dec", "dur", "cf0", "vuv", "style"):
        ref = float(g["tr_" + k])
        assert abs(float(out[k]) - ref) < 1e-4 * max(1.0, abs(ref)), k
    out["loss"].backward()
    params = dict(m.named_parameters())
    # Cross-device gradient comparison of the WHOLE model is limited by the ~0.5 M ReLU
    # pre-activations of this batch: a handful lie within ~1e-5 of zero, where the CPU
    # reference and the GPU kernels legitimately round to different signs (observed: one
    # element at -5e-6 vs +3e-6 changes the max-norm error of upstream gradients to ~1e-2
    # while everything else agrees to 1e-6; tools/diag_combo.py).  So: tight agreement for
    # (almost) all elements + bounded L2 error here, and TIGHT (1e-5) module-level gradient
    # checks against the oracle in test_module_gradients_match_oracle below.
    for key in [k for k in g if k.startswith("g:")]:
        name = key[2:]
        gr = params[name].grad
        assert gr is not None, name
        gr = gr.detach().cpu()
        if gr.numel() > 70000:
            gr = gr.flatten()[:: max(1, gr.numel() // 4096)][:4096]
        ref = g[key].reshape(gr.shape)
        scale = float(ref.abs().max()) + 1e-30
        if scale < 1e-7:
            continue  # structurally zero gradient
        l2 = float((gr - ref).norm() / (ref.norm() + 1e-30))
        close = float(((gr - ref).abs() <= 2e-2 * scale).float().mean())
        assert l2 < 3e-2 and close > 0.98, (name, l2, close)
    total = sum(float(p.grad.double().pow(2).sum()) for p in m.parameters() if p.grad is not None)
    assert abs(np.sqrt(total) - float(g["grad_norm"])) < 1e-3 * float(g["grad_norm"])


def _zero_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        for a in ("dropout_rate", "positional_dropout_rate", "p_dropout", "p"):
            if isinstance(getattr(mod, a, None), float):
                setattr(mod, a, 0.0)
    return m.train()


def l2_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _cmp_grads(module, prefix, sdo, names, grads_ref, tol=2e-5, err=rel_err, log=None):
    P = dict(module.named_parameters())
    for n, gg in zip(names, grads_ref):
        e = err(P[n[len(prefix):]].grad.cpu(), gg)
        if log is not None:
            log.append((n, e))
        else:
            assert e < tol, (n, e)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_module_gradients_match_oracle(mode, dev):
    """Forward + hand-written backward of each HIP-backed module against torch autograd of
    the f32 ORACLE on the same inputs (train mode, dropout off).  f32 compute: element-wise 2e-5 of the
    largest element.  bf16 compute (the benchmarked mode: bf16 storage, bf16 MFMA, f32 accumulation):
    relative L2 error of every output and gradient against the ORACLE (not against this package's own f32 path),
    bounded per module at ~1.5x what was measured on MI355X (profiles/r02_bf16_module_errors.txt): frame prior
    0.9 %, Conformer 1-6 %, DiffNet stack 6-7 %, pitch predictor 1.4 % forward but 9-14 % in the gradients (five
    ReLU -> LayerNorm layers on synthetic weights: a pre-activation that changes sign under bf16 rounding switches a
    whole gradient path)."""
    from promptttspp_amd import config, ops

    with config.use_dtype(torch.float32 if mode == "f32" else torch.bfloat16):
        _module_gradients(dev, ops, mode)


def _module_gradients(dev, ops, mode):
    cdt = torch.float32 if mode == "f32" else torch.bfloat16
    err, tol = (rel_err, 2e-5) if mode == "f32" else (l2_err, None)
    BF16_TOL = {"frame_prior_network": 0.015, "pitch_predictor": 0.2, "diffusion": 0.1, "dec": 0.1, "conformer": 0.08,
                "enc": 0.1}

    def bound(name):
        if tol is not None:
            return tol
        for key, v in BF16_TOL.items():
            if key in name:
                return v
        raise KeyError(name)

    log = []  # (what, error): asserted together at the end so that a failure shows every number

    def chk(a, b, what="out"):
        log.append((what, err(a.float(), b)))

    # --- frame prior + pitch predictor -------------------------------------------------
    g = load_golden("variance_adaptor")
    m, sd = load(node("variance_adaptor"), key_shapes(g["keys"]), 60, dev, TAME, TAME_OFF)
    _zero_dropout(m)
    B, Tf = 3, 40
    flen = torch.tensor([40, 29, 7])
    fm = R.sequence_mask(flen, Tf).unsqueeze(1).float()
    x = rnd(1, B, 256, Tf, scale=0.5) * fm
    sdo = {("va." + k): v.clone().requires_grad_() for k, v in sd.items() if v.is_floating_point()}
    for fn_o, fn_p, dy, names in (
        (lambda xo: R.frame_prior(sdo, "va.frame_prior_network", xo, fm),
         lambda xc: m.frame_prior_network.forward_cl(xc, flen.to(dev).int()), rnd(2, B, 256, Tf),
         ["va.frame_prior_network.convs.0.weight", "va.frame_prior_network.convs.5.weight", "va.frame_prior_network.convs.3.bias",
          "va.frame_prior_network.norms.2.gamma", "va.frame_prior_network.norm_emb.beta"]),
        (lambda xo: R.pitch_predictor(sdo, "va.pitch_predictor", xo, fm),
         lambda xc: m.pitch_predictor.cl(xc, flen.to(dev).int()), rnd(3, B, 2, Tf),
         ["va.pitch_predictor.layers.0.conv.weight", "va.pitch_predictor.layers.4.conv.weight", "va.pitch_predictor.layers.2.norm.gamma",
          "va.pitch_predictor.out_layer.weight", "va.pitch_predictor.out_layer.bias"]),
    ):
        m.zero_grad()
        xo = x.clone().requires_grad_()
        yo = fn_o(xo)
        gro = torch.autograd.grad(yo, [xo] + [sdo[n] for n in names], dy)
        xc = ops.bct_to_btc(x.to(dev), cdt).requires_grad_()
        yc = fn_p(xc)
        chk(yc.detach().cpu().transpose(1, 2), yo.detach(), names[0].split(".")[1] + ":y")
        yc.backward(dy.transpose(1, 2).contiguous().to(dev).to(yc.dtype))
        chk(xc.grad.cpu().transpose(1, 2), gro[0], names[0].split(".")[1] + ":dx")
        _cmp_grads(m, "va.", sdo, names, gro[1:], tol, err, log)

    # --- diffusion decoder (DiffNet stack with the hand-written backward) ---------------
    g = load_golden("diffusion")
    md, sdd = load(node("decoder"), key_shapes(g["keys"]), 90, dev)
    md.train()
    sdo = {("dec." + k): (v.clone().requires_grad_() if v.is_floating_point() else v) for k, v in sdd.items()}
    co = g["cond"].transpose(1, 2).clone().requires_grad_()
    _, pred = R.diffusion_train(sdo, "dec", co, g["mel"].transpose(1, 2), g["mask"], g["t"], g["noise"])
    dyp = rnd(5, *pred.shape)
    names = ["dec.denoise_fn.residual_layers.3.conditioner_projection.weight", "dec.denoise_fn.residual_layers.0.diffusion_projection.weight",
             "dec.denoise_fn.residual_layers.11.dilated_conv.weight", "dec.denoise_fn.residual_layers.19.output_projection.bias",
             "dec.denoise_fn.input_projection.weight", "dec.denoise_fn.mlp.0.weight", "dec.denoise_fn.skip_projection.weight"]
    gro = torch.autograd.grad(pred, [co] + [sdo[n] for n in names], dyp)
    cc = g["cond"].to(dev).to(cdt).requires_grad_()
    md.injected = {"t": g["t"], "noise": g["noise"]}
    _, pred2 = md.forward_cl(cc, g["mel"].to(dev), g["mask"].sum(dim=(1, 2)).int().to(dev))
    pred2.backward(dyp.transpose(1, 2).contiguous().to(dev).to(pred2.dtype))
    chk(cc.grad.cpu(), gro[0].transpose(1, 2), "diffusion:dcond")
    _cmp_grads(md, "dec.", sdo, names, gro[1:], tol, err, log)

    # --- Conformer (attention backward, BatchNorm batch statistics) ---------------------
    g = load_golden("conformer")
    mc, sdc = load(node("encoder"), key_shapes(g["keys_new"]), 40, dev)
    _zero_dropout(mc)
    sdo = {("enc." + k): (v.clone().requires_grad_() if v.is_floating_point() else v) for k, v in sdc.items()}
    xo = g["x"].clone().requires_grad_()
    yo = R.conformer_encoder(sdo, "enc", xo, g["lens"], train_bn=True)
    dy = rnd(7, *yo.shape)
    names = ["enc.encoder.encoders.0.feed_forward_macaron.w_1.weight", "enc.encoder.encoders.1.self_attn.linear_pos.weight",
             "enc.encoder.encoders.3.self_attn.pos_bias_u", "enc.encoder.encoders.2.self_attn.pos_bias_v",
             "enc.encoder.encoders.3.self_attn.linear_q.weight", "enc.encoder.encoders.0.self_attn.linear_v.bias",
             "enc.encoder.encoders.2.conv_module.depthwise_conv.weight", "enc.encoder.encoders.3.norm_final.weight",
             "enc.encoder.encoders.3.feed_forward.w_2.weight", "enc.encoder.encoders.3.self_attn.linear_out.weight"]
    gro = torch.autograd.grad(yo, [xo] + [sdo[n] for n in names], dy)
    xc = g["x"].to(dev).to(cdt).requires_grad_()
    lens = g["lens"].to(dev).int()
    mask = (torch.arange(xc.shape[1], device=dev)[None] < lens[:, None]).unsqueeze(-1).float()
    yc = mc.forward_cl(xc * 1.0, lens, mask)
    chk(yc.detach().cpu(), yo.detach(), "conformer:y")
    yc.backward(dy.to(dev).to(yc.dtype))
    chk(xc.grad.cpu(), gro[0], "conformer:dx")
    _cmp_grads(mc, "enc.", sdo, names, gro[1:], tol, err, log)
    print(mode, "module errors vs oracle:", [(n, float(f"{e:.3g}")) for n, e in log])
    bad = [(n, e) for n, e in log if not e < bound(n)]
    assert not bad, (bad, log)


@pytest.mark.parametrize("variant", ["new", "legacy"])
def test_model_infer_batch(variant, dev):
    gi = load_golden("model_infer")
    m, _ = _model(dev, variant)
    m.eval()
    B = gi["phon"].shape[0]
    Tf = int(gi[f"{variant}_flen_ref"].max())
    def noise_fn(i, shape):
        t = rnd(112, B, 80, Tf) if i < 0 else rnd(2000 + i, B, 80, Tf)
        return t.transpose(1, 2).contiguous().to(dev)
    mel, cf0, vuv, flen = m.infer_batch(gi["phon"].to(dev), gi["plen"].to(dev), reference_mel=gi["mel"].to(dev),
                                        ref_lengths=gi["flen_in"], return_f0=True, noise_fn=noise_fn)
    assert torch.equal(m.last_durations.cpu(), gi[f"{variant}_dur_ref"].squeeze(1))   # integer: bit exact
    assert torch.equal(flen.cpu(), gi[f"{variant}_flen_ref"])
    assert rel_err(cf0.cpu(), gi[f"{variant}_cf0_ref"]) < 1e-4
    assert rel_err(mel.cpu(), gi[f"{variant}_mel_ref"]) < 1e-3
    assert float(((mel.cpu() - gi[f"{variant}_mel_ref"]) ** 2).mean()) < 1e-3


class _inject:
    """Feed prepared draws to the RNG consumers of the style path (test-only): torch.randn_like
    (sample_style_emb, model.py:192) and Categorical.sample (mdn_sample_sigma_and_mu, mdn.py:239)."""

    def __init__(self, dev, style_noise=None, comp=None):
        self.dev, self.sn, self.comp = dev, style_noise, comp

    def __enter__(self):
        self.o = (torch.randn_like, torch.distributions.Categorical.sample)
        sn, comp, dev = self.sn, self.comp, self.dev
        if sn is not None:
            torch.randn_like = lambda t, *a, **k: sn.to(dev).reshape(t.shape).to(t.dtype)
        if comp is not None:
            torch.distributions.Categorical.sample = lambda self_, *a, **k: comp.to(dev).reshape(self_._batch_shape)
        return self

    def __exit__(self, *exc):
        torch.randn_like, torch.distributions.Categorical.sample = self.o
        return False


def test_model_infer_batch_prompt_branch(dev):
    """infer_batch(style_prompt=...) (model.py:261-325, prompt branch :279-288): BERT -> adaptor -> style MDN ->
    most probable component + injected noise; integer durations / frame lengths bit-exact, mel 1e-3."""
    gi = load_golden("model_infer")
    m, _ = _model(dev)
    m.eval()
    B = gi["phon"].shape[0]
    Tf = int(gi["prompt_flen"].max())

    def noise_fn(i, shape):
        t = rnd(114, B, 80, Tf) if i < 0 else rnd(3000 + i, B, 80, Tf)
        return t.transpose(1, 2).contiguous().to(dev)

    with _inject(dev, gi["style_noise"]):
        mel, cf0, vuv, flen = m.infer_batch(gi["phon"].to(dev), gi["plen"].to(dev), style_prompt=(gi["ids"].to(dev), gi["am"].to(dev)),
                                            use_max=True, noise_scale=0.5, return_f0=True, noise_fn=noise_fn)
    assert torch.equal(m.last_durations.cpu(), gi["prompt_dur"].squeeze(1))   # integer: bit exact
    assert torch.equal(flen.cpu().float(), gi["prompt_flen"].float())
    assert rel_err(mel.cpu(), gi["prompt_mel"]) < 1e-3
    assert float(((mel.cpu() - gi["prompt_mel"]) ** 2).mean()) < 1e-3


def test_model_infer_single_utterance_and_style_embeddings(dev):
    """infer() (model.py:198-259): prompt branch with use_max True and False (injected Categorical draw),
    reference-mel branch; generate_style_emb (model.py:327-344)."""
    g = load_golden("model_infer_single")
    m, _ = _model(dev)
    m.eval()
    x1 = g["phon"].to(dev)
    prompt = (g["ids"].to(dev), g["am"].to(dev))
    cases = [("prompt_max", 215, dict(style_prompt=prompt, use_max=True, noise_scale=0.5), None),
             ("ref", 216, dict(reference_mel=g["mel"].to(dev)), None),
             ("prompt_sample", 217, dict(style_prompt=prompt, use_max=False, noise_scale=0.7), g["comp"])]
    for tag, seed, kw, comp in cases:
        Tf = g[tag + "_mel"].shape[-1]

        def noise_fn(i, shape, seed=seed, Tf=Tf):
            t = rnd(seed, 1, 80, Tf) if i < 0 else rnd(seed * 100 + i, 1, 80, Tf)
            return t.transpose(1, 2).contiguous().to(dev)

        with _inject(dev, g["style_noise"], comp):
            mel, cf0, vuv = m.infer(x1, return_f0=True, noise_fn=noise_fn, **kw)
        assert mel.shape[-1] == Tf, tag    # sum of the integer durations: exact
        assert rel_err(mel.cpu(), g[tag + "_mel"]) < 1e-3, tag
        assert rel_err(cf0.cpu(), g[tag + "_cf0"]) < 1e-4 and rel_err(vuv.cpu(), g[tag + "_vuv"]) < 1e-4, tag
    with _inject(dev, g["style_noise"]):
        pe, re_ = m.generate_style_emb(prompt, g["mel"].to(dev), use_max=True, noise_scale=0.5)
    assert rel_err(pe.cpu(), g["gen_prompt_emb"]) < 1e-4 and rel_err(re_.cpu(), g["gen_ref_emb"]) < 1e-4


def test_variance_adaptor_energy_branch(dev):
    """The optional energy head (variance_adaptor.py:139-146): energy_predictor reads the frame-prior output
    BEFORE the pitch embedding is added; forward, infer_batch and infer against the reference."""
    import torch.nn as nn

    from promptttspp.modules.frame_prior import FramePriorNetwork
    from promptttspp.modules.variance_adaptor import MDNPredictor, Predictor, VarianceAdaptor

    g = load_golden("variance_adaptor_energy")
    va = VarianceAdaptor(
        duration_predictor=MDNPredictor(256, 1, 3, 0.5, 2, num_gaussians=4, detach=True, disable_amp=True),
        pitch_predictor=Predictor(256, 2, 5, 0.5, 5, detach=False), pitch_emb=nn.Conv1d(1, 256, 1),
        energy_predictor=Predictor(256, 1, 3, 0.5, 2, detach=False), energy_emb=nn.Conv1d(1, 256, 1),
        frame_prior_network=FramePriorNetwork(256, 256, 6, 17, 0.1))
    m, _ = load(va, key_shapes(g["keys"]), 65, dev, TAME, TAME_OFF)
    m.eval()
    Tp, Tf = g["x"].shape[-1], g["cf0"].shape[-1]
    pm = R.sequence_mask(g["plen"], Tp).unsqueeze(1).long().to(dev)
    fm = R.sequence_mask(g["flen"], Tf).unsqueeze(1).float().to(dev)
    with torch.no_grad():
        h, _, cf0p, vuvp, enp = m(g["x"].to(dev), pm, fm, g["dur"].to(dev), g["cf0"].to(dev), None, g["energy"].to(dev))
        assert rel_err(h.cpu(), g["h"]) < 5e-5 and rel_err(enp.cpu(), g["enp"]) < 5e-5
        assert rel_err(cf0p.cpu(), g["cf0p"]) < 5e-5 and rel_err(vuvp.cpu(), g["vuvp"]) < 5e-5
        hi, fmi, cf0i, vuvi = m.infer_batch(g["x"].to(dev), pm, return_f0=True)
        assert torch.equal(fmi.cpu(), g["fmi"])
        assert rel_err(hi.cpu(), g["hi"]) < 5e-5 and rel_err(cf0i.cpu(), g["cf0i"]) < 5e-5
        h1, _, cf01, _ = m.infer(g["x"][:1].to(dev), pm[:1], return_f0=True)
        assert rel_err(h1.cpu(), g["h1"]) < 5e-5 and rel_err(cf01.cpu(), g["cf01"]) < 5e-5


def test_model_bf16_train_step_is_close(dev):
    """bf16 storage / bf16 MFMA: losses within a few 1e-2 of the f32 reference."""
    from promptttspp_amd import config

    m, g = _model(dev)
    m.eval()
    with config.use_dtype(torch.bfloat16):
        m.decoder.injected = {"t": g["t"], "noise": g["noise"]}
        with torch.no_grad():
            out = m(_batch(g, dev))
    for k in ("loss", "dec", "dur", "cf0", "vuv", "style"):
        ref = float(g["ev_" + k])
        assert abs(float(out[k]) - ref) < 5e-2 * max(1.0, abs(ref)), (k, float(out[k]), ref)


def test_bf16_training_trajectory_tracks_f32(dev):
    """The benchmarked dtype held to the parity dtype over OPTIMISER STEPS, not just one forward: the same model, the same
    batch (tests/golden/model_forward.npz), dropout off, the diffusion step and noise injected, N steps of the fused
    clip + AdamW in f32 mode (whose first step is pinned to the reference by test_model_train_step_losses_and_grads) and in bf16
    mode.  Bounds (per loss term, relative to max(1, |loss|), measured 1.5-4x below them): total 3e-2, diffusion loss 3e-3,
    duration MDN 2.5e-1, log-F0 / V-UV 1.2e-1, style MDN 1e-3; the total loss decreases by the same amount within 15 %; the parameter update of the whole run agrees
    with the f32 run's to a relative L2 of 0.6 and a cosine of 0.8 (Adam normalises by |g|: an element whose gradient is rounding noise moves
    by +-lr in either run, so this bound is loose by construction -- the loss trajectory is the statement)."""
    from promptttspp_amd import config
    from promptttspp_amd import functional as PF
    from promptttspp_amd.optim import FusedAdamW
    from promptttspp_amd.parallel import FlatGradReducer

    N = 6
    keys = ("loss", "dec", "dur", "cf0", "vuv", "style")

    def run(dt):
        m, g = _model(dev)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            for a in ("dropout_rate", "positional_dropout_rate", "p_dropout", "p"):
                if isinstance(getattr(mod, a, None), float):
                    setattr(mod, a, 0.0)
        m.train()
        params = [p for p in m.parameters() if p.requires_grad]
        p0 = torch.cat([p.detach().flatten().clone() for p in params])
        traj = []
        with config.use_dtype(dt):
            PF.clear_caches()
            red = FlatGradReducer(params)
            opt = FusedAdamW(params, lr=2e-5, max_grad_norm=1.0)  # (small enough that the f32 trajectory itself is smooth)
            try:
                for _ in range(N):
                    red.zero_grad()
                    m.decoder.injected = {"t": g["t"], "noise": g["noise"]}
                    PF.manual_seed(11)
                    out = m(_batch(g, dev))
                    out["loss"].backward()
                    red.finish()
                    opt.step()
                    traj.append([float(out[k]) for k in keys])
            finally:
                PF.enable_direct_grads(False)
                PF.clear_caches()
        torch.cuda.synchronize()
        p1 = torch.cat([p.detach().flatten() for p in params])
        return np.asarray(traj), (p1 - p0).double().cpu()

    ref, dref = run(torch.float32)
    got, dgot = run(torch.bfloat16)
    assert np.isfinite(got).all() and ref[0, 0] > ref[-1, 0], "the f32 run must make progress"
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    print("per-step deviation of (loss, dec, dur, cf0, vuv, style):\n", np.array2string(err, precision=4), "\nf32:\n",
          np.array2string(ref, precision=4), "\nbf16:\n", np.array2string(got, precision=4))
    # measured: total 2.1e-2, decoder (diffusion) loss 8e-4, duration MDN 1.6e-1 (the two-layer predictor's NLL moves in
    # steps; its trajectory is a step ahead / behind rather than off), log-F0 4e-2, V/UV 3e-2, style MDN 1e-5.
    # Two equally valid bf16 roundings of the DiffNet pre-activation (conditioner slice rounded before / after it joins the
    # dilated conv; the second is CLOSER to f32 at step 0: 5.2e-5 vs 9.2e-5 on the diffusion loss) move the V/UV trajectory
    # between 2.6e-2 and 9.1e-2 at step 3: the small predictors' paths are only loosely determined, their bounds say so.
    # The total is the sum of the parts: its bound is what the parts' bounds allow (0.003 x 8 + 0.25 + 0.12 + 0.12 of ~11.4 = 4.5e-2;
    # round 6, losses as one fused launch: total 3.1e-2 with the duration NLL 2.2e-1 off at step 4, every other column as before).
    bounds = np.asarray([4.5e-2, 3e-3, 2.5e-1, 1.2e-1, 1.2e-1, 1e-3])
    assert (err.max(axis=0) < bounds).all(), (err.max(axis=0), bounds)
    drop_ref, drop_got = ref[0, 0] - ref[-1, 0], got[0, 0] - got[-1, 0]
    assert abs(drop_got - drop_ref) < 0.15 * abs(drop_ref), (drop_got, drop_ref)
    rel = float((dgot - dref).norm() / dref.norm())
    cos = float((dgot * dref).sum() / (dgot.norm() * dref.norm()))
    assert rel < 0.6 and cos > 0.8, (rel, cos)  # measured 0.46 / 0.89
    print(f"bf16 vs f32 over {N} steps: max loss-term deviation {err.max():.2e}, loss drop {drop_got:.4f} vs {drop_ref:.4f}, "
          f"update rel L2 {rel:.3f} cos {cos:.3f}")


def test_direct_gradient_accumulation_equals_autograd(dev):
    """Weight-gradient kernels accumulating straight into the flat gradient buffer
    (FlatGradReducer(direct=True)) give the gradients autograd's own accumulation gives:
    same step (dropout ON, same counter seed), every parameter, both compute dtypes."""
    from promptttspp_amd import config
    from promptttspp_amd import functional as PF
    from promptttspp_amd.parallel import FlatGradReducer

    # f32: element-wise 2e-5.  bf16: the step is not run-to-run bit-reproducible (f32 atomics in the
    # BatchNorm statistics can move a value across a bf16 rounding boundary, 0.4 % of that element), so
    # the bf16 pass bounds the relative L2 error of every gradient instead.
    for dt, tol in ((torch.float32, 2e-5), (torch.bfloat16, None)):
        m, g = _model(dev)
        m.train()
        params = [p for p in m.parameters() if p.requires_grad]

        def step():
            m.decoder.injected = {"t": g["t"], "noise": g["noise"]}  # consumed by each forward
            PF.manual_seed(7)
            torch.manual_seed(3)
            m(_batch(g, dev))["loss"].backward()

        with config.use_dtype(dt):
            PF.enable_direct_grads(False)
            step()
            ref = [p.grad.detach().clone() if p.grad is not None else None for p in params]
            for p in params:
                p.grad = None
            try:
                red = FlatGradReducer(params)  # lays p.grad out as views of one buffer, enables direct mode
                assert PF.direct_grads_enabled()
                from promptttspp_amd import ops

                # second pass under ops.pinned_stream (the trainer's / bench.py's step): the cached main-stream handle
                # must not keep side-stream launches (weight gradients under wgrad_stream, also its torch_ops mode)
                # on the main stream
                for pinned in (False, True):
                    red.zero_grad()
                    if pinned:
                        with ops.pinned_stream():
                            step()
                            red.finish()
                    else:
                        step()
                        red.finish()
                    n_checked = 0
                    for p, r in zip(params, ref):
                        if r is None:
                            assert float(p.grad.abs().max()) == 0.0
                            continue
                        scale = float(r.abs().max()) + 1e-12
                        # (attention key biases have a structurally zero gradient -- softmax is shift
                        # invariant -- so theirs is rounding noise ~1e-8: absolute floor)
                        if tol is None:
                            assert float((p.grad - r).norm()) <= 2e-2 * float(r.norm()) + 1e-6, (tuple(p.shape), scale, pinned)
                        else:
                            assert float((p.grad - r).abs().max()) <= tol * scale + 1e-6, (tuple(p.shape), scale, pinned)
                        n_checked += 1
                    assert n_checked > 300
            finally:
                PF.enable_direct_grads(False)


def test_trainer_one_epoch_and_app_synthesis(dev, tmp_path):
    """The drop-in entry points end to end on the GPU: TTSTrainer (conf/train.yaml, synthetic data,
    FusedAdamW, flat-gradient DP plumbing at world size 1) trains one epoch, writes the reference's
    checkpoint format, and app.synthesize runs infer -> low-pass F0 -> F0-aware BigVGAN."""
    import os
    import sys

    from promptttspp.trainers.tts import TTSTrainer
    from promptttspp_amd import config
    from promptttspp_amd import functional as PF
    from promptttspp_amd.hydra_lite import compose

    root = os.path.join(os.path.dirname(__file__), "..")
    conf = os.path.join(root, "egs", "proposed", "bin", "conf")
    cfg = compose(conf, "train", ["dataset=synthetic", "optimizer=fused_adamw", "dataset.train.num_utts=24",
                                  "dataset.valid.num_utts=4", "dataset.max_tokens=4000", "train.num_epochs=1",
                                  "train.num_workers=0", "train.batch_size=2", "train.save_interval=1",
                                  f"output_dir={tmp_path}"])
    try:
        TTSTrainer(cfg)._train(0, 0, 1)
        ck = torch.load(tmp_path / "ckpt" / "last.ckpt", map_location="cpu")
        assert set(ck) == {"epoch", "model", "optimizer", "lr_scheduler"} and ck["epoch"] == 1
        assert (tmp_path / "ckpt" / "epoch-1.ckpt").exists() and (tmp_path / "logs" / "loss.csv").exists()
        rows = open(tmp_path / "logs" / "loss.csv").read().strip().splitlines()
        assert len(rows) >= 2 and "train/loss" in rows[0]
        assert all(np.isfinite(float(x)) for x in rows[-1].split(",")[1:])

        sys.path.insert(0, root)
        import app

        demo = compose(conf, "demo", [])
        model, voc = app.load_model(demo.model, None, demo.vocoder, None, device=dev)
        model.load_state_dict(ck["model"])  # the trainer's checkpoint loads into the demo model
        ids = torch.randint(1, 80, (1, 12))
        bert_ids = torch.randint(1000, 2000, (1, 8), device=dev)
        wav, mel = app.synthesize(model, voc, ids, style_prompt=(bert_ids, torch.ones_like(bert_ids)))
        assert wav.dim() == 2 and wav.shape[1] == mel.shape[-1] * 240 and torch.isfinite(wav).all()
    finally:
        PF.enable_direct_grads(False)
        config.set_compute_dtype(torch.float32)


def test_trainer_resume_continues_from_checkpoint(dev, tmp_path):
    """n4, the resume path (reference trainers/tts.py:97-110): a second run with ``ckpt_path=last.ckpt`` restores model,
    optimizer (fused AdamW moments + step count) and Noam scheduler, starts at epoch 2, APPENDS to loss.csv, and its
    first update continues the bias-corrected schedule instead of restarting it -- checked against an uninterrupted
    two-epoch run of the same seed: same step counter, same learning rate, same checkpoint key set."""
    import os

    from promptttspp.trainers.tts import TTSTrainer
    from promptttspp_amd import config
    from promptttspp_amd import functional as PF
    from promptttspp_amd.hydra_lite import compose

    conf = os.path.join(os.path.dirname(__file__), "..", "egs", "proposed", "bin", "conf")
    base = ["dataset=synthetic", "optimizer=fused_adamw", "dataset.train.num_utts=16", "dataset.valid.num_utts=4",
            "dataset.max_tokens=4000", "train.num_workers=0", "train.batch_size=2", "train.save_interval=1"]
    a, b = tmp_path / "interrupted", tmp_path / "straight"
    try:
        TTSTrainer(compose(conf, "train", base + ["train.num_epochs=1", f"output_dir={a}"]))._train(0, 0, 1)
        ck1 = torch.load(a / "ckpt" / "last.ckpt", map_location="cpu")
        rows1 = open(a / "logs" / "loss.csv").read().strip().splitlines()
        TTSTrainer(compose(conf, "train", base + ["train.num_epochs=2", f"output_dir={a}",
                                                  f"ckpt_path={a / 'ckpt' / 'last.ckpt'}"]))._train(0, 0, 1)
        ck2 = torch.load(a / "ckpt" / "last.ckpt", map_location="cpu")
        rows2 = open(a / "logs" / "loss.csv").read().strip().splitlines()
        assert ck1["epoch"] == 1 and ck2["epoch"] == 2 and (a / "ckpt" / "epoch-2.ckpt").exists()
        assert rows2[: len(rows1)] == rows1 and len(rows2) > len(rows1)  # appended, header kept, nothing rewritten
        assert all(np.isfinite(float(x)) for x in rows2[-1].split(",")[1:])

        TTSTrainer(compose(conf, "train", base + ["train.num_epochs=2", f"output_dir={b}"]))._train(0, 0, 1)
        ref = torch.load(b / "ckpt" / "last.ckpt", map_location="cpu")
        assert set(ck2) == set(ref) and set(ck2["model"]) == set(ref["model"])

        def steps(ck):  # per-parameter step counters of the optimizer state (torch.optim.AdamW layout)
            return sorted({int(v["step"]) for v in ck["optimizer"]["state"].values() if "step" in v})

        assert steps(ck2) == steps(ref) and steps(ck2)[0] == 2 * steps(ck1)[0]  # resumed, not restarted
        assert ck2["lr_scheduler"]["last_epoch"] == ref["lr_scheduler"]["last_epoch"] == 2 * ck1["lr_scheduler"]["last_epoch"]
        assert ck2["lr_scheduler"]["_last_lr"] == ref["lr_scheduler"]["_last_lr"]
        # the restored moments were used: after one more epoch the weights moved away from the checkpoint, by a
        # distance comparable to the uninterrupted run's second epoch (a restart from zero moments at step 1 would
        # take bias-correction-inflated first steps)
        k = next(n for n, v in ck1["model"].items() if v.dtype.is_floating_point and v.numel() > 1000 and "bert" not in n)
        d_res = float((ck2["model"][k] - ck1["model"][k]).norm())
        ref1 = torch.load(b / "ckpt" / "epoch-1.ckpt", map_location="cpu")
        d_ref = float((ref["model"][k] - ref1["model"][k]).norm())
        assert d_res > 0 and 0.3 < d_res / d_ref < 3.0, (d_res, d_ref)
    finally:
        PF.enable_direct_grads(False)
        config.set_compute_dtype(torch.float32)


def test_bert_frozen_layers_on_hip_match_library_layers(dev):
    """The 11 frozen BERT layers on the HIP kernels (fused QKV GEMM, PLAIN attention, GELU epilogue,
    fused residual + LayerNorm) vs the same weights through the transformers modules: CLS embedding
    within 1e-4 in f32 (ragged attention masks), bf16 close, train mode (dropout sites on) finite."""
    from promptttspp_amd import config
    from promptttspp_amd.modules.prompt_encoder import BertWrapper

    torch.manual_seed(0)
    bw = BertWrapper("bert-base-uncased").to(dev).eval()
    ids = torch.randint(1000, 20000, (5, 17), device=dev)
    am = torch.ones_like(ids)
    for b, n in enumerate((17, 9, 12, 3, 16)):
        am[b, n:] = 0
    with torch.no_grad():
        bw.hip_frozen_layers = False
        ref = bw((ids, am), dev)
        bw.hip_frozen_layers = True
        out = bw((ids, am), dev)
        assert rel_err(out.cpu(), ref.cpu()) < 1e-4
        with config.use_dtype(torch.bfloat16):
            out16 = bw((ids, am), dev)
        assert rel_err(out16.cpu(), ref.cpu()) < 5e-2
    bw.train()
    out_t = bw((ids, am), dev)
    assert out_t.requires_grad and torch.isfinite(out_t).all()
    # (a plain .sum() of a LayerNorm output with gamma = 1 is identically 0: project on a fixed random vector)
    (out_t * rnd(9, *out_t.shape).to(dev)).sum().backward()
    g = bw.model.encoder.layer[-1].attention.self.query.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0


def test_bert_in_tree_matches_reference_golden_and_oracle_gradients(dev):
    """a16 on the HIP kernels end to end: the CLS state against the fixture generated from transformers'
    BertModel (tests/golden/bert.npz), and the gradients of the ONLY trainable part -- encoder.layer[-1].attention.*
    (modules/prompt_encoder.py:29-31) -- against torch autograd of the oracle (dropout off: the masks differ by
    construction); then, with dropout on, the backward's regenerated attention-probability mask is checked through the
    linearity of the step in the upstream gradient and the reproducibility under the same seed."""
    from test_oracle_golden_am import synth_sd

    from promptttspp_amd import functional as PF
    from promptttspp_amd.modules.prompt_encoder import BertWrapper

    g = load_golden("bert")
    bw = BertWrapper("bert-base-uncased")
    sd = synth_sd(key_shapes(g["keys"]), 80)
    missing, unexpected = bw.model.load_state_dict(sd, strict=False)
    assert not unexpected and all("position_ids" in k or "token_type_ids" in k or "pooler" in k for k in missing), missing
    bw = bw.to(dev).eval()
    ids, am = g["ids"].to(dev), g["am"].to(dev)
    with torch.no_grad():
        cls = bw((ids, am), dev)
    assert rel_err(cls.cpu(), g["cls"]) < 1e-3       # north_star tolerance
    assert rel_err(cls.cpu(), g["cls"]) < 1e-4       # held

    # gradients of the trainable attention block vs the oracle
    for m in bw.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    bw.train()
    names = [n for n, p in bw.model.named_parameters() if p.requires_grad]
    assert names and all(n.startswith("encoder.layer.11.attention.") for n in names) and len(names) == 10
    dy = rnd(11, *cls.shape)
    out = bw((ids, am), dev)
    (out * dy.to(dev)).sum().backward()
    sdo = {("b." + k): (v.clone().requires_grad_() if ("b." + k)[2:] in names else v) for k, v in sd.items()}
    ref = R.bert_cls(sdo, "b.", g["ids"], g["am"])
    assert rel_err(out.detach().cpu(), ref.detach()) < 1e-4
    gro = torch.autograd.grad(ref, [sdo["b." + n] for n in names], dy)
    P = dict(bw.model.named_parameters())
    for n, gr in zip(names, gro):
        if "key.bias" in n:  # structurally zero (softmax is shift invariant): rounding noise on both sides
            assert float(P[n].grad.abs().max()) < 1e-5
            continue
        # (gradients after a 12-layer f32 stack: the CPU oracle itself is pinned to the reference at 3e-3 here,
        #  tests/test_oracle_golden_am.py::test_model_forward_grads; measured on MI355X: <= 8e-4)
        assert rel_err(P[n].grad.cpu(), gr) < 3e-3, (n, rel_err(P[n].grad.cpu(), gr))

    # dropout on (BERT's 0.1 on hidden states and attention probabilities): same seed -> same step, bit for bit;
    # gradient linear in the upstream gradient (the backward regenerates exactly the forward's masks)
    for m in bw.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.1

    def step(scale):
        for p in bw.parameters():
            p.grad = None
        PF.manual_seed(123)
        o = bw((ids, am), dev)
        (o * (scale * dy).to(dev)).sum().backward()
        return o.detach().clone(), [P[n].grad.detach().clone() for n in names]

    o1, g1 = step(1.0)
    o2, g2 = step(1.0)
    o3, g3 = step(-2.0)
    assert torch.equal(o1, o2) and all(torch.equal(a, b) for a, b in zip(g1, g2))
    assert not torch.equal(o1, out.detach())  # dropout really is on
    for n, a, c in zip(names, g1, g3):
        if "key.bias" not in n:
            assert rel_err(c, -2.0 * a) < 1e-5, n


def test_plms_sampler_matches_reference(dev):
    """pndm_speedup (diffusion.py:223-277): 10 and 4 outer steps against the reference's sampler run with the same
    initial noise; and the constructor accepts the argument the reference's constructor refuses."""
    from promptttspp.modules.denoiser import DiffNet
    from promptttspp.modules.diffusion import GaussianDiffusion

    g = load_golden("diffusion_plms")
    m, _ = load(node("decoder"), key_shapes(g["keys"]), 90, dev)
    m.eval()
    for interval in (10, 25):
        m.pndm_speedup = interval
        with torch.no_grad():
            mel = m.inference_cl(g["cond"].to(dev), noise_fn=lambda i, s: g["x_init"].transpose(1, 2).contiguous().to(dev))
        assert rel_err(mel.cpu(), g[f"mel_{interval}"]) < 1e-3, interval
        assert rel_err(mel.cpu(), g[f"mel_{interval}"]) < 2e-4, interval
    m2 = GaussianDiffusion(in_dim=256, out_dim=80, norm_scale=6.0, pndm_speedup=10,
                           denoise_fn=DiffNet(in_dim=80, encoder_hidden_dim=256, residual_layers=2, residual_channels=256,
                                              kernel_size=3, dilation_cycle_length=2))
    assert m2.pndm_speedup == 10


@pytest.mark.parametrize("rel", [True, False])
def test_transformer_encoder_plugin(rel, dev):
    """modules/transformer.py::Transformer on the HIP kernels: eval output (with / without the per-layer conditioning
    g) against the fixture generated from the reference, train-mode gradients against the oracle's autograd, and the
    model accepts it in the encoder slot (model.py:95)."""
    from promptttspp.modules.transformer import Transformer

    g = load_golden("transformer")
    tag = "rel" if rel else "abs"
    m = Transformer(channels=256, num_head=2, num_layers=2, kernel_size=3, dropout=0.1, scale=4, window_size=4, use_rel=rel)
    m, sd = load(m, key_shapes(g[f"{tag}_keys"]), 510 + int(rel), dev, {"emb_rel_k": 0.5, "emb_rel_v": 0.5})
    m.eval()
    T = g["x"].shape[-1]
    mask = R.sequence_mask(g["lens"], T).unsqueeze(1).float().to(dev)
    with torch.no_grad():
        assert rel_err(m(g["x"].to(dev), mask).cpu(), g[f"{tag}_y"]) < 5e-5
        assert rel_err(m(g["x"].to(dev), mask, g=g["g"].to(dev)).cpu(), g[f"{tag}_yg"]) < 5e-5
    _zero_dropout(m)
    from promptttspp_amd import ops

    xc = ops.bct_to_btc(g["x"].to(dev), torch.float32).requires_grad_()
    y = m.forward_cl(xc, g["lens"].to(dev).int())
    y.backward(g["dy"].transpose(1, 2).contiguous().to(dev))
    assert rel_err(xc.grad.cpu().transpose(1, 2), g[f"{tag}_dx"]) < 5e-5
    P = dict(m.named_parameters())
    for key in [k for k in g if k.startswith(f"{tag}_g:")]:
        n = key[len(tag) + 3:]
        gr = P[n].grad.cpu()
        if gr.numel() > 70000:
            gr = gr.flatten()[:: max(1, gr.numel() // 4096)][:4096]
        assert rel_err(gr, g[key].reshape(gr.shape)) < 1e-4, n
    if rel:  # the encoder slot of the model takes it (model.py:95: ``self.encoder(x, phone_mask)``)
        from promptttspp_amd import hydra_lite as H

        cfg = H.load_node(os.path.join(CONF, "prompttts_mdn_v2_wo_erg_final.yaml"))
        cfg["encoder"] = {"_target_": "promptttspp.modules.transformer.Transformer", "channels": 256, "num_head": 2,
                          "num_layers": 2, "kernel_size": 3, "dropout": 0.1, "scale": 4, "window_size": 4, "use_rel": True}
        model = H.instantiate(cfg).to(dev).eval()
        phon = torch.randint(3, 89, (3, 17), device=dev)
        plen = torch.tensor([17, 9, 5], device=dev)
        with torch.no_grad():
            enc, _, pm = model._encode(phon * (torch.arange(17, device=dev)[None] < plen[:, None]), plen)
        assert enc.shape == (3, 17, 256) and torch.isfinite(enc.float()).all()
        assert float(enc.float()[1, 9:].abs().max()) == 0.0  # padded phones are zero, as in the reference (x * mask)


BF16_TRANSFORMER_BOUNDS = {"y": 0.01, "dx": 0.07, "grad": 0.10}  # ~1.6-2 x the measured relative L2 errors (see the test)


def test_transformer_plugin_bf16_one_hop_from_the_oracle(dev):
    """the windowed relative-position attention (ptpp_attention_win_fwd / _bwd inside
    modules/transformer.py::Transformer, reference modules/transformer.py:59-137) in the BENCHMARKED dtype, bf16, directly against
    the fixture generated from the reference (outputs) and the f32 oracle's autograd (gradients) -- not against this package's own
    f32 path.  Relative L2 error per tensor; measured on MI355X: y 0.5 %, dx 4.2 %, parameter gradients 0.5-6.2 % (the
    relative-position tables emb_rel_k / emb_rel_v 3.1-3.3 %, the first feed-forward conv 6.2 %), bounds ~1.6-2 x that."""
    from promptttspp.modules.transformer import Transformer
    from promptttspp_amd import config, ops

    g = load_golden("transformer")
    tag = "rel"
    m = Transformer(channels=256, num_head=2, num_layers=2, kernel_size=3, dropout=0.1, scale=4, window_size=4, use_rel=True)
    m, sd = load(m, key_shapes(g[f"{tag}_keys"]), 511, dev, {"emb_rel_k": 0.5, "emb_rel_v": 0.5})
    T = g["x"].shape[-1]
    log = []
    with config.use_dtype(torch.bfloat16):
        m.eval()
        mask = R.sequence_mask(g["lens"], T).unsqueeze(1).float().to(dev)
        with torch.no_grad():
            log.append(("y", "y", l2_err(m(g["x"].to(dev), mask).float().cpu(), g[f"{tag}_y"])))
        _zero_dropout(m)
        m.train()
        xc = ops.bct_to_btc(g["x"].to(dev), torch.bfloat16).requires_grad_()
        y = m.forward_cl(xc, g["lens"].to(dev).int())
        y.backward(g["dy"].transpose(1, 2).contiguous().to(dev).to(y.dtype))
    log.append(("dx", "dx", l2_err(xc.grad.float().cpu().transpose(1, 2), g[f"{tag}_dx"])))
    P = dict(m.named_parameters())
    for key in [k for k in g if k.startswith(f"{tag}_g:")]:
        n = key[len(tag) + 3:]
        gr = P[n].grad.float().cpu()
        if gr.numel() > 70000:
            gr = gr.flatten()[:: max(1, gr.numel() // 4096)][:4096]
        log.append(("grad", n, l2_err(gr, g[key].reshape(gr.shape))))
    print("bf16 transformer errors:", [(n, round(e, 4)) for _, n, e in log])
    bad = [(n, e) for kind, n, e in log if not e < BF16_TRANSFORMER_BOUNDS[kind]]
    assert not bad, (bad, log)


def test_sampler_fused_gate_and_graph_consistency(dev):
    """bf16 sampler: the fused gate epilogue (PTPP_ACT_GATE, interleaved weight rows) and the HIP-graph replay
    against the unfused / eager path on the same injected noise: same mel within bf16 rounding."""
    from promptttspp_amd import config
    from promptttspp_amd import functional as PF

    g = load_golden("diffusion")
    m, _ = load(node("decoder"), key_shapes(g["keys"]), 90, dev)
    m.eval()
    B, T = 3, 90
    cond = rnd(5, B, T, 256).to(dev)
    noise_fn = lambda i, s: rnd(2000 + i, *s).to(dev)  # noqa: E731
    outs = {}
    with config.use_dtype(torch.bfloat16), torch.no_grad():
        for name, fused, graph in (("ref", False, False), ("fused", True, False), ("fused+graph", True, True)):
            orig = PF.diffnet_fused_gate
            PF.diffnet_fused_gate = (lambda dt: dt == torch.bfloat16) if fused else (lambda dt: False)
            try:
                outs[name] = m.inference_cl(cond.bfloat16(), noise_fn, use_graph=graph).float().cpu()
            finally:
                PF.diffnet_fused_gate = orig
    assert torch.isfinite(outs["ref"]).all()
    assert rel_err(outs["fused"], outs["ref"]) < 3e-2
    assert rel_err(outs["fused+graph"], outs["fused"]) < 1e-6   # the graph replays the same kernels


@pytest.mark.parametrize("B,T", [(3, 90), (5, 333), (2, 64)])
def test_sampler_loop_with_the_fused_head_matches_the_plain_loop(dev, B, T):
    """The eager reverse loop with everything between two DiffNet stacks as ONE launch (ops.sampler_head: skip / output
    projections, the reverse update, the next step's input projection and first-layer input; GaussianDiffusion._inference_fused)
    against the plain loop's seven launches on the same injected noise, row counts that are / are not multiples of the 64-row
    tile.  Same operand order and rounding points; the conv kernels feed the MFMA's K slots in another order, so about one eps
    element in 10^4 per step rounds to the other bf16 neighbour (tests/test_hip_kernels.py::test_sampler_head_kernel) -- after
    100 steps the mels agree like two bf16 runs do, and are equal bit for bit when no such flip happens."""
    from promptttspp_amd import config
    from promptttspp_amd.modules.diffusion import GaussianDiffusion

    g = load_golden("diffusion")
    m, _ = load(node("decoder"), key_shapes(g["keys"]), 90, dev)
    m.eval()
    cond = rnd(5, B, T, 256).to(dev)
    noise_fn = lambda i, s: rnd(3000 + i, *s).to(dev)  # noqa: E731
    outs = {}
    with config.use_dtype(torch.bfloat16), torch.no_grad():
        m.use_graph = False  # (small shapes would replay a graph: this test is about the eager loop)
        for fused in (False, True):
            GaussianDiffusion.FUSED_LOOP = fused
            try:
                assert m._fused_loop_ok(cond.bfloat16()) == (False)  # the step-projection table exists only inside inference_cl
                outs[fused] = m.inference_cl(cond.bfloat16(), noise_fn).float().cpu()
            finally:
                GaussianDiffusion.FUSED_LOOP = True
    assert torch.isfinite(outs[True]).all() and float(outs[True].abs().max()) > 0
    assert rel_err(outs[True], outs[False]) < 1e-2, rel_err(outs[True], outs[False])


def test_sampler_split_over_two_streams_is_bit_identical(dev, monkeypatch):
    """Large batches: the two halves of the batch run their reverse loops on two streams (GaussianDiffusion._inference_split);
    per utterance the arithmetic is that of the unsplit loop, so the mel is equal bit for bit (bf16 and f32, odd batch).
    (Where the one-launch DiffNet layer serves, the product keeps the batch whole: the guard is lifted here so that the
    split path itself is what runs in bf16 too.)"""
    from promptttspp_amd import config
    from promptttspp_amd.modules.diffusion import GaussianDiffusion

    monkeypatch.setattr(GaussianDiffusion, "_one_launch_layers", lambda self, cond: False)

    g = load_golden("diffusion")
    m, _ = load(node("decoder"), key_shapes(g["keys"]), 90, dev)
    m.eval()
    B, T = 5, 300
    cond = rnd(7, B, T, 256).to(dev)
    noise_fn = lambda i, s: rnd(3000 + i, *s).to(dev)  # noqa: E731
    for dt in (torch.bfloat16, torch.float32):
        with config.use_dtype(dt), torch.no_grad():
            outs = []
            for split in (False, True):
                m.split_streams, m.split_min_rows = split, 1
                try:
                    outs.append(m.inference_cl(cond.to(dt), noise_fn).float().cpu())
                finally:
                    m.split_streams, m.split_min_rows = True, 8192
            assert torch.isfinite(outs[0]).all()
            assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())


def test_model_forward_bench_size_properties(dev):
    """One BASELINE-sized training batch (max_tokens = 30 000 synthetic LibriTTS-R utterances), eval mode, f32:
    the forward is bit-reproducible, and the five losses do not change when the batch is padded with extra
    phone / frame columns (every mask, the length regulator and the tile edge handling at full size)."""
    import sys

    import torch.nn.functional as F

    sys.path.insert(0, ROOT)
    import bench

    torch.manual_seed(0)
    model = bench.build_model(dev).eval()
    batch = bench.make_batches(0, 1, 1, 30000, dev)[0]
    phon, dur, plen, mel, cf0, vuv, energy, flen, prompt = batch
    B, Tf = mel.shape[0], mel.shape[2]
    g = torch.Generator().manual_seed(5)
    inj = {"t": torch.randint(0, 100, (B,), generator=g), "noise": torch.randn(B, 80, Tf, generator=g)}

    def losses(b, noise):
        model.decoder.injected = {"t": inj["t"], "noise": noise}
        with torch.no_grad():
            out = model(b)
        return {k: float(v) for k, v in out.items()}

    a = losses(batch, inj["noise"])
    assert all(np.isfinite(v) for v in a.values())
    assert losses(batch, inj["noise"]) == a                                  # bit-reproducible
    pp, pf = 9, 70                                                              # extra padded phones / frames
    padp = lambda x: F.pad(x, (0, pp))                                          # noqa: E731
    padf = lambda x: F.pad(x, (0, pf))                                          # noqa: E731
    padded = [padp(phon), padp(dur), plen, padf(mel), padf(cf0), padf(vuv), padf(energy), flen, prompt]
    b2 = losses(padded, padf(inj["noise"]))
    for k in a:
        assert abs(a[k] - b2[k]) <= 2e-5 * max(1.0, abs(a[k])), (k, a[k], b2[k])


def test_train_mode_forward_is_bit_reproducible_at_bench_size(dev, monkeypatch):
    """bf16, TRAIN mode (BatchNorm batch statistics of the Conformer conv modules and the reference encoder, dropout from the
    counter-based generator with a fixed seed): five forwards of one BASELINE-sized batch give the same five losses bit for bit.
    Round 3 summed the statistics' block totals into 32 replicas by f32 atomics in arrival order and repeated runs fell into
    two classes (2.7e-3 apart in the style embedding); with PTPP_BN_DET=1 every replica has one writer (csrc/bn_dw.hip
    det_grid: opt-in, it costs 1.6 ms per training step)."""
    import sys

    monkeypatch.setenv("PTPP_BN_DET", "1")

    sys.path.insert(0, ROOT)
    import bench
    from promptttspp_amd import config
    from promptttspp_amd import functional as PF

    torch.manual_seed(0)
    with config.use_dtype(torch.bfloat16):
        model = bench.build_model(dev).train()
        batch = bench.make_batches(0, 1, 1, 30000, dev)[0]
        B, Tf = batch[3].shape[0], batch[3].shape[2]
        g = torch.Generator().manual_seed(5)
        inj = {"t": torch.randint(0, 100, (B,), generator=g), "noise": torch.randn(B, 80, Tf, generator=g)}
        runs = []
        for _ in range(5):
            model.decoder.injected = dict(inj)
            PF.manual_seed(123)
            with torch.no_grad():
                out = model(batch)
            torch.cuda.synchronize()
            runs.append({k: float(v) for k, v in out.items()})
    assert all(np.isfinite(v) for v in runs[0].values())
    for r in runs[1:]:
        assert r == runs[0], (r, runs[0])


def _confdec_model(dev):
    """PromptTTSMDNDurCFG with the non-diffusion decoder branch (reference model.py:123-126): the final
    config with ``decoder`` = a 2-block ConformerEncoder and ``out_conv`` = Conv1d(256, 80, 1), weights
    from the same (seed, name) recipe as the fixture (oracle/gen_golden_am.py::model_conformer_decoder)."""
    from promptttspp_amd import hydra_lite as H

    cfg = H.load_node(os.path.join(CONF, "prompttts_mdn_v2_wo_erg_final.yaml"))
    dec = dict(cfg["encoder"])
    dec["num_blocks"] = 2
    cfg["decoder"] = dec
    cfg["out_conv"] = {"_target_": "torch.nn.Conv1d", "in_channels": 256, "out_channels": 80, "kernel_size": 1}
    g = load_golden("model_conformer_decoder")
    m, _ = load(H.instantiate(cfg), key_shapes(g["keys"]), 300, dev, TAME, TAME_OFF)
    return m, g


def test_conformer_decoder_branch_losses_grads_and_infer(dev):
    """the class's Conformer-decoder + out_conv branch against the reference's own outputs --
    eval losses, train-mode losses, gradients, and the (deterministic) infer_batch mel within 1e-3."""
    m, g = _confdec_model(dev)
    m.eval()
    batch = _batch(g, dev)
    with torch.no_grad():
        out = m(batch)
        mel, cf0, vuv, flen = m.infer_batch(g["phon"].to(dev), g["plen"].to(dev), reference_mel=g["mel"].to(dev),
                                            ref_lengths=g["flen"], return_f0=True)
    for k in ("loss", "dec", "dur", "cf0", "vuv", "style"):
        ref = float(g["ev_" + k])
        assert abs(float(out[k]) - ref) < 1e-4 * max(1.0, abs(ref)), k
    assert torch.equal(flen.cpu().long(), g["infer_flen"].long())  # integer frame lengths: bit-exact
    assert mel.shape == g["infer_mel"].shape
    assert rel_err(mel.float().cpu(), g["infer_mel"]) < 1e-3
    assert rel_err(cf0.float().cpu(), g["infer_cf0"]) < 1e-3

    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        for a in ("dropout_rate", "positional_dropout_rate", "p_dropout", "p"):
            if isinstance(getattr(mod, a, None), float):
                setattr(mod, a, 0.0)
    m.train()
    out = m(_batch(g, dev))
    for k in ("loss", "dec", "dur", "cf0", "vuv", "style"):
        ref = float(g["tr_" + k])
        assert abs(float(out[k]) - ref) < 1e-4 * max(1.0, abs(ref)), k
    out["loss"].backward()
    params = dict(m.named_parameters())
    for key in g:
        if key.startswith("g:"):
            gr = params[key[2:]].grad
            ref = g[key]
            got = gr.cpu() if gr.numel() <= 70000 else gr.flatten()[:: max(1, gr.numel() // 4096)][:4096].cpu()
            assert rel_err(got, ref) < 2e-3, key
    total = sum(float(p.grad.pow(2).sum()) for p in m.parameters() if p.grad is not None) ** 0.5
    assert abs(total - float(g["grad_norm"])) < 2e-3 * float(g["grad_norm"])
