"""TEST INFRASTRUCTURE ONLY -- golden vectors for the acoustic model
(build-container only; imports /root/reference).  See gen_golden.py.

    python oracle/gen_golden_am.py [name ...]
"""
import contextlib
import os
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.fill import fill_state_dict  # noqa: E402
from oracle.gen_golden import GENERATORS, _keys, _ref, _save, gen, rnd  # noqa: E402

GENERATORS.clear()

ENC_KW = dict(idim=256, attention_dim=256, attention_heads=2, linear_units=1024, num_blocks=4,
              positionwise_layer_type="conv1d", positionwise_conv_kernel_size=9, dropout_rate=0.2,
              pos_enc_layer_type="rel_pos", selfattention_layer_type="rel_selfattn", activation_type="swish",
              macaron_style=True, use_cnn_module=True, cnn_module_kernel=7, return_mask=False)

# keeps exp(mu + sigma^2/2) of the synthetic duration head in a sane range
TAME = {"duration_predictor.out_layer.mu.weight": 0.5, "duration_predictor.out_layer.log_sigma.weight": 0.1}
TAME_OFF = {"duration_predictor.out_layer.mu.bias": 1.3, "duration_predictor.out_layer.log_sigma.bias": -1.5}


def zero_dropout(m):
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0


@contextlib.contextmanager
def injected_rng(randn_list=None, rand_list=None, randn_like_list=None):
    """Replace torch.randn / torch.rand / torch.randn_like by queues of prepared tensors."""
    o = (torch.randn, torch.rand, torch.randn_like)
    qs = [list(randn_list or []), list(rand_list or []), list(randn_like_list or [])]

    def mk(q, name):
        def f(*a, **k):
            t = q.pop(0)
            return t
        return f

    inner = old_defaults = None
    if randn_list is not None:
        torch.randn = mk(qs[0], "randn")
        # p_sample binds `noise_fn=torch.randn` at definition time (diffusion.py:203):
        # re-point that default at the queue too, or the per-step noise is not injected
        import promptttspp.modules.diffusion as D

        inner = D.GaussianDiffusion.p_sample.__wrapped__
        old_defaults = inner.__defaults__
        inner.__defaults__ = (torch.randn,) + tuple(old_defaults[1:])
    if rand_list is not None:
        torch.rand = mk(qs[1], "rand")
    if randn_like_list is not None:
        torch.randn_like = mk(qs[2], "randn_like")
    try:
        yield
    finally:
        torch.randn, torch.rand, torch.randn_like = o
        if inner is not None:
            inner.__defaults__ = old_defaults


@gen
def conformer():
    _ref()
    from promptttspp.modules.esp import ConformerEncoder

    out = {}
    lens = torch.tensor([13, 9, 4])
    x = rnd(41, 3, 13, 256, scale=0.3)
    for variant in ("new", "legacy"):
        torch.manual_seed(0)
        m = ConformerEncoder(rel_pos_type=variant, **ENC_KW)
        fill_state_dict(m, seed=40)
        m.eval()
        with torch.no_grad():
            out[f"y_{variant}"] = m(x, lens)
        zero_dropout(m)
        m.train()
        with torch.no_grad():
            out[f"y_{variant}_trainbn"] = m(x, lens)
        out[f"keys_{variant}"] = _keys(m)
    _save("conformer", x=x, lens=lens, **out)


@gen
def mdn():
    _ref()
    from promptttspp.modules.mdn import MDNLayer, mdn_get_most_probable_sigma_and_mu, mdn_loss

    torch.manual_seed(0)
    m = MDNLayer(256, 1, 4, True)
    s = MDNLayer(256, 256, 10, True)
    fill_state_dict(m, seed=50)
    fill_state_dict(s, seed=51)
    x = rnd(52, 3, 11, 256)
    tgt = rnd(53, 3, 11, 1, scale=2.0)
    mask = (torch.arange(11)[None, :] < torch.tensor([11, 6, 2])[:, None])[:, :, None]
    with torch.no_grad():
        lp, ls, mu = m(x)
        loss_m = mdn_loss(lp.clone(), ls, mu, tgt, reduce=False, mask=mask)
        loss_u = mdn_loss(lp.clone(), ls, mu, tgt, reduce=False)
        sg, mm = mdn_get_most_probable_sigma_and_mu(lp, ls, mu)
        xs = rnd(54, 3, 1, 256)
        ts = rnd(55, 3, 1, 256, scale=0.1)
        lps, lss, mus = s(xs)
        loss_s = mdn_loss(lps, lss, mus, ts)
        sgs, mms = mdn_get_most_probable_sigma_and_mu(lps, lss, mus)
    _save("mdn", x=x, tgt=tgt, mask=mask, lp=lp, ls=ls, mu=mu, loss_m=loss_m, loss_u=loss_u, sg=sg, mm=mm,
          idx=lp.argmax(2), xs=xs, ts=ts, loss_s=loss_s, sgs=sgs, mms=mms, idxs=lps.argmax(2), keys_d=_keys(m), keys_s=_keys(s))


def build_va():
    from promptttspp.modules.frame_prior import FramePriorNetwork
    from promptttspp.modules.variance_adaptor import MDNPredictor, Predictor, VarianceAdaptor

    return VarianceAdaptor(
        duration_predictor=MDNPredictor(256, 1, 3, 0.5, 2, num_gaussians=4, detach=True, disable_amp=True),
        pitch_predictor=Predictor(256, 2, 5, 0.5, 5, detach=False),
        pitch_emb=nn.Conv1d(1, 256, 1),
        energy_predictor=None, energy_emb=None,
        frame_prior_network=FramePriorNetwork(256, 256, 6, 17, 0.1),
    )


@gen
def variance_adaptor():
    _ref()
    from promptttspp.utils.model import sequence_mask

    torch.manual_seed(0)
    va = build_va()
    fill_state_dict(va, seed=60, overrides=TAME, offsets=TAME_OFF)
    va.eval()
    plen = torch.tensor([10, 7, 3])
    Tp = 10
    pm = sequence_mask(plen, Tp).unsqueeze(1).long()
    x = rnd(61, 3, 256, Tp, scale=0.5) * pm
    dur = torch.from_numpy(np.random.default_rng(62).integers(1, 7, size=(3, 1, Tp))).float() * pm
    flen = dur.squeeze(1).sum(-1).long()
    Tf = int(flen.max())
    fm = sequence_mask(flen, Tf).unsqueeze(1).float()
    cf0 = (5.2 + 0.25 * rnd(63, 3, 1, Tf)) * fm
    with torch.no_grad():
        h, (lp, ls, mu), cf0p, vuvp, _ = va(x, pm, fm, dur.clone(), cf0, None, None)
        fp = va.frame_prior_network(rnd(64, 3, 256, Tf, scale=0.5), fm)
        hi, fmi, cf0i, vuvi = va.infer_batch(x, pm, return_f0=True)
        logd = va.duration_predictor.infer(x, pm)
        duri = (logd.exp().round().clamp_min(1).long() * pm)
    frac = (logd.exp() - logd.exp().floor() - 0.5).abs()
    assert float((frac * pm).masked_select(pm.bool()).min()) > 1e-3, "duration too close to a rounding boundary"
    _save("variance_adaptor", x=x, plen=plen, dur=dur, flen=flen, cf0=cf0, h=h, lp=lp, ls=ls, mu=mu, cf0p=cf0p,
          vuvp=vuvp, fp_x=rnd(64, 3, 256, Tf, scale=0.5), fp_y=fp, hi=hi, fmi=fmi, cf0i=cf0i, vuvi=vuvi, logd=logd,
          duri=duri, keys=_keys(va))


@gen
def variance_adaptor_energy():
    """The energy branch of the same class (variance_adaptor.py:139-146, :166-172, :196-202): not used by the
    wo_erg YAML but part of VarianceAdaptor's contract -- energy_predictor reads the frame-prior output
    BEFORE the pitch embedding is added."""
    _ref()
    from promptttspp.modules.frame_prior import FramePriorNetwork
    from promptttspp.modules.variance_adaptor import MDNPredictor, Predictor, VarianceAdaptor
    from promptttspp.utils.model import sequence_mask

    torch.manual_seed(0)
    va = VarianceAdaptor(
        duration_predictor=MDNPredictor(256, 1, 3, 0.5, 2, num_gaussians=4, detach=True, disable_amp=True),
        pitch_predictor=Predictor(256, 2, 5, 0.5, 5, detach=False),
        pitch_emb=nn.Conv1d(1, 256, 1),
        energy_predictor=Predictor(256, 1, 3, 0.5, 2, detach=False), energy_emb=nn.Conv1d(1, 256, 1),
        frame_prior_network=FramePriorNetwork(256, 256, 6, 17, 0.1),
    )
    fill_state_dict(va, seed=65, overrides=TAME, offsets=TAME_OFF)
    va.eval()
    plen = torch.tensor([9, 6, 2])
    Tp = 9
    pm = sequence_mask(plen, Tp).unsqueeze(1).long()
    x = rnd(66, 3, 256, Tp, scale=0.5) * pm
    dur = torch.from_numpy(np.random.default_rng(67).integers(1, 7, size=(3, 1, Tp))).float() * pm
    flen = dur.squeeze(1).sum(-1).long()
    Tf = int(flen.max())
    fm = sequence_mask(flen, Tf).unsqueeze(1).float()
    cf0 = (5.2 + 0.25 * rnd(68, 3, 1, Tf)) * fm
    energy = (0.3 * rnd(69, 3, 1, Tf)) * fm
    with torch.no_grad():
        h, _, cf0p, vuvp, enp = va(x, pm, fm, dur.clone(), cf0, None, energy)
        hi, fmi, cf0i, vuvi = va.infer_batch(x, pm, return_f0=True)
        h1, fm1, cf01, vuv1 = va.infer(x[:1], pm[:1], return_f0=True)
    _save("variance_adaptor_energy", x=x, plen=plen, dur=dur, flen=flen, cf0=cf0, energy=energy, h=h, cf0p=cf0p, vuvp=vuvp,
          enp=enp, hi=hi, fmi=fmi, cf0i=cf0i, vuvi=vuvi, h1=h1, cf01=cf01, keys=_keys(va))


@gen
def style_encoder():
    _ref()
    from promptttspp.modules.style_encoder import StyleEncoder

    torch.manual_seed(0)
    m = StyleEncoder(idim=80, gst_tokens=10, gst_heads=4, conv_layers=6, conv_chans_list=[128, 128, 256, 256, 512, 512],
                     conv_kernel_size=3, conv_stride=2, gru_layers=1, gru_units=256)
    fill_state_dict(m, seed=70)
    mel = rnd(71, 3, 80, 150)
    lens = torch.tensor([150, 97, 20])
    m.eval()
    with torch.no_grad():
        y = m(mel, lens)
    m.train()
    with torch.no_grad():
        yt = m(mel, lens)
    _save("style_encoder", mel=mel, lens=lens, y=y, y_trainbn=yt, keys=_keys(m))


def make_bert():
    from transformers import BertConfig, BertModel

    torch.manual_seed(0)
    b = BertModel(BertConfig())
    fill_state_dict(b, seed=80)
    return b.eval()


def prompt_ids(seed, B, Lmax):
    r = np.random.default_rng(seed)
    ids = np.zeros((B, Lmax), dtype=np.int64)
    am = np.zeros((B, Lmax), dtype=np.int64)
    for b in range(B):
        L = Lmax if b == 0 else int(r.integers(5, Lmax))
        ids[b, 0], ids[b, L - 1] = 101, 102
        ids[b, 1 : L - 1] = r.integers(1000, 30000, size=L - 2)
        am[b, :L] = 1
    return torch.from_numpy(ids), torch.from_numpy(am)


@gen
def bert():
    b = make_bert()
    ids, am = prompt_ids(81, 3, 14)
    with torch.no_grad():
        cls = b(input_ids=ids, attention_mask=am).last_hidden_state[:, 0]
    _save("bert", ids=ids, am=am, cls=cls, keys=_keys(b))


def build_model(variant="new", conformer_decoder=False):
    """Reference PromptTTSMDNDurCFG from explicit kwargs (conf/model/prompttts_mdn_v2_wo_erg_final.yaml)
    with BertWrapper replaced by a random-init HF BertModel taking (ids, mask).  ``conformer_decoder``: the
    non-diffusion decoder branch of the same class (model.py:123-126: Conformer on the frame sequence +
    out_conv), 2 blocks."""
    _ref()
    from promptttspp.layers.embedding import PhonemeEmbedding
    from promptttspp.models.prompttts_mdn_v2_final.model import PromptTTSMDNDurCFG
    from promptttspp.modules import prompt_encoder as PE
    from promptttspp.modules.denoiser import DiffNet
    from promptttspp.modules.diffusion import GaussianDiffusion
    from promptttspp.modules.esp import ConformerEncoder
    from promptttspp.modules.mdn import MDNLayer
    from promptttspp.modules.style_encoder import StyleEncoder

    class BW(nn.Module):
        def __init__(self, name=None):
            super().__init__()
            from transformers import BertConfig, BertModel

            self.model = BertModel(BertConfig())

        def forward(self, prompts, device):
            ids, am = prompts
            return self.model(input_ids=ids, attention_mask=am).last_hidden_state[:, 0, :]

    orig = PE.BertWrapper
    PE.BertWrapper = BW
    try:
        pe = PE.PromptEncoder("bert-base-uncased", 768, 512, 256)
    finally:
        PE.BertWrapper = orig
    torch.manual_seed(0)
    return PromptTTSMDNDurCFG(
        phoneme_embedding=PhonemeEmbedding(90, 256, do_scale=False, init_normal=False),
        encoder=ConformerEncoder(rel_pos_type=variant, **ENC_KW),
        variance_adaptor=build_va(),
        reference_encoder=StyleEncoder(idim=80, gst_tokens=10, gst_heads=4, conv_layers=6,
                                       conv_chans_list=[128, 128, 256, 256, 512, 512], conv_kernel_size=3,
                                       conv_stride=2, gru_layers=1, gru_units=256),
        prompt_encoder=pe,
        decoder=(ConformerEncoder(rel_pos_type=variant, **dict(ENC_KW, num_blocks=2)) if conformer_decoder else
                 GaussianDiffusion(in_dim=256, out_dim=80, norm_scale=6.0,
                                   denoise_fn=DiffNet(in_dim=80, encoder_hidden_dim=256, residual_layers=20,
                                                      residual_channels=256, kernel_size=3, dilation_cycle_length=4))),
        out_conv=nn.Conv1d(256, 80, 1) if conformer_decoder else None,
        style_mdn=MDNLayer(256, 256, 10, True),
        norm_style_emb=True, mdn_disable_amp=True,
    )


def synth_batch(seed, B=3, Tp=12):
    r = np.random.default_rng(seed)
    plen = torch.tensor([Tp, Tp - 4, 5][:B])
    pm = (torch.arange(Tp)[None, :] < plen[:, None])
    phon = torch.from_numpy(r.integers(3, 90, size=(B, Tp))) * pm
    dur = torch.from_numpy(r.integers(1, 8, size=(B, 1, Tp))).float() * pm[:, None, :]
    flen = dur.squeeze(1).sum(-1).long()
    Tf = int(flen.max())
    fm = (torch.arange(Tf)[None, :] < flen[:, None]).float()[:, None]
    mel = rnd(seed + 1, B, 80, Tf) * fm
    cf0 = (5.2 + 0.25 * rnd(seed + 2, B, 1, Tf)) * fm
    vuv = (rnd(seed + 3, B, 1, Tf) > -0.4).float() * fm
    energy = torch.zeros(B, 1, Tf)
    ids, am = prompt_ids(seed + 4, B, 14)
    return phon, dur, plen, mel, cf0, vuv, energy, flen, ids, am


@gen
def diffusion():
    _ref()
    m = build_model().decoder
    fill_state_dict(m, seed=90)
    m.eval()
    B, T = 2, 23
    cond = rnd(91, B, T, 256, scale=0.5)
    mel = rnd(92, B, T, 80)
    mask = (torch.arange(T)[None, :] < torch.tensor([23, 15])[:, None]).float()[:, None]
    t = torch.tensor([7, 93])
    noise = rnd(93, B, 80, T)
    with torch.no_grad():
        with injected_rng(randn_like_list=[noise]):
            orig_randint = torch.randint
            torch.randint = lambda *a, **k: t.clone()
            try:
                nz, pred = m(cond=cond, y=mel, mask=mask)
            finally:
                torch.randint = orig_randint
        eps = m.denoise_fn(rnd(94, B, 80, T), t, cond.transpose(1, 2), mask=mask)
        # sampler with injected noise (loop restated from GaussianDiffusion.inference, p_sample is the reference's)
        x = rnd(95, B, 80, T)
        x_init = x.clone()
        steps = [rnd(1000 + i, B, 80, T) for i in range(100)]
        c = cond.transpose(1, 2)
        for i in reversed(range(100)):
            x = m.p_sample(x, torch.full((B,), i, dtype=torch.long), c, noise_fn=lambda *s, device=None, _i=i: steps[_i])
        sampled = m._denorm(x.transpose(1, 2))
    bufs = {k: getattr(m, k) for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                                       "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
                                       "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
                                       "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2")}
    _save("diffusion", cond=cond, mel=mel, mask=mask, t=t, noise=noise, nz=nz, pred=pred, eps_x=rnd(94, B, 80, T), eps=eps,
          x_init=x_init, sampled=sampled, keys=_keys(m), **{"buf_" + k: v for k, v in bufs.items()})


@gen
def model_forward():
    m = build_model()
    fill_state_dict(m, seed=100, overrides=TAME, offsets=TAME_OFF)
    batch = synth_batch(101)
    phon, dur, plen, mel, cf0, vuv, energy, flen, ids, am = batch
    B, Tf = mel.shape[0], mel.shape[-1]
    t = torch.tensor([3, 55, 97])
    noise = rnd(105, B, 80, Tf)

    def run():
        orig_randint = torch.randint
        torch.randint = lambda *a, **k: t.clone()
        try:
            with injected_rng(randn_like_list=[noise]):
                return m([phon, dur.clone(), plen, mel, cf0, vuv, energy, flen, (ids, am)])
        finally:
            torch.randint = orig_randint

    m.eval()
    with torch.no_grad():
        ev = run()
    zero_dropout(m)
    m.train()
    # BERT dropout lives in HF modules (nn.Dropout too) -> already zeroed
    out = run()
    out["loss"].backward()
    grads = {}
    for name in ("encoder.encoder.encoders.0.feed_forward.w_1.weight", "encoder.encoder.encoders.3.self_attn.pos_bias_u",
                 "encoder.encoder.encoders.1.self_attn.linear_pos.weight", "encoder.encoder.encoders.2.conv_module.norm.weight",
                 "variance_adaptor.frame_prior_network.convs.2.weight", "variance_adaptor.pitch_predictor.layers.4.norm.gamma",
                 "variance_adaptor.duration_predictor.out_layer.mu.weight", "decoder.denoise_fn.residual_layers.7.dilated_conv.weight",
                 "decoder.denoise_fn.mlp.0.weight", "reference_encoder.ref_enc.gru.weight_hh_l0", "style_mdn.mu.bias",
                 "prompt_encoder.adaptor.2.weight", "phoneme_emb.emb.weight", "variance_adaptor.pitch_emb.weight",
                 "prompt_encoder.bert.model.encoder.layer.11.attention.self.query.weight"):
        g = dict(m.named_parameters())[name].grad
        grads["g:" + name] = g if g.numel() <= 70000 else g.flatten()[:: max(1, g.numel() // 4096)][:4096]
    total_sq = sum(float(p.grad.pow(2).sum()) for p in m.parameters() if p.grad is not None)
    _save("model_forward", phon=phon, dur=dur, plen=plen, mel=mel, cf0=cf0, vuv=vuv, flen=flen, ids=ids, am=am, t=t,
          noise=noise, **{"ev_" + k: v for k, v in ev.items()}, **{"tr_" + k: v.detach() for k, v in out.items()},
          grad_norm=np.sqrt(total_sq), keys=_keys(m), **grads)


@gen
def model_infer():
    out = {}
    for variant in ("new", "legacy"):
        m = build_model(variant)
        fill_state_dict(m, seed=100, overrides=TAME, offsets=TAME_OFF)
        m.eval()
        phon, dur, plen, mel, cf0, vuv, energy, flen, ids, am = synth_batch(111)
        with torch.no_grad():
            # pass 1 (throw-away noise) to learn Tf, then the real pass with injected noise of that shape
            from promptttspp.utils.model import sequence_mask
            pm = sequence_mask(plen).unsqueeze(1).to(phon.dtype)
            x = m.phoneme_emb(phon, pm)
            x = m.encoder(x.transpose(1, 2), plen).transpose(1, 2)
            se = torch.nn.functional.normalize(m.reference_encoder(mel, flen), dim=1)
            logd = m.variance_adaptor.duration_predictor.infer(x + se, pm)
            durs = logd.exp().round().clamp_min(1).long() * pm
            Tf = int(durs.squeeze(1).sum(-1).max())
            B = phon.shape[0]
            x_init = rnd(112, B, 80, Tf)
            steps = [rnd(2000 + i, B, 80, Tf) for i in range(100)]
            # reference draws: randn(shape) for x, then per step i=99..0 one randn(shape) (also at i==0)
            q = [x_init] + [steps[i] for i in reversed(range(100))]
            with injected_rng(randn_list=q):
                y, c0, vv, fl = m.infer_batch(phon, plen, reference_mel=mel, ref_lengths=flen, return_f0=True)
            out[f"{variant}_mel_ref"] = y
            out[f"{variant}_flen_ref"] = fl
            out[f"{variant}_dur_ref"] = durs
            out[f"{variant}_cf0_ref"] = c0
            out[f"{variant}_vuv_ref"] = vv
            if variant == "new":
                out["x_init"] = x_init
                # prompt path (use_max), style noise injected through randn_like
                sn = rnd(113, B, 1, 256)
                with injected_rng(randn_like_list=[sn]):
                    pe = torch.nn.functional.normalize(m.prompt_encoder((ids, am), x.device), dim=1)
                    st = m.sample_style_emb(*m.style_mdn(pe.transpose(-1, -2)), noise_scale=0.5, use_max=True)
                d2 = m.variance_adaptor.duration_predictor.infer(x + st, pm).exp().round().clamp_min(1).long() * pm
                Tf2 = int(d2.squeeze(1).sum(-1).max())
                x_init2 = rnd(114, B, 80, Tf2)
                steps2 = [rnd(3000 + i, B, 80, Tf2) for i in range(100)]
                out["prompt_dur"] = d2
                with injected_rng(randn_list=[x_init2] + [steps2[i] for i in reversed(range(100))], randn_like_list=[sn]):
                    y2, c02, vv2, fl2 = m.infer_batch(phon, plen, style_prompt=(ids, am), use_max=True, noise_scale=0.5,
                                                      return_f0=True)
                # Tf may differ between style paths; only record if equal shape else store own noise
                out["prompt_mel"] = y2
                out["prompt_flen"] = fl2
                out["style_noise"] = sn
        out["phon"], out["plen"], out["mel"], out["flen_in"], out["ids"], out["am"] = phon, plen, mel, flen, ids, am
    _save("model_infer", **out)


@gen
def model_infer_single():
    """The single-utterance entry points (model.py:198-262, :327-344): infer() on both style branches,
    use_max True and False (the Categorical draw is injected), and generate_style_emb()."""
    m = build_model("new")
    fill_state_dict(m, seed=100, overrides=TAME, offsets=TAME_OFF)
    m.eval()
    phon, dur, plen, mel, cf0, vuv, energy, flen, ids, am = synth_batch(211)
    x1, ids1, am1, mel1 = phon[:1], ids[:1], am[:1], mel[:1, :, : int(flen[0])]
    sn = rnd(213, 1, 1, 256)
    comp = torch.from_numpy(np.random.default_rng(214).integers(0, 10, size=(1, 256)))  # (B, C) component ids
    out = dict(phon=x1, ids=ids1, am=am1, mel=mel1, style_noise=sn, comp=comp)
    Cat = torch.distributions.Categorical
    orig_sample = Cat.sample

    def run(tag, noise_seed, **kw):
        # pass 1 with throw-away noise to learn Tf (durations do not depend on the diffusion noise)
        with injected_rng(randn_like_list=[sn] if "style_prompt" in kw else None):
            y0 = m.infer(x1, **kw)
        Tf = y0.shape[-1]
        x_init = rnd(noise_seed, 1, 80, Tf)
        steps = [rnd(noise_seed * 100 + i, 1, 80, Tf) for i in range(100)]
        with injected_rng(randn_list=[x_init] + [steps[i] for i in reversed(range(100))],
                          randn_like_list=[sn] if "style_prompt" in kw else None):
            y, c0, vv = m.infer(x1, return_f0=True, **kw)
        out[tag + "_mel"], out[tag + "_cf0"], out[tag + "_vuv"] = y, c0, vv

    with torch.no_grad():
        run("prompt_max", 215, style_prompt=(ids1, am1), use_max=True, noise_scale=0.5)
        run("ref", 216, reference_mel=mel1)
        Cat.sample = lambda self, *a, **k: comp.clone()
        try:
            run("prompt_sample", 217, style_prompt=(ids1, am1), use_max=False, noise_scale=0.7)
        finally:
            Cat.sample = orig_sample
        with injected_rng(randn_like_list=[sn]):
            pe, re_ = m.generate_style_emb((ids1, am1), mel1, use_max=True, noise_scale=0.5)
        out["gen_prompt_emb"], out["gen_ref_emb"] = pe, re_
    _save("model_infer_single", **out)


@gen
def transformer():
    """The FFT-block Transformer encoder plug-in (modules/transformer.py:226-263), with and without the windowed
    relative attention: eval output, and train-mode output + gradients with dropout zeroed."""
    _ref()
    from promptttspp.modules.transformer import Transformer
    from promptttspp.utils.model import sequence_mask

    out = {}
    lens = torch.tensor([19, 11, 4])
    T = 19
    mask = sequence_mask(lens, T).unsqueeze(1).float()
    x = rnd(501, 3, 256, T, scale=0.7) * mask
    gcond = rnd(502, 3, 256, 1, scale=0.3)
    dy = rnd(503, 3, 256, T)
    for tag, rel in (("rel", True), ("abs", False)):
        torch.manual_seed(0)
        m = Transformer(channels=256, num_head=2, num_layers=2, kernel_size=3, dropout=0.1, scale=4, window_size=4, use_rel=rel)
        fill_state_dict(m, seed=510 + int(rel), overrides={"emb_rel_k": 0.5, "emb_rel_v": 0.5})
        m.eval()
        with torch.no_grad():
            out[f"{tag}_y"] = m(x, mask)
            out[f"{tag}_yg"] = m(x, mask, g=gcond)
        zero_dropout(m)
        m.train()
        xx = x.clone().requires_grad_()
        y = m(xx, mask)
        y.backward(dy)
        out[f"{tag}_dx"] = xx.grad
        names = ["layers.0.ffn.ffn.conv1.weight", "layers.1.ffn.norm.gamma", "layers.1.attention.norm.beta",
                 "layers.0.ffn.ffn.conv2.bias"]
        names += (["layers.0.attention.attention_layer.emb_rel_k", "layers.1.attention.attention_layer.emb_rel_v",
                   "layers.0.attention.attention_layer.conv_q.weight", "layers.1.attention.attention_layer.conv_o.bias"]
                  if rel else ["layers.0.attention.attention_layer.qkv.weight", "layers.1.attention.attention_layer.out.weight"])
        P = dict(m.named_parameters())
        for n in names:
            gr = P[n].grad
            out[f"{tag}_g:{n}"] = gr if gr.numel() <= 70000 else gr.flatten()[:: max(1, gr.numel() // 4096)][:4096]
        out[f"{tag}_keys"] = _keys(m)
    _save("transformer", x=x, lens=lens, g=gcond, dy=dy, **out)


@gen
def diffusion_plms():
    """The PLMS sampler (diffusion.py:223-277, reached through inference() when pndm_speedup is set -- the
    constructor refuses the argument at :104-105, so the attribute is set on the built module)."""
    _ref()
    from collections import deque  # noqa: F401

    m = build_model().decoder
    fill_state_dict(m, seed=90)
    m.eval()
    B, T = 2, 23
    cond = rnd(91, B, T, 256, scale=0.5)
    x_init = rnd(95, B, 80, T)
    out = {}
    for interval in (10, 25):
        m.pndm_speedup = interval
        with torch.no_grad(), injected_rng(randn_list=[x_init]):
            out[f"mel_{interval}"] = m.inference(cond)
    _save("diffusion_plms", cond=cond, x_init=x_init, keys=_keys(m), **out)


@gen
def model_conformer_decoder():
    """The class's other decoder branch: losses (eval, train with dropout zeroed), gradients, and the
    deterministic infer_batch mel of the reference-mel path."""
    m = build_model(conformer_decoder=True)
    fill_state_dict(m, seed=300, overrides=TAME, offsets=TAME_OFF)
    phon, dur, plen, mel, cf0, vuv, energy, flen, ids, am = synth_batch(301)

    def run():
        return m([phon, dur.clone(), plen, mel, cf0, vuv, energy, flen, (ids, am)])

    m.eval()
    with torch.no_grad():
        ev = run()
        y, c0, vv, fl = m.infer_batch(phon, plen, reference_mel=mel, ref_lengths=flen, return_f0=True)
    zero_dropout(m)
    m.train()
    out = run()
    out["loss"].backward()
    grads = {}
    for name in ("decoder.encoder.encoders.1.feed_forward.w_2.weight", "decoder.encoder.encoders.0.self_attn.pos_bias_v",
                 "out_conv.weight", "out_conv.bias", "variance_adaptor.frame_prior_network.convs.2.weight",
                 "encoder.encoder.encoders.0.feed_forward.w_1.weight"):
        g = dict(m.named_parameters())[name].grad
        grads["g:" + name] = g if g.numel() <= 70000 else g.flatten()[:: max(1, g.numel() // 4096)][:4096]
    total_sq = sum(float(p.grad.pow(2).sum()) for p in m.parameters() if p.grad is not None)
    _save("model_conformer_decoder", phon=phon, dur=dur, plen=plen, mel=mel, cf0=cf0, vuv=vuv, flen=flen, ids=ids, am=am,
          **{"ev_" + k: v for k, v in ev.items()}, **{"tr_" + k: v.detach() for k, v in out.items()},
          grad_norm=np.sqrt(total_sq), keys=_keys(m), infer_mel=y, infer_flen=fl, infer_cf0=c0, infer_vuv=vv, **grads)


@gen
def host_logic():
    _ref()
    from promptttspp.datasets.utils import batch_by_size
    from promptttspp.utils.lr_scheduler import NoamLR

    r = np.random.default_rng(1234)
    n = 3000
    tp = np.clip(np.round(r.lognormal(np.log(62), 0.55, n)), 8, 260).astype(np.int64)
    frames = np.array([int((1 + r.poisson(7, k)).sum()) for k in tp], dtype=np.int64)
    order = np.argsort(frames, kind="stable")
    out = {"frames": frames}
    for mt in (10000, 30000):
        for W in (1, 2, 8):
            bs = batch_by_size(order, lambda i: int(frames[i]), max_tokens=mt * W, required_batch_size_multiple=W)
            flat = np.concatenate([np.asarray(b, dtype=np.int64) for b in bs])
            sizes = np.array([len(b) for b in bs], dtype=np.int64)
            out[f"flat_{mt}_{W}"] = flat
            out[f"sizes_{mt}_{W}"] = sizes
    p = nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-3)
    sch = NoamLR(opt, warmup_steps=4000)
    lrs = {}
    want = {1, 2, 100, 3999, 4000, 4001, 100000}
    for s in range(1, 100001):
        opt.step()
        sch.step()
        if s in want:
            lrs[s] = sch.get_last_lr()[0]
    out["noam_steps"] = np.array(sorted(lrs))
    out["noam_lr"] = np.array([lrs[k] for k in sorted(lrs)])
    _save("host_logic", **out)


@gen
def bigvgan_f0():
    _ref()
    from promptttspp.vocoders import F0AwareBigVGAN
    from oracle.gen_golden import BIGVGAN_KW, VOC_GAIN, mel_like

    torch.manual_seed(0)
    m = F0AwareBigVGAN(sampling_rate=24000, harmonic_num=8, **BIGVGAN_KW).eval()
    fill_state_dict(m, seed=120, overrides=VOC_GAIN)
    B, T = 2, 10
    x = mel_like(121, B, T)
    f0 = torch.tensor(np.abs(120 + 40 * np.random.default_rng(122).standard_normal((B, 1, T))).astype(np.float32))
    f0[0, 0, 3:5] = 0
    f0[1, 0, :2] = 0
    L = T * 240
    rand_ini = torch.from_numpy(np.random.default_rng(123).random((B, 9)).astype(np.float32))
    nz = rnd(124, B, L, 9)
    with torch.no_grad():
        # SineGen draws rand(B,9), randn_like(sine_waves); SourceModule draws randn_like(uv) (unused)
        with injected_rng(rand_list=[rand_ini.clone()], randn_like_list=[nz, torch.zeros(B, L, 1)]):
            y = m(x, f0)
        with injected_rng(rand_list=[rand_ini.clone()], randn_like_list=[nz, torch.zeros(B, L, 1)]):
            src, _, _ = m.m_source(m.f0_up(f0).transpose(-1, -2))
    _save("bigvgan_f0", x=x, f0=f0, rand_ini=rand_ini, nz=nz, y=y, src=src.transpose(1, 2), keys=_keys(m))


if __name__ == "__main__":
    names = sys.argv[1:] or list(GENERATORS)
    for n in names:
        GENERATORS[n]()
