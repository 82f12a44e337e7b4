import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
warnings.simplefilter("ignore")
import numpy as np, torch
from conftest import load_golden, rel_err, key_shapes
import test_hip_acoustic as T
from test_oracle_golden_am import synth_sd, TAME, TAME_OFF
from oracle import ref_torch as R
from promptttspp_amd import config, ops
dev = torch.device("cuda:0")
config.set_compute_dtype(torch.float32)
def rnd(seed, *shape, scale=1.0):
    return torch.from_numpy((scale * np.random.default_rng(seed).standard_normal(shape)).astype(np.float32))
m, g = T._model(dev)
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
    for a in ("dropout_rate", "positional_dropout_rate", "p_dropout", "p"):
        if isinstance(getattr(mod, a, None), float): setattr(mod, a, 0.0)
m.train()
keys = key_shapes(g["keys"])
sd = synth_sd(keys, 100, TAME, TAME_OFF)
plen, flen, dur, mel, cf0, vuv = g["plen"], g["flen"], g["dur"], g["mel"], g["cf0"], g["vuv"]
B, Tp = g["phon"].shape; Tf = mel.shape[-1]
pm = R.sequence_mask(plen, Tp).unsqueeze(1); fm = R.sequence_mask(flen, Tf).unsqueeze(1).float()
x = rnd(1, B, 256, Tp, scale=0.5)
for which in ("lr_only", "fp", "pitch", "dec", "all"):
    xo = x.clone().requires_grad_()
    h = R.length_regulate(xo, dur.squeeze(1), pm.to(xo.dtype), fm)
    if which != "lr_only":
        h = R.frame_prior(sd, "variance_adaptor.frame_prior_network", h, fm)
    L = 0
    if which in ("lr_only", "fp"):
        L = (h * rnd(9, *h.shape)).sum()
    if which in ("pitch", "all"):
        pv = R.pitch_predictor(sd, "variance_adaptor.pitch_predictor", h, fm)
        L = L + (pv[:, 0:1] - cf0).abs().sum() + (pv[:, 1:2] - vuv).abs().sum()
    if which in ("dec", "all"):
        h2 = h + R._conv(sd, "variance_adaptor.pitch_emb", cf0) * fm
        nz, pred = R.diffusion_train(sd, "decoder", h2, mel, fm, g["t"], g["noise"])
        L = L + ((nz - pred) * fm).abs().sum()
    (gxo,) = torch.autograd.grad(L, xo)
    # product
    va = m.variance_adaptor
    xc = ops.bct_to_btc(x.to(dev), torch.float32).requires_grad_()
    fl = flen.to(dev).int(); fm1 = fm.transpose(1, 2).to(dev)
    from promptttspp_amd import functional as PF
    hc = PF.length_regulate(xc, dur.squeeze(1).to(dev), Tf)
    if which != "lr_only":
        hc = va.frame_prior_network.forward_cl(hc, fl)
    Lc = 0
    if which in ("lr_only", "fp"):
        Lc = (hc * rnd(9, B, 256, Tf).transpose(1, 2).to(dev)).sum()
    if which in ("pitch", "all"):
        pvc = va.pitch_predictor.cl(hc, fl)
        Lc = Lc + (pvc[..., 0] - cf0.squeeze(1).to(dev)).abs().sum() + (pvc[..., 1] - vuv.squeeze(1).to(dev)).abs().sum()
    if which in ("dec", "all"):
        h2c = hc + va._embed_scalar(va.pitch_emb, cf0.squeeze(1).to(dev), fm1, hc.dtype)
        m.decoder.injected = {"t": g["t"], "noise": g["noise"]}
        nzc, predc = m.decoder.forward_cl(h2c, mel.transpose(1, 2).contiguous().to(dev), fl)
        Lc = Lc + ((nzc - predc) * fm1).abs().sum()
    Lc.backward()
    print(which, "L", float(L), float(Lc), "dx err", rel_err(xc.grad.cpu().transpose(1, 2), gxo))
