import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
warnings.simplefilter("ignore")
import math, torch, torch.nn.functional as F
import test_hip_acoustic as T
from promptttspp_amd import config, ops
from promptttspp_amd import functional as PF
from oracle import ref_torch as R
dev = torch.device("cuda:0")
config.set_compute_dtype(torch.float32)
m, g = T._model(dev)
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
    for a in ("dropout_rate", "positional_dropout_rate", "p_dropout", "p"):
        if isinstance(getattr(mod, a, None), float): setattr(mod, a, 0.0)
m.train()
def rel(a, b): return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
va = m.variance_adaptor; fpn = va.frame_prior_network
cap = {}
o_fp = fpn.forward_cl
def fp(x, lengths):
    cap["xlr"] = x; return o_fp(x, lengths)
fpn.forward_cl = fp
m.decoder.injected = {"t": g["t"], "noise": g["noise"]}
with torch.no_grad():
    m(T._batch(g, dev))
x0 = cap["xlr"].detach()
flen = g["flen"].to(dev); Tf = x0.shape[1]; lens = flen.int()
fmask = (torch.arange(Tf, device=dev)[None] < flen[:, None]).float()
Rnd = torch.randn(x0.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
sd = {k: v.detach() for k, v in fpn.state_dict().items()}
for use_rand in (False, True):
    xin = torch.randn_like(x0) * fmask.unsqueeze(-1) if use_rand else x0
    for k in range(0, 7):
        # product
        xp = xin.clone().requires_grad_()
        h = fpn.norm_emb.forward_cl(fpn.embed.forward_cl(xp))
        for i in range(k):
            conv, norm = fpn.convs[i], fpn.norms[i]
            z = PF.conv1d(h, conv.weight, conv.bias, ks=17, pad=8, lengths=lens, in_mask=True)
            h = norm.forward_cl(z, res=h, act_in="gelu", lengths=lens, out_mask=False)
        lp = (h * Rnd * fmask.unsqueeze(-1)).sum()
        lp.backward()
        # torch
        xt = xin.clone().requires_grad_()
        mk = fmask.unsqueeze(1)
        y = xt.transpose(1, 2) * mk
        y = y * 16.0 + R.sinusoid(torch.arange(Tf), 256).t()[None].to(dev)
        y = R.layer_norm_c(y, sd["norm_emb.gamma"], sd["norm_emb.beta"])
        for i in range(k):
            r = F.gelu(F.conv1d(y * mk, sd[f"convs.{i}.weight"], sd[f"convs.{i}.bias"], padding=8))
            y = R.layer_norm_c(y + r, sd[f"norms.{i}.gamma"], sd[f"norms.{i}.beta"])
        lt = (y.transpose(1, 2) * Rnd * fmask.unsqueeze(-1)).sum()
        (gt,) = torch.autograd.grad(lt, xt)
        eb = [rel(xp.grad[b], gt[b]) for b in range(3)]
        print("rand" if use_rand else "real", "layers", k, "fwd", f"{rel(h, y.transpose(1,2)):.1e}", "dx err per utt", " ".join(f"{e:.1e}" for e in eb))
