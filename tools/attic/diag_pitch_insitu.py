import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
warnings.simplefilter("ignore")
import torch, torch.nn.functional as F
import test_hip_acoustic as T
from promptttspp_amd import config, ops
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
pp = m.variance_adaptor.pitch_predictor
cap = {}
orig = pp.cl
def cl(x, lengths):
    cap["h"] = x
    x.retain_grad()
    y = orig(x, lengths)
    cap["pv"] = y
    return y
pp.cl = cl
m.decoder.injected = {"t": g["t"], "noise": g["noise"]}
out = m(T._batch(g, dev))
out["cf0"].backward()
h = cap["h"]
print("h grad norm", float(h.grad.norm()))
# torch re-implementation of the pitch predictor on GPU, same parameters
sd = {k: v.detach().clone().requires_grad_() for k, v in pp.state_dict().items()}
hh = h.detach().clone().requires_grad_()
flen = g["flen"].to(dev); Tf = h.shape[1]
fm = (torch.arange(Tf, device=dev)[None] < flen[:, None]).float().unsqueeze(1)
pv = R.pitch_predictor({("p." + k): v for k, v in sd.items()}, "p", hh.transpose(1, 2), fm)
print("pv fwd", rel(cap["pv"].transpose(1, 2), pv))
loss = (pv[:, 0] - g["cf0"].squeeze(1).to(dev)).abs().sum() / fm.sum()
print("loss", float(loss), float(out["cf0"]))
grads = torch.autograd.grad(loss, [hh] + list(sd.values()))
print("dh", rel(h.grad, grads[0]))
P = dict(pp.named_parameters())
for (k, v), gr in zip(sd.items(), grads[1:]):
    print(f"  {rel(P[k].grad, gr):.3e} {k}")
# same torch re-implementation on the CPU, same h: is the function itself ill-conditioned?
sdc = {k: v.detach().cpu().clone().requires_grad_() for k, v in pp.state_dict().items()}
hc = h.detach().cpu().clone().requires_grad_()
pvc = R.pitch_predictor({("p." + k): v for k, v in sdc.items()}, "p", hc.transpose(1, 2), fm.cpu())
lc = (pvc[:, 0] - g["cf0"].squeeze(1)).abs().sum() / fm.sum().cpu()
gc = torch.autograd.grad(lc, [hc] + list(sdc.values()))
print("CPU vs GPU torch: pv", rel(pv.cpu(), pvc), "dh", rel(grads[0].cpu(), gc[0]))
for (k, v), a, b in zip(sd.items(), grads[1:], gc[1:]):
    print(f"  {rel(a.cpu(), b):.3e} {k}")
d = (pv[:, 0].cpu() - g["cf0"].squeeze(1)).abs()
print("min |pred-target| on valid frames", float(d[fm.cpu()[:, 0] > 0].min()))
sg = torch.sign(pv[:, 0].cpu() - g["cf0"].squeeze(1)); sc = torch.sign(pvc[:, 0] - g["cf0"].squeeze(1))
print("sign flips", int((sg != sc).sum()))
