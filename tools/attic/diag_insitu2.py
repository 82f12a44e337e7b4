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
va = m.variance_adaptor
cap = {}
o_fp = va.frame_prior_network.forward_cl
def fp(x, lengths):
    cap["xlr"] = x; x.retain_grad(); return o_fp(x, lengths)
va.frame_prior_network.forward_cl = fp
o_va = va.forward_cl
def vaf(x, *a, **k):
    cap["xenc"] = x; x.retain_grad(); return o_va(x, *a, **k)
va.forward_cl = vaf
m.decoder.injected = {"t": g["t"], "noise": g["noise"]}
out = m(T._batch(g, dev))
which = sys.argv[1] if len(sys.argv) > 1 else "cf0"
out[which].backward()
flen = g["flen"].to(dev); Tf = cap["xlr"].shape[1]
fm = (torch.arange(Tf, device=dev)[None] < flen[:, None]).float().unsqueeze(1)
sd = {("variance_adaptor." + k): v.detach().clone().requires_grad_() for k, v in va.state_dict().items() if v.is_floating_point()}
def tail(h):
    pv = R.pitch_predictor(sd, "variance_adaptor.pitch_predictor", h, fm)
    if which == "cf0": return (pv[:, 0] - g["cf0"].squeeze(1).to(dev)).abs().sum() / fm.sum()
    return (pv[:, 1] - g["vuv"].squeeze(1).to(dev)).abs().sum() / fm.sum()
# stage 1: from the frame prior input
x1 = cap["xlr"].detach().clone().requires_grad_()
l1 = tail(R.frame_prior(sd, "variance_adaptor.frame_prior_network", x1.transpose(1, 2), fm))
names = [k for k in sd if "frame_prior" in k]
gr = torch.autograd.grad(l1, [x1] + [sd[k] for k in names])
print("loss", float(l1), float(out[which]))
print("d x_lr", rel(cap["xlr"].grad, gr[0]))
err = (cap["xlr"].grad - gr[0]).abs().amax(-1)   # (B, T)
ref = gr[0].abs().amax(-1)
print("flen", g["flen"].tolist(), "Tf", Tf)
for b in range(err.shape[0]):
    print("b", b, " ".join(f"{float(e):.1e}" for e in err[b]))
    print("  ref", " ".join(f"{float(e):.1e}" for e in ref[b]))
# product recomputation from the captured frame-prior input (fresh graph)
x3 = cap["xlr"].detach().clone().requires_grad_()
h3 = o_fp(x3, flen.int())
pv3 = va.pitch_predictor.cl(h3, flen.int())
l3 = (pv3[..., 0] - g["cf0"].squeeze(1).to(dev)).abs().sum() / fm.sum()
l3.backward()
print("fresh product graph vs torch ref :", rel(x3.grad, gr[0]))
print("fresh product graph vs in-graph  :", rel(x3.grad, cap["xlr"].grad))
sys.exit(0)
P = dict(m.named_parameters())
print("fp params max err", max(rel(P[k].grad, a) for k, a in zip(names, gr[1:])))
# stage 2: from the encoder output (+style) through the length regulator
x2 = cap["xenc"].detach().clone().requires_grad_()
pm = (torch.arange(x2.shape[1], device=dev)[None] < g["plen"].to(dev)[:, None]).float().unsqueeze(1)
h2 = R.length_regulate(x2.transpose(1, 2), g["dur"].squeeze(1).to(dev), pm, fm)
print("LR fwd", rel(cap["xlr"].transpose(1, 2), h2))
l2 = tail(R.frame_prior(sd, "variance_adaptor.frame_prior_network", h2, fm))
(gx2,) = torch.autograd.grad(l2, x2)
print("d x_enc", rel(cap["xenc"].grad, gx2))
