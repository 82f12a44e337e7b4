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
va = m.variance_adaptor; fpn = va.frame_prior_network; pp = va.pitch_predictor
cap = {}
o_fp = fpn.forward_cl
def fp(x, lengths):
    cap["xlr"] = x; return o_fp(x, lengths)
fpn.forward_cl = fp
m.decoder.injected = {"t": g["t"], "noise": g["noise"]}
with torch.no_grad():
    m(T._batch(g, dev))
fpn.forward_cl = o_fp
x0 = cap["xlr"].detach()
flen = g["flen"].to(dev); Tf = x0.shape[1]; lens = flen.int()
fmask = (torch.arange(Tf, device=dev)[None] < flen[:, None]).float()
mk = fmask.unsqueeze(1)
Rnd = torch.randn(x0.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
sdf = {("f." + k): v.detach() for k, v in fpn.state_dict().items()}
sdp = {("p." + k): v.detach() for k, v in pp.state_dict().items()}
cf0 = g["cf0"].squeeze(1).to(dev)
def run(name, prod_fn, ref_fn):
    xp = x0.clone().requires_grad_(); lp = prod_fn(xp); lp.backward()
    xt = x0.clone().requires_grad_(); lt = ref_fn(xt); (gt,) = torch.autograd.grad(lt, xt)
    print(f"{name:34s} loss {float(lp):.5f} {float(lt):.5f}  dx err per utt", " ".join(f"{rel(xp.grad[b], gt[b]):.1e}" for b in range(3)))
# A: frame prior module + dense loss (all rows)
run("A fp(module)+dense(all rows)", lambda x: (o_fp(x, lens) * Rnd).sum(),
    lambda x: (R.frame_prior(sdf, "f", x.transpose(1, 2), mk).transpose(1, 2) * Rnd).sum())
# C: module + pitch + L1
run("C fp(module)+pitch+L1", lambda x: (pp.cl(o_fp(x, lens), lens)[..., 0] - cf0).abs().sum(),
    lambda x: (R.pitch_predictor(sdp, "p", R.frame_prior(sdf, "f", x.transpose(1, 2), mk), mk)[:, 0] - cf0).abs().sum())
# D: module + pitch + dense loss
R2 = torch.randn(3, Tf, 2, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
run("D fp(module)+pitch+dense", lambda x: (pp.cl(o_fp(x, lens), lens) * R2).sum(),
    lambda x: (R.pitch_predictor(sdp, "p", R.frame_prior(sdf, "f", x.transpose(1, 2), mk), mk).transpose(1, 2) * R2).sum())
# E: module + k pitch layers + dense
for k in range(1, 6):
    R3 = torch.randn(3, Tf, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    def pk(x):
        h = o_fp(x, lens)
        for i in range(k): h = pp.layers[i].cl(h, lens)
        return (h * R3).sum()
    def rk(x):
        h = R.frame_prior(sdf, "f", x.transpose(1, 2), mk)
        h = R.predictor_layers(sdp, "p", h, mk, k, 5)
        return (h.transpose(1, 2) * R3).sum()
    run(f"E fp(module)+{k} pitch layers", pk, rk)
# F: pitch layers only from h
h0 = o_fp(x0, lens).detach()
def runh(name, prod_fn, ref_fn):
    xp = h0.clone().requires_grad_(); lp = prod_fn(xp); lp.backward()
    xt = h0.clone().requires_grad_(); lt = ref_fn(xt); (gt,) = torch.autograd.grad(lt, xt)
    print(f"{name:34s} loss {float(lp):.5f} {float(lt):.5f}  dh err per utt", " ".join(f"{rel(xp.grad[b], gt[b]):.1e}" for b in range(3)),
          " padded-row grad (prod, ref):", float(xp.grad[2, 21:].abs().max()), float(gt[2, 21:].abs().max()))
runh("F 1 pitch layer from h", lambda h: (pp.layers[0].cl(h, lens) * R3).sum(),
     lambda h: (R.predictor_layers(sdp, "p", h.transpose(1, 2), mk, 1, 5).transpose(1, 2) * R3).sum())
# ---- kink hunt at pitch layer 4
with torch.no_grad():
    hp = o_fp(x0, lens)
    for i in range(4): hp = pp.layers[i].cl(hp, lens)
    c4 = pp.layers[4].conv
    yp = PF.conv1d(hp, c4.weight, c4.bias, ks=5, pad=2, act="relu")
    ht = R.frame_prior(sdf, "f", x0.transpose(1, 2), mk)
    ht = R.predictor_layers(sdp, "p", ht, mk, 4, 5)
    zt = F.conv1d(ht, sdp["p.layers.4.conv.weight"], sdp["p.layers.4.conv.bias"], padding=2)
    yt = torch.relu(zt).transpose(1, 2)
    flips = ((yp > 0) != (yt > 0))
    print("relu flips per utt", [int(flips[b].sum()) for b in range(3)], "fwd err", rel(yp, yt))
    zz = zt.transpose(1, 2)
    print("min |z| per utt", [float(zz[b, : int(flen[b])].abs().min()) for b in range(3)])
    act = (yt > 0).float().sum(-1)
    print("active channels per frame utt2:", act[2, :21].int().tolist())
    print("active channels per frame utt0:", act[0, :36].int().tolist())
    var = yt.var(-1, unbiased=False)
    print("row var utt2:", [f"{float(v):.1e}" for v in var[2, :21]])
# ---- intermediate gradients, fp(module) + 5 pitch layers
R3 = torch.randn(3, Tf, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
xp = x0.clone().requires_grad_()
hs = [o_fp(xp, lens)]
for i in range(5): hs.append(pp.layers[i].cl(hs[-1], lens))
for h in hs: h.retain_grad()
(hs[-1] * R3).sum().backward()
xt = x0.clone().requires_grad_()
ht = [R.frame_prior(sdf, "f", xt.transpose(1, 2), mk)]
for i in range(5):
    q = f"p.layers.{i}"
    y = torch.relu(F.conv1d(ht[-1], sdp[q + ".conv.weight"], sdp[q + ".conv.bias"], padding=2))
    ht.append(R.layer_norm_c(y, sdp[q + ".norm.gamma"], sdp[q + ".norm.beta"]) * mk)
for h in ht: h.retain_grad()
(ht[-1].transpose(1, 2) * R3).sum().backward()
for i in reversed(range(6)):
    gp, gt = hs[i].grad, ht[i].grad.transpose(1, 2)
    print("grad of h%d" % i, "fwd", f"{rel(hs[i], ht[i].transpose(1, 2)):.1e}", "err per utt", " ".join(f"{rel(gp[b], gt[b]):.1e}" for b in range(3)),
          "| utt2 valid-rows err", f"{rel(gp[2, :21], gt[2, :21]):.1e}", "padded rows max (prod, ref)", f"{float(gp[2, 21:].abs().max()):.2e} {float(gt[2, 21:].abs().max()):.2e}")
print("grad of x", " ".join(f"{rel(xp.grad[b], xt.grad[b]):.1e}" for b in range(3)))
# ---- layer 4 alone, step by step with TRUE weights
h4 = hs[4].detach()
L4 = pp.layers[4]
w, bb = L4.conv.weight.detach(), L4.conv.bias.detach()
gam, bet = L4.norm.gamma.detach().reshape(-1), L4.norm.beta.detach().reshape(-1)
# torch
a = h4.clone().requires_grad_()
zt = F.conv1d(a.transpose(1, 2), w, bb, padding=2).transpose(1, 2); zt.retain_grad()
yt = torch.relu(zt); yt.retain_grad()
ot = R.layer_norm_last(yt, gam, bet, 1e-5) * fmask.unsqueeze(-1)
(ot * R3).sum().backward()
# product pieces
yp = ops.conv1d(h4, ops.pack_conv_weight(w, torch.float32), bb.contiguous(), 256, ks=5, pad=2, act="relu")
op_, mean, rstd, _ = ops.layernorm_fwd(yp, gam.contiguous(), bet.contiguous(), 1e-5, lengths=lens, out_mask=True, save_stats=True)
print("L4 fwd conv", rel(yp, yt), "ln", rel(op_, ot))
dsum, _, dg, db = ops.layernorm_bwd(R3.contiguous(), yp, gam.contiguous(), mean, rstd, lengths=lens, out_mask=True)
print("L4 LN bwd dsum err per utt", " ".join(f"{rel(dsum[b], yt.grad[b]):.1e}" for b in range(3)))
dz = ops.epilogue_bwd(dsum, yp, None, 1.0, True, False, 0.0, 0)
print("L4 relu bwd err per utt", " ".join(f"{rel(dz[b], zt.grad[b]):.1e}" for b in range(3)))
dx = ops.conv1d(dz, ops.pack_conv_weight(w, torch.float32, mode=1), None, 256, ks=5, pad=2)
print("L4 dgrad err per utt", " ".join(f"{rel(dx[b], a.grad[b]):.1e}" for b in range(3)))
dx2 = ops.conv1d(zt.grad.contiguous(), ops.pack_conv_weight(w, torch.float32, mode=1), None, 256, ks=5, pad=2)
print("L4 dgrad(from torch dz) err per utt", " ".join(f"{rel(dx2[b], a.grad[b]):.1e}" for b in range(3)))
# via the autograd Function
a2 = h4.clone().requires_grad_()
(L4.cl(a2, lens) * R3).sum().backward()
print("L4 Function path err per utt", " ".join(f"{rel(a2.grad[b], a.grad[b]):.1e}" for b in range(3)))
# ---- record the tensors entering layer 4's backward inside the chain
calls = []
o_lnb, o_epi, o_conv = ops.layernorm_bwd, ops.epilogue_bwd, ops.conv1d
def w_lnb(dy, xsum, gamma, mean, rstd, lengths=None, out_mask=False, **kw):
    out = o_lnb(dy, xsum, gamma, mean, rstd, lengths, out_mask, **kw)
    calls.append(("lnb", dy.clone(), xsum.clone(), mean.clone(), rstd.clone(), out[0].clone(), lengths, out_mask))
    return out
def w_epi(dy, y=None, lengths=None, scale=1.0, relu=False, out_mask=False, drop_p=0.0, seed=0):
    out = o_epi(dy, y, lengths, scale, relu, out_mask, drop_p, seed)
    calls.append(("epi", dy.clone(), y.clone() if y is not None else None, out.clone(), relu))
    return out
ops.layernorm_bwd, ops.epilogue_bwd = w_lnb, w_epi
xp = x0.clone().requires_grad_()
h = o_fp(xp, lens)
for i in range(5): h = pp.layers[i].cl(h, lens)
(h * R3).sum().backward()
ops.layernorm_bwd, ops.epilogue_bwd = o_lnb, o_epi
c0, c1 = calls[0], calls[1]
print("first bwd calls:", c0[0], c1[0])
print(" LN4 dy == R3:", rel(c0[1], R3), " xsum == yp:", rel(c0[2], yp), " mean/rstd same:", rel(c0[3], mean), rel(c0[4], rstd),
      " out_mask", c0[7], " lengths", None if c0[6] is None else c0[6].tolist())
print(" LN4 dsum vs isolated:", " ".join(f"{rel(c0[5][b], dsum[b]):.1e}" for b in range(3)))
print(" relu: dy == dsum_chain:", rel(c1[1], c0[5]), " y == yp:", rel(c1[2], yp), " dz vs isolated:", " ".join(f"{rel(c1[3][b], dz[b]):.1e}" for b in range(3)))
# ---- record the dgrad conv call of layer 4 in the chain
calls2 = []
def w_conv(x, wp, bias, cout, **kw):
    y = o_conv(x, wp, bias, cout, **kw)
    calls2.append((x.clone(), wp.clone(), y.clone(), dict(kw)))
    return y
xp = x0.clone().requires_grad_()
h = o_fp(xp, lens)
hh = []
for i in range(5):
    h = pp.layers[i].cl(h, lens); hh.append(h)
hh[3].retain_grad()
loss = (h * R3).sum()
ops.conv1d = w_conv
loss.backward()
ops.conv1d = o_conv
cx, cwp, cy, ckw = calls2[0]
print("dgrad call kw:", {k: (v if not torch.is_tensor(v) else v.tolist()) for k, v in ckw.items()})
print(" dz == isolated dz:", rel(cx, dz), " wp == fresh pack:", rel(cwp, ops.pack_conv_weight(w, torch.float32, mode=1)))
print(" dx_chain vs isolated dx per utt:", " ".join(f"{rel(cy[b], dx[b]):.1e}" for b in range(3)))
print(" h4.grad vs dx_chain per utt:", " ".join(f"{rel(hh[3].grad[b], cy[b]):.1e}" for b in range(3)))
# ---- torch chain vs torch isolated
print("h4 product vs torch-chain per utt:", " ".join(f"{rel(hs[4][b], ht[4].transpose(1,2)[b]):.1e}" for b in range(3)))
print("torch chain grad(h4) vs torch isolated grad per utt:", " ".join(f"{rel(ht[4].grad.transpose(1,2)[b], a.grad[b]):.1e}" for b in range(3)))
# torch isolated but starting from torch's own h4
a3 = ht[4].detach().transpose(1, 2).clone().requires_grad_()
z3 = F.conv1d(a3.transpose(1, 2), w, bb, padding=2).transpose(1, 2)
o3 = R.layer_norm_last(torch.relu(z3), gam, bet, 1e-5) * fmask.unsqueeze(-1)
(o3 * R3).sum().backward()
print("torch isolated(from torch h4) vs torch chain:", " ".join(f"{rel(a3.grad[b], ht[4].grad.transpose(1,2)[b]):.1e}" for b in range(3)))
print("R3 reuse check: R3 in chain section id", R3.shape, float(R3.sum()))
# ---- kink test: pre-activations of layer 4 from the two inputs
with torch.no_grad():
    zp_t = F.conv1d(h4.transpose(1, 2), w, bb, padding=2).transpose(1, 2)                  # torch conv on PRODUCT h4
    zt_t = F.conv1d(ht[4].detach(), w, bb, padding=2).transpose(1, 2)                       # torch conv on TORCH h4
    zp_k = ops.conv1d(h4, ops.pack_conv_weight(w, torch.float32), bb.contiguous(), 256, ks=5, pad=2)  # product kernel on product h4
    for name, za, zb in (("torch(h4p) vs torch(h4t)", zp_t, zt_t), ("kernel(h4p) vs torch(h4t)", zp_k, zt_t), ("kernel(h4p) vs torch(h4p)", zp_k, zp_t)):
        fl = (za > 0) != (zb > 0)
        print(name, "sign flips per utt", [int(fl[b].sum()) for b in range(3)], "max |dz|", float((za - zb).abs().max()))
        if fl.any():
            idx = fl.nonzero()[:5]
            for b_, t_, c_ in idx.tolist():
                print("   flip at", (b_, t_, c_), float(za[b_, t_, c_]), float(zb[b_, t_, c_]), "valid" if t_ < int(flen[b_]) else "padded")
