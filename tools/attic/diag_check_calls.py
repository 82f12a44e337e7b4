"""Debug: run the full model fwd+bwd with every conv1d / conv1d_wgrad / layernorm_bwd /
epilogue_bwd call cross-checked against a torch reference computed from the SAME inputs."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
warnings.simplefilter("ignore")
import torch, torch.nn.functional as F
import test_hip_acoustic as T
from promptttspp_amd import config, ops
dev = torch.device("cuda:0")
config.set_compute_dtype(torch.float32)
which = sys.argv[1] if len(sys.argv) > 1 else "cf0"
m, g = T._model(dev)
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
    for a in ("dropout_rate", "positional_dropout_rate", "p_dropout", "p"):
        if isinstance(getattr(mod, a, None), float): setattr(mod, a, 0.0)
m.train()
def rel(a, b): return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
def mask_of(lengths, T_, dev): return (torch.arange(T_, device=dev)[None] < lengths[:, None]).unsqueeze(-1).float()
stats = {"conv": [0, 0], "wgrad": [0, 0], "lnb": [0, 0], "epi": [0, 0]}
o_conv, o_wg, o_lnb, o_epi = ops.conv1d, ops.conv1d_wgrad, ops.layernorm_bwd, ops.epilogue_bwd
def conv(x, wp, bias, cout, ks=1, dil=1, pad=0, act=None, lengths=None, in_mask=False, out_mask=False, res=None,
         out_scale=1.0, res2=None, res_scale=1.0, drop_p=0.0, drop_seed=0, out=None):
    y = o_conv(x, wp, bias, cout, ks=ks, dil=dil, pad=pad, act=act, lengths=lengths, in_mask=in_mask, out_mask=out_mask,
               res=res, out_scale=out_scale, res2=res2, res_scale=res_scale, drop_p=drop_p, drop_seed=drop_seed, out=out)
    B, T_, cin = x.shape
    w = wp[:, :, :cin].permute(0, 2, 1).float()
    xm = x.float()
    if in_mask: xm = xm * mask_of(lengths, T_, x.device)
    r = F.conv1d(xm.transpose(1, 2), w, bias, padding=pad, dilation=dil).transpose(1, 2)
    if act == "relu": r = torch.relu(r)
    r = r * out_scale
    if out_mask: r = r * mask_of(lengths, T_, x.device)
    if res is not None: r = r + res_scale * res.float()
    if res2 is not None: r = r + res2.float()
    e = rel(y, r); stats["conv"][0] += 1
    if e > 1e-4:
        stats["conv"][1] += 1; print("BAD conv", tuple(x.shape), cout, ks, dil, pad, act, in_mask, out_mask, res is not None, e)
    return y
def wgrad(x, dy, cin, cout, ks, dil, pad, lengths=None, in_mask=False, want_bias=True):
    dw, db = o_wg(x, dy, cin, cout, ks, dil, pad, lengths, in_mask, want_bias)
    B, T_, _ = x.shape
    xm = x[..., :cin].float()
    if in_mask: xm = xm * mask_of(lengths, T_, x.device)
    ref = torch.nn.grad.conv1d_weight(xm.transpose(1, 2).contiguous(), (cout, cin, ks), dy.float().transpose(1, 2).contiguous(), padding=pad, dilation=dil)
    e = rel(dw, ref); stats["wgrad"][0] += 1
    eb = rel(db, dy.float().sum((0, 1))) if db is not None else 0
    if e > 1e-4 or eb > 1e-4:
        stats["wgrad"][1] += 1; print("BAD wgrad", tuple(x.shape), tuple(dy.shape), dy.stride(), cin, cout, ks, dil, pad, in_mask, e, eb)
    return dw, db
def lnb(dy, xsum, gamma, mean, rstd, lengths=None, out_mask=False, z=None, act_in=None, drop_in=(0.0, 0), drop_out=(0.0, 0), want_dz=False):
    dsum, dz, dg, db = o_lnb(dy, xsum, gamma, mean, rstd, lengths, out_mask, z, act_in, drop_in, drop_out, want_dz)
    B, T_, C = dy.shape
    with torch.enable_grad():
        xs = xsum.float().detach().clone().requires_grad_()
        gg = gamma.detach().clone().requires_grad_()
        bb = torch.zeros_like(gg).requires_grad_()
        mu = xs.mean(-1, keepdim=True); var = ((xs - mu) ** 2).mean(-1, keepdim=True)
        eps = (1.0 / rstd.view(B, T_, 1) ** 2 - var.detach()).clamp_min(0)
        y = (xs - mu) / torch.sqrt(var + eps) * gg + bb
        d = dy.float()
        if out_mask: d = d * mask_of(lengths, T_, dy.device)
        r_x, r_g, r_b = torch.autograd.grad(y, (xs, gg, bb), d)
    if act_in == "gelu":
        with torch.enable_grad():
            zz = z.float().detach().clone().requires_grad_()
            (gz,) = torch.autograd.grad(F.gelu(zz), zz, r_x)
        ez = rel(dz, gz)
        if ez > 1e-4: print("BAD lnb dz", tuple(dy.shape), out_mask, ez, float(z.abs().max()))
    e = max(rel(dsum, r_x), rel(dg, r_g), rel(db, r_b)); stats["lnb"][0] += 1
    if e > 1e-4:
        stats["lnb"][1] += 1; print("BAD lnb", tuple(dy.shape), out_mask, act_in, rel(dsum, r_x), rel(dg, r_g), rel(db, r_b))
    return dsum, dz, dg, db
def epi(dy, y=None, lengths=None, scale=1.0, relu=False, out_mask=False, drop_p=0.0, seed=0):
    dz = o_epi(dy, y, lengths, scale, relu, out_mask, drop_p, seed)
    r = dy.float() * scale
    if out_mask: r = r * mask_of(lengths, dy.shape[1], dy.device)
    if relu: r = r * (y > 0).float()
    e = rel(dz, r); stats["epi"][0] += 1
    if e > 1e-5:
        stats["epi"][1] += 1; print("BAD epi", tuple(dy.shape), scale, relu, out_mask, e)
    return dz
ops.conv1d, ops.conv1d_wgrad, ops.layernorm_bwd, ops.epilogue_bwd = conv, wgrad, lnb, epi
m.decoder.injected = {"t": g["t"], "noise": g["noise"]}
out = m(T._batch(g, dev))
out[which].backward()
torch.cuda.synchronize()
print(stats)
