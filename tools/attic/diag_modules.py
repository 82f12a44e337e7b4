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

# ---- frame prior + pitch predictor
g = load_golden("variance_adaptor")
keys = key_shapes(g["keys"])
m, sd = T.load(T.node("variance_adaptor"), keys, 60, dev, TAME, TAME_OFF)
for mod in m.modules():
    for a in ("p_dropout", "p", "dropout_rate"):
        if isinstance(getattr(mod, a, None), float): setattr(mod, a, 0.0)
m.train()
B, Tf = 3, 40
flen = torch.tensor([40, 29, 7]); fm = R.sequence_mask(flen, Tf).unsqueeze(1).float()
x = rnd(1, B, 256, Tf, scale=0.5) * fm
dy = rnd(2, B, 256, Tf)
sdo = {("va."+k): v.clone().requires_grad_() for k, v in sd.items() if v.is_floating_point()}
xo = x.clone().requires_grad_()
yo = R.frame_prior(sdo, "va.frame_prior_network", xo, fm)
names = ["va.frame_prior_network.convs.0.weight", "va.frame_prior_network.convs.5.weight", "va.frame_prior_network.norms.2.gamma", "va.frame_prior_network.norm_emb.beta"]
gro = torch.autograd.grad(yo, [xo] + [sdo[n] for n in names], dy)
xc = ops.bct_to_btc(x.to(dev), torch.float32).requires_grad_()
yc = m.frame_prior_network.forward_cl(xc, flen.to(dev).int())
print("frame_prior fwd", rel_err(yc.detach().cpu().transpose(1,2), yo.detach()))
yc.backward(dy.transpose(1,2).contiguous().to(dev))
print("frame_prior dx", rel_err(xc.grad.cpu().transpose(1,2), gro[0]))
P = dict(m.named_parameters())
for n, gg in zip(names, gro[1:]):
    print("  ", n, rel_err(P[n[3:]].grad.cpu(), gg))
# pitch predictor
m.zero_grad()
xo = x.clone().requires_grad_()
yo = R.pitch_predictor(sdo, "va.pitch_predictor", xo, fm)
dy2 = rnd(3, B, 2, Tf)
names = ["va.pitch_predictor.layers.0.conv.weight", "va.pitch_predictor.layers.2.norm.gamma", "va.pitch_predictor.out_layer.weight"]
gro = torch.autograd.grad(yo, [xo] + [sdo[n] for n in names], dy2)
xc = ops.bct_to_btc(x.to(dev), torch.float32).requires_grad_()
yc = m.pitch_predictor.cl(xc, flen.to(dev).int())
print("pitch fwd", rel_err(yc.detach().cpu().transpose(1,2), yo.detach()))
yc.backward(dy2.transpose(1,2).contiguous().to(dev))
print("pitch dx", rel_err(xc.grad.cpu().transpose(1,2), gro[0]))
for n, gg in zip(names, gro[1:]):
    print("  ", n, rel_err(P[n[3:]].grad.cpu(), gg))

# ---- diffusion decoder: grad wrt cond
g = load_golden("diffusion")
md, sdd = T.load(T.node("decoder"), key_shapes(g["keys"]), 90, dev)
md.train()
sdo = {("dec."+k): v.clone().requires_grad_() if v.is_floating_point() else v for k, v in sdd.items()}
co = g["cond"].transpose(1,2).clone().requires_grad_()
nz, pred = R.diffusion_train(sdo, "dec", co, g["mel"].transpose(1,2), g["mask"], g["t"], g["noise"])
dyp = rnd(5, *pred.shape)
names = ["dec.denoise_fn.residual_layers.3.conditioner_projection.weight", "dec.denoise_fn.residual_layers.0.diffusion_projection.weight", "dec.denoise_fn.input_projection.weight", "dec.denoise_fn.residual_layers.19.output_projection.bias"]
gro = torch.autograd.grad(pred, [co] + [sdo[n] for n in names], dyp)
cc = g["cond"].to(dev).clone().requires_grad_()
md.injected = {"t": g["t"], "noise": g["noise"]}
lens = g["mask"].sum(dim=(1,2)).int().to(dev)
nz2, pred2 = md.forward_cl(cc, g["mel"].to(dev), lens)
print("diff fwd", rel_err(pred2.detach().cpu(), pred.detach().transpose(1,2)))
pred2.backward(dyp.transpose(1,2).contiguous().to(dev))
print("diff dcond", rel_err(cc.grad.cpu(), gro[0].transpose(1,2)))
P = dict(md.named_parameters())
for n, gg in zip(names, gro[1:]):
    print("  ", n, rel_err(P[n[4:]].grad.cpu(), gg))

# ---- conformer
g = load_golden("conformer")
mc, sdc = T.load(T.node("encoder"), key_shapes(g["keys_new"]), 40, dev)
for mod in mc.modules():
    for a in ("dropout_rate", "positional_dropout_rate"):
        if isinstance(getattr(mod, a, None), float): setattr(mod, a, 0.0)
mc.train()
sdo = {("enc."+k): v.clone().requires_grad_() if v.is_floating_point() else v for k, v in sdc.items()}
xo = g["x"].clone().requires_grad_()
yo = R.conformer_encoder(sdo, "enc", xo, g["lens"], train_bn=True)
dy = rnd(7, *yo.shape)
names = ["enc.encoder.encoders.0.feed_forward_macaron.w_1.weight", "enc.encoder.encoders.1.self_attn.linear_pos.weight", "enc.encoder.encoders.3.self_attn.pos_bias_u", "enc.encoder.encoders.3.self_attn.linear_q.weight", "enc.encoder.encoders.2.conv_module.depthwise_conv.weight", "enc.encoder.encoders.3.norm_final.weight", "enc.encoder.encoders.3.feed_forward.w_2.weight", "enc.encoder.encoders.3.self_attn.linear_out.weight"]
gro = torch.autograd.grad(yo, [xo] + [sdo[n] for n in names], dy)
xc = g["x"].to(dev).clone().requires_grad_()
lens = g["lens"].to(dev).int()
mask = (torch.arange(xc.shape[1], device=dev)[None] < lens[:, None]).unsqueeze(-1).float()
yc = mc.forward_cl(xc * 1.0, lens, mask)
print("conf fwd", rel_err(yc.detach().cpu(), yo.detach()))
yc.backward(dy.to(dev))
print("conf dx", rel_err(xc.grad.cpu(), gro[0]))
P = dict(mc.named_parameters())
for n, gg in zip(names, gro[1:]):
    print("  ", n, rel_err(P[n[4:]].grad.cpu(), gg))
