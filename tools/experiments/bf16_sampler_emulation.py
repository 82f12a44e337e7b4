"""CPU experiment: which bf16 roundings of the DiffNet sampler cost the mel accuracy?  The oracle's sampler (f32) is re-run
with bf16 rounding inserted at chosen points (what the HIP path does in bf16 mode) and compared with the all-f32 result.
  in : conv operands (weights, activations) rounded to bf16, f32 accumulation  (the MFMA inputs)
  h  : residual stream h stored in bf16 between layers
  o  : the 1x1 output projection's result rounded to bf16 before the residual / skip update
  a  : gate pre-activation and conditioner projection stored in bf16
Usage: python tools/experiments/bf16_sampler_emulation.py [steps]"""
import math, sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import key_shapes, load_golden
from test_oracle_golden_am import synth_sd
from oracle import ref_torch as R

torch.set_num_threads(16)
bf = lambda t: t.to(torch.bfloat16).float()

def conv(sd, name, x, flags, **kw):
    w, b = sd[name + ".weight"], sd.get(name + ".bias")
    if "in" in flags:
        w, x = bf(w), bf(x)
    return F.conv1d(x, w, b, **kw)

def lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))

def diffnet(sd, p, x, t, cond, flags, layers=20, cycle=4):
    C = sd[p + ".input_projection.weight"].shape[0]
    h = torch.relu(conv(sd, p + ".input_projection", x, flags))
    if "h" in flags: h = bf(h)
    half = C // 2
    freq = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
    e = t[:, None].float() * freq[None]
    e = torch.cat([e.sin(), e.cos()], dim=-1)
    e = lin(sd, p + ".mlp.0", e); e = e * torch.tanh(F.softplus(e)); e = lin(sd, p + ".mlp.2", e)
    skip = 0
    for i in range(layers):
        q = f"{p}.residual_layers.{i}"
        d = 2 ** (i % cycle)
        y = h + lin(sd, q + ".diffusion_projection", e)[:, :, None]
        if "h" in flags: y = bf(y)
        c = conv(sd, q + ".conditioner_projection", cond, flags)
        if "a" in flags: c = bf(c)
        y = conv(sd, q + ".dilated_conv", y, flags, padding=d, dilation=d) + c
        gate, filt = y.chunk(2, dim=1)
        g = torch.sigmoid(gate) * torch.tanh(filt)
        if "a" in flags: g = bf(g)
        y = conv(sd, q + ".output_projection", g, flags)
        if "o" in flags: y = bf(y)
        res, sk = y.chunk(2, dim=1)
        h = (h + res) / math.sqrt(2.0)
        if "h" in flags: h = bf(h)
        skip = skip + sk
    s = skip / math.sqrt(layers)
    h = torch.relu(conv(sd, p + ".skip_projection", s, flags))
    return conv(sd, p + ".output_projection", h, flags)

def sample(sd, p, cond, x_init, K, flags):
    g = lambda n, t: sd[f"{p}.{n}"][t][:, None, None]
    x = x_init; B = x.shape[0]
    for i in reversed(range(K)):
        t = torch.full((B,), i, dtype=torch.long)
        eps = diffnet(sd, p + ".denoise_fn", x, t, cond, flags)
        x0 = (g("sqrt_recip_alphas_cumprod", t) * x - g("sqrt_recipm1_alphas_cumprod", t) * eps).clamp(-1.0, 1.0)
        mean = g("posterior_mean_coef1", t) * x0 + g("posterior_mean_coef2", t) * x
        x = mean + (0.5 * g("posterior_log_variance_clipped", t)).exp() * noise[i] if i > 0 else mean
    return x * 6.0

K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
gd = load_golden("diffusion")
sd = synth_sd(key_shapes(gd["keys"]), 90, None, None)
sd = {("dec." + k): v for k, v in sd.items()}
for k, v in R.diffusion_schedule().items():
    sd["dec." + k] = v
cond = gd["cond"].transpose(1, 2) if gd["cond"].shape[-1] == 256 else gd["cond"]
x_init = gd["x_init"]
B, _, T = x_init.shape
rng = np.random.default_rng(0)
noise = [torch.from_numpy(rng.standard_normal((B, 80, T)).astype(np.float32)) for _ in range(K)]
print("B", B, "T", T, "K", K)
with torch.no_grad():
    ref = sample(sd, "dec", cond, x_init, K, set())
    for flags in ({"in"}, {"h"}, {"o"}, {"a"}, {"in", "a"}, {"in", "a", "o"}, {"in", "h", "o", "a"}):
        y = sample(sd, "dec", cond, x_init, K, flags)
        print(f"{'+'.join(sorted(flags)):12s} mel MSE vs f32 {float(((y - ref) ** 2).mean()):.3e}   max |diff| {float((y - ref).abs().max()):.3e}")
