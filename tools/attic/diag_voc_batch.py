"""Diagnostic: BigVGAN batch independence / run-to-run determinism with and without the fused AMP-layer kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.fill import fill_state_dict
from promptttspp_amd.vocoders import BigVGAN
from promptttspp_amd import ops
from promptttspp_amd.layers.activations import AntiAliasActivation

dev = torch.device("cuda:0")
torch.manual_seed(3)
m = BigVGAN(80, 512, [6, 5, 4, 2], [12, 10, 8, 4], [3, 7, 11], [[1, 3, 5]] * 3)
fill_state_dict(m, seed=5, overrides={"weight_g": 0.4})
m = m.to(dev).eval().set_compute_dtype(torch.bfloat16)
x = torch.clamp(-5.5 + 2.1 * torch.randn(64, 80, 1000, device=dev), -11.5, 2.0)
for fuse in (False, True):
    m.fuse_amp_layers = fuse
    y = m(x); y2 = m(x)
    print("fuse", fuse, "repeat identical:", torch.equal(y, y2), "batch-independent:", [bool(torch.equal(m(x[b:b+1])[0], y[b])) for b in (0, 37, 63)])
taps = AntiAliasActivation(8).taps()
for C, T in ((32, 240000), (64, 120000)):
    for B in (64,):
        xx = torch.randn(B, T, C, device=dev).bfloat16()
        for ks, d in ((3, 1), (11, 5)):
            w = [ops.pack_conv_weight(torch.randn(C, C, ks, device=dev) / (C * ks) ** 0.5, torch.bfloat16) for _ in range(2)]
            b = [0.1 * torch.randn(C, device=dev) for _ in range(2)]
            la = [0.3 * torch.randn(C, device=dev) for _ in range(2)]
            f = lambda z: ops.amp_layer(z, w[0], b[0], w[1], b[1], la[0], la[1], taps, taps, ks, d)
            y = f(xx); y2 = f(xx)
            y1 = f(xx[5:6].contiguous())
            bad = (y[5] != y1[0]).nonzero()
            print(f"layer C={C} ks={ks} d={d}: repeat {torch.equal(y, y2)} single {torch.equal(y[5], y1[0])}", "first diffs (t, c):", bad[:4].tolist(), "n", len(bad))
