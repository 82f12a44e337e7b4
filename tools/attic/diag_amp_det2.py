import os, sys
import torch
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptttspp_amd import ops
from promptttspp_amd.layers.activations import AntiAliasActivation
dev = torch.device("cuda:0")
taps = AntiAliasActivation(8).taps()
C, B, T, ks, d = 32, 1, 140000, 3, 1
xx = torch.randn(B, T, C, device=dev).bfloat16()
w = [ops.pack_conv_weight(torch.randn(C, C, ks, device=dev) / (C * ks) ** 0.5, torch.bfloat16) for _ in range(2)]
b = [0.1 * torch.randn(C, device=dev) for _ in range(2)]
la = [0.3 * torch.randn(C, device=dev) for _ in range(2)]
f = lambda z: ops.amp_layer(z, w[0], b[0], w[1], b[1], la[0], la[1], taps, taps, ks, d)
ys = [f(xx) for _ in range(6)]
for y in ys[1:]:
    bad = (ys[0] != y).nonzero()
    print("n", len(bad), "tiles", Counter((bad[:, 1] // 256).tolist()).most_common(5), "row-in-tile", sorted(Counter((bad[:, 1] % 256).tolist()).items())[:40], "channels", sorted(Counter(bad[:, 2].tolist()).items()))
