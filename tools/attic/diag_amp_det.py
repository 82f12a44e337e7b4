import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptttspp_amd import ops
from promptttspp_amd.layers.activations import AntiAliasActivation
dev = torch.device("cuda:0")
taps = AntiAliasActivation(8).taps()
C = 32
for B, T in ((8, 60000), (32, 60000), (1, 70000), (1, 240000), (2, 240000)):
    xx = torch.randn(B, T, C, device=dev).bfloat16()
    ks, d = 3, 1
    w = [ops.pack_conv_weight(torch.randn(C, C, ks, device=dev) / (C * ks) ** 0.5, torch.bfloat16) for _ in range(2)]
    b = [0.1 * torch.randn(C, device=dev) for _ in range(2)]
    la = [0.3 * torch.randn(C, device=dev) for _ in range(2)]
    f = lambda z: ops.amp_layer(z, w[0], b[0], w[1], b[1], la[0], la[1], taps, taps, ks, d)
    ys = [f(xx) for _ in range(4)]
    nd = [(ys[0] != y).sum().item() for y in ys[1:]]
    dmax = max(float((ys[0].float() - y.float()).abs().max()) for y in ys[1:])
    bad = (ys[0] != ys[1]).nonzero()
    print(os.environ.get("PTPP_AMP_VARIANT", "default"), f"B={B} T={T}: differing elements per repeat {nd} max|diff| {dmax:.3g}", "min/max differing row:", (int(bad[:, 1].min()), int(bad[:, 1].max())) if len(bad) else None, "batches", sorted(set(bad[:, 0].tolist()))[:4])
