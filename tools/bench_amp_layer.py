"""Micro-benchmark of the fused AMP-layer kernel (ptpp_amp_layer_fwd) on the BASELINE config-4 stage shapes
(64 x 10 s: C = 32 at T = 240 000, C = 64 at T = 120 000), every (kernel size, dilation) of the generator."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptttspp_amd import ops  # noqa: E402
from promptttspp_amd.layers.activations import AntiAliasActivation  # noqa: E402

dev = torch.device("cuda:0")
taps = AntiAliasActivation(8).taps()
B = int(os.environ.get("B", 64))
tot = 0.0
for C, T in ((32, 240000), (64, 120000)):
    x = torch.randn(B, T, C, device=dev).bfloat16()
    y = torch.empty_like(x)
    for ks in (3, 7, 11):
        for d in (1, 3, 5):
            w = [ops.pack_conv_weight(torch.randn(C, C, ks, device=dev) / (C * ks) ** 0.5, torch.bfloat16) for _ in range(2)]
            b = [0.1 * torch.randn(C, device=dev) for _ in range(2)]
            la = [0.3 * torch.randn(C, device=dev) for _ in range(2)]
            run = lambda: ops.amp_layer(x, w[0], b[0], w[1], b[1], la[0], la[1], taps, taps, ks, d, out=y)  # noqa: E731
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            tot += ms
            gb = 2 * x.numel() * 2 / 1e9
            fl = 2 * 2 * B * T * C * C * ks / 1e12
            print(f"C={C:3d} ks={ks:2d} d={d}: {ms:7.3f} ms  {gb / ms:6.2f} TB/s (x+y)  {fl / ms * 1e3:6.1f} TFLOP/s")
print(f"sum over the 18 layers of the two narrow stages: {tot:.2f} ms")
