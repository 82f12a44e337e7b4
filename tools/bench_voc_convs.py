"""Micro-benchmark of the BigVGAN stage-3/4 convolutions (C <= 64, very long T) through the C ABI."""
import os
import sys

import torch

sys.path.insert(0, "/root/repo")
from promptttspp_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
print("PTPP_CONV_ALT", os.environ.get("PTPP_CONV_ALT"))
for (B, T, C, ks, dil) in [(64, 240000, 32, 3, 1), (64, 240000, 32, 7, 3), (64, 240000, 32, 11, 5), (64, 120000, 64, 3, 1),
                           (64, 120000, 64, 7, 3), (64, 120000, 64, 11, 5), (64, 30000, 128, 7, 3)]:
    x = torch.randn(B, T, C, device=dev).bfloat16()
    res = torch.randn(B, T, C, device=dev).bfloat16()
    w = torch.randn(C, C, ks, device=dev) * 0.05
    b = torch.zeros(C, device=dev)
    wp = ops.pack_conv_weight(w, torch.bfloat16)
    y = torch.empty_like(x)
    pad = dil * (ks - 1) // 2
    f = lambda: ops.conv1d(x, wp, b, C, ks=ks, dil=dil, pad=pad, res=res, out=y)  # noqa: E731
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        f()
    e.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(e) / 5 * 1e3
    gb = 3 * x.numel() * 2 / 1e9
    tf = 2.0 * B * T * C * C * ks / 1e12
    print(f"B={B} T={T} C={C} k={ks} d={dil}: {us:8.1f} us  {gb / us * 1e6 / 1e3:6.2f} TB/s (x+res+y)  {tf / us * 1e6:6.1f} TFLOP/s")
    del x, res, y
