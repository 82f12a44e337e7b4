"""Micro-benchmark of the phone-level convolutions (Conformer FFN k = 9, few rows per utterance)."""
import os
import sys

import torch

sys.path.insert(0, "/root/repo")
from promptttspp_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
print("PTPP_NO_T64", os.environ.get("PTPP_NO_T64"))
for (B, T, cin, cout, ks) in [(19, 230, 256, 1024, 9), (19, 230, 1024, 256, 9), (19, 150, 256, 1024, 9), (19, 150, 1024, 256, 9), (32, 100, 1024, 256, 9), (32, 100, 256, 1024, 9), (60, 55, 1024, 256, 9), (19, 199, 1024, 256, 9),
                              (96, 44, 1024, 256, 9), (32, 100, 256, 256, 1), (32, 100, 256, 512, 1)]:
    x = torch.randn(B, T, cin, device=dev).bfloat16()
    w = torch.randn(cout, cin, ks, device=dev) * 0.02
    b = torch.zeros(cout, device=dev)
    wp = ops.pack_conv_weight(w, torch.bfloat16)
    y = torch.empty(B, T, cout, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.conv1d(x, wp, b, cout, ks=ks, pad=ks // 2, out=y)  # noqa: E731
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        f()
    e.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(e) / 20 * 1e3
    print(f"B={B} T={T} {cin}->{cout} k={ks}: {us:7.1f} us  {2.0 * B * T * cin * cout * ks / us / 1e6:6.1f} TFLOP/s")
