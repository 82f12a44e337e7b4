"""The DiffNet output projection's data gradient with the gate backward fused (512 -> 256, 1 x 1): tile kernel
(ptpp_conv1d_gate_bwd) against the row-tile engine (ptpp_conv1d_rt_gate_bwd) at the training bucket shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptttspp_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, T, C = 19, 1550, 256
torch.manual_seed(0)
do = torch.randn(B, T, 2 * C, device=dev).bfloat16()
a = torch.randn(B, T, 2 * C, device=dev).bfloat16()
w = torch.randn(2 * C, C, 1, device=dev) * 0.04
da = torch.empty(B, T, 20 * 2 * C, device=dev, dtype=torch.bfloat16)
wp, ws = ops.pack_conv_weight(w, torch.bfloat16, 1), ops.pack_conv_weight(w, torch.bfloat16, 4)


def timeit(f, n=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


old = lambda: ops.conv1d_gate_bwd(do, wp, a, da[:, :, 2 * C:4 * C])
new = lambda: ops.conv1d_rt_gate_bwd(do, ws, a, da[:, :, 2 * C:4 * C])
t = {"old": [], "new": []}
for _ in range(5):
    t["old"].append(timeit(old))
    t["new"].append(timeit(new))
fl = 2.0 * B * T * 2 * C * C
by = B * T * (2 * C + 2 * C + 2 * C) * 2
for k, v in t.items():
    print(f"{k}: {min(v):6.1f} us  {fl / min(v) * 1e-6:6.1f} TFLOP/s  {by / min(v) * 1e-3:6.1f} GB/s (algorithmic)")
