"""Row-tile conv kernel (csrc/conv1d_rt.hip) against the tile kernel on the frame-level 256-channel shapes of the training step
(interleaved rounds in one process)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptttspp_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
ops.CONV_RT_MIN_ROWS = 1  # (the sampler-size case is below the product's threshold)
SHAPES = [("frame prior 256->256 k17", 19, 1550, 256, 17, 1, None, False), ("pitch predictor 256->256 k5 relu", 19, 1550, 256, 5, 1, "relu", False),
          ("DiffNet dgrad 512->256 k3 d8 +res", 19, 1550, 512, 3, 8, None, True), ("DiffNet dgrad 512->256 k3 d1 +res", 19, 1550, 512, 3, 1, None, True),
          ("frame prior, sampler-size batch", 32, 540, 256, 17, 1, None, False)]


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3


torch.manual_seed(0)
for name, B, T, cin, ks, dil, act, use_res in SHAPES:
    x = torch.randn(B, T, cin, device=dev).bfloat16()
    w = torch.randn(256, cin, ks, device=dev) * (cin * ks) ** -0.5
    b = torch.zeros(256, device=dev)
    res = torch.randn(B, T, 256, device=dev).bfloat16() if use_res else None
    y = torch.empty(B, T, 256, device=dev, dtype=torch.bfloat16)
    wp, ws = ops.pack_conv_weight(w, torch.bfloat16), ops.pack_conv_weight(w, torch.bfloat16, 3)
    pad = dil * (ks - 1) // 2
    kw = dict(ks=ks, dil=dil, pad=pad, act=act, res=res, res_scale=0.7071 if use_res else 1.0, out=y)
    old = lambda: ops.conv1d(x, wp, b, 256, **kw)
    new = lambda: ops.conv1d(x, None, b, 256, wstream=ws, **kw)
    t = {"old": [], "new": []}
    for _ in range(5):
        t["old"].append(timeit(old))
        t["new"].append(timeit(new))
    fl = 2.0 * B * T * cin * 256 * ks
    to, tn = min(t["old"]), min(t["new"])
    print(f"{name:36s} B {B} T {T}: tile kernel {to:7.1f} us ({fl / to * 1e-6:6.1f} TF/s = {fl / to * 1e-6 / 2500:.3f})   row-tile {tn:7.1f} us "
          f"({fl / tn * 1e-6:6.1f} TF/s = {fl / tn * 1e-6 / 2500:.3f})   x{to / tn:.2f}", flush=True)
