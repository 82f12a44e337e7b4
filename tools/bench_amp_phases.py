"""Phase ablation of the fused AMP-layer kernel (PTPP_AMP_SKIP bits: 1 snake 1, 2 conv 1, 4 snake 2, 8 conv 2 K loop) on the
BASELINE config-4 shapes; PTPP_AMP_VARIANT picks the tile height."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptttspp_amd import ops  # noqa: E402
from promptttspp_amd.layers.activations import AntiAliasActivation  # noqa: E402

dev = torch.device("cuda:0")
taps = AntiAliasActivation(8).taps()
B = int(os.environ.get("B", 64))
cases = [(32, 240000, 3, 1), (32, 240000, 11, 3), (64, 120000, 3, 1), (64, 120000, 7, 3), (64, 120000, 11, 3)]
if os.environ.get('CASES'):
    cases = cases[-1:]
for skip in [int(v) for v in os.environ.get('SKIPS', '0,5,10,15,1,2').split(',')]:
    os.environ["PTPP_AMP_SKIP"] = str(skip)
    row = []
    for C, T, ks, d in cases:
        x = torch.randn(B, T, C, device=dev).bfloat16()
        y = torch.empty_like(x)
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
        row.append(e0.elapsed_time(e1) / 5)
        del x, y
    print(f"skip {skip:2d}: " + "  ".join(f"C={c} k={k} d={d}: {ms:6.3f}" for (c, _, k, d), ms in zip(cases, row)))
