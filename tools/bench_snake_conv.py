"""Micro-benchmark of ptpp_snake_conv1d_fwd on the wide BigVGAN stage shapes of BASELINE config 4 (64 x 10 s: C = 256 at
T = 6000, C = 128 at T = 30000), with phase ablation (PTPP_AMP_SKIP: 1 no Snake, 2 no conv)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptttspp_amd import ops  # noqa: E402
from promptttspp_amd.layers.activations import AntiAliasActivation  # noqa: E402

dev = torch.device("cuda:0")
taps = AntiAliasActivation(8).taps()
B = int(os.environ.get("B", 64))
for skip in [int(v) for v in os.environ.get("SKIPS", "0,1,2,3").split(",")]:
    os.environ["PTPP_AMP_SKIP"] = str(skip)
    for C, T in ((256, 6000), (128, 30000)):
        x = torch.randn(B, T, C, device=dev).bfloat16()
        res = torch.randn(B, T, C, device=dev).bfloat16()
        y = torch.empty_like(x)
        row = []
        for ks, d in ((3, 1), (7, 3), (11, 5), (11, 1)):
            wp = ops.pack_conv_weight(torch.randn(C, C, ks, device=dev) / (C * ks) ** 0.5, torch.bfloat16)
            ws = ops.amp_pack_wstream(wp, C, ks)
            b = 0.1 * torch.randn(C, device=dev)
            la = 0.3 * torch.randn(C, device=dev)
            run = lambda: ops.snake_conv1d(x, ws, b, la, taps, ks, d, res=res, out=y)  # noqa: E731
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            fl = 2 * B * T * C * C * ks / 1e12
            row.append(f"k={ks} d={d}: {ms:6.3f} ms {fl / ms * 1e3:5.0f} TF/s")
        print(f"skip {skip} C={C}: " + "  ".join(row))
