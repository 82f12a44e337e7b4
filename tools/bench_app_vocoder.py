"""F0-aware BigVGAN at the config-5 app-path shape (32 prompts, ~590 frames) -- old against new kernels."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.fill import fill_state_dict  # noqa: E402
from promptttspp_amd.vocoders import BigVGAN, F0AwareBigVGAN  # noqa: E402

dev = torch.device("cuda:0")
kw = dict(in_channel=80, upsample_initial_channel=512, upsample_rates=[6, 5, 4, 2], upsample_kernel_sizes=[12, 10, 8, 4],
          resblock_kernel_sizes=[3, 7, 11], resblock_dilations=[[1, 3, 5]] * 3)
for name, m in (("F0AwareBigVGAN", F0AwareBigVGAN(sampling_rate=24000, harmonic_num=8, **kw)), ("BigVGAN", BigVGAN(**kw))):
    fill_state_dict(m, seed=5, overrides={"weight_g": 0.4})
    m = m.to(dev).eval().set_compute_dtype(torch.bfloat16)
    for B, T in ((32, 590), (1, 590), (8, 300)):
        x = torch.clamp(-5.5 + 2.1 * torch.randn(B, 80, T, device=dev), -11.5, 2.0)
        f0 = 120 + 50 * torch.rand(B, 1, T, device=dev)
        args = (x, f0) if name.startswith("F0") else (x,)
        for wide, amp in ((True, True), (False, True), (False, False)):
            m.fuse_wide_layers, m.fuse_amp_layers = wide, amp
            for _ in range(2):
                m(*args)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                m(*args)
            torch.cuda.synchronize()
            print(f"{name} B={B} T={T} wide={wide} amp={amp}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
