"""Quick BigVGAN timing (BASELINE config 4 shape by default)."""
import argparse
import sys
import os
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.fill import fill_state_dict  # noqa: E402  (synthetic weights only)
from promptttspp_amd.vocoders import BigVGAN  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--frames", type=int, default=1000)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
dev = torch.device("cuda:0")
m = BigVGAN(80, 512, [6, 5, 4, 2], [12, 10, 8, 4], [3, 7, 11], [[1, 3, 5]] * 3)
fill_state_dict(m, seed=5, overrides={"weight_g": 0.4})
m = m.to(dev).eval().set_compute_dtype(torch.bfloat16 if a.dtype == "bf16" else torch.float32)
x = torch.clamp(-5.5 + 2.1 * torch.randn(a.batch, 80, a.frames, device=dev), -11.5, 2.0)
for _ in range(2):
    y = m(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    y = m(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
audio_s = a.batch * a.frames * 0.01
flop = a.batch * a.frames * 444.5e6
print(f"BigVGAN {a.dtype} B={a.batch} T={a.frames}: {dt*1e3:.2f} ms/iter  RTF={dt/audio_s:.3e}  "
      f"{flop/dt/1e12:.1f} TFLOP/s algorithmic  {a.batch*a.frames*789312*2/dt/1e12:.2f} TB/s algorithmic(bf16)")
