"""Micro-benchmark of the frame-level convolutions of the training step (and the wide vocoder stages) through the
C ABI: forward launches of the 128-row tile configuration.  PTPP_CONV_TILE selects experimental wave-tile shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptttspp_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
print("PTPP_CONV_TILE", os.environ.get("PTPP_CONV_TILE"), "PTPP_CONV_GLDS", os.environ.get("PTPP_CONV_GLDS"))
dump = os.environ.get("PTPP_CONV_DUMP")  # directory: save outputs (first run) / compare bit for bit (later runs)
shapes = [("DiffNet dilated 256->512 k3", 52, 576, 256, 512, 3, 2), ("DiffNet 1x1 256->512", 52, 576, 256, 512, 1, 1),
          ("frame prior 256->256 k17", 52, 576, 256, 256, 17, 1), ("pitch pred 256->256 k5", 52, 576, 256, 256, 5, 1),
          ("dgrad 512->256 k3", 52, 576, 512, 256, 3, 2), ("BigVGAN C=128 k7 d3", 64, 30000, 128, 128, 7, 3),
          ("BigVGAN C=256 k11 d5", 64, 6000, 256, 256, 11, 5), ("BigVGAN C=256 k3", 64, 6000, 256, 256, 3, 1),
          ("DiffNet cond_all 256->10240", 19, 1100, 256, 10240, 1, 1),
          ("phone FFN 1024->256 k9", 19, 150, 1024, 256, 9, 1), ("phone FFN 256->1024 k9", 19, 150, 256, 1024, 9, 1),
          ("phone linear 256->768", 19, 150, 256, 768, 1, 1), ("phone linear 256->256", 19, 150, 256, 256, 1, 1)]
if os.environ.get("PTPP_BENCH_ONLY"):
    shapes = [s_ for s_ in shapes if os.environ["PTPP_BENCH_ONLY"] in s_[0]]
tot = 0.0
torch.manual_seed(0)
for name, B, T, cin, cout, ks, dil in shapes:
    x = torch.randn(B, T, cin, device=dev).bfloat16()
    res = torch.randn(B, T, cout, device=dev).bfloat16()
    wp = ops.pack_conv_weight(torch.randn(cout, cin, ks, device=dev) * 0.05, torch.bfloat16)
    b = torch.zeros(cout, device=dev)
    y = torch.empty(B, T, cout, device=dev, dtype=torch.bfloat16)
    pad = dil * (ks - 1) // 2
    f = (lambda: ops.conv1d(x, wp, b, cout, ks=ks, dil=dil, pad=pad, out=y)) if cout > 4096 else \
        (lambda: ops.conv1d(x, wp, b, cout, ks=ks, dil=dil, pad=pad, res=res, out=y))  # noqa: E731
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    if dump:
        path = os.path.join(dump, name.replace(" ", "_").replace(">", "") + ".pt")
        if os.path.exists(path):
            ref = torch.load(path).to(dev)
            print("   bit-identical to the saved output:", bool(torch.equal(ref, y)), " max |diff|", float((ref.float() - y.float()).abs().max()))
        else:
            os.makedirs(dump, exist_ok=True)
            torch.save(y.cpu(), path)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        f()
    e.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(e) / 10 * 1e3
    tot += us
    tf = 2.0 * B * T * cin * cout * ks / 1e12
    gb = (x.numel() + 2 * y.numel()) * 2 / 1e9
    print(f"{name:28s} B={B} T={T}: {us:8.1f} us  {tf / us * 1e6:6.1f} TFLOP/s = {tf / us * 1e6 / 2500:5.3f} of 2.5 PF   {gb / us * 1e3:5.2f} TB/s")
print(f"sum {tot:.1f} us")
