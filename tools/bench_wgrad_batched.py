"""Per-layer weight gradients (ptpp_conv1d_wgrad: split-K partials + reduce) against the batched launch
(ptpp_conv1d_wgrad_batched: one owner block per dw tile) on the DiffNet stack's 20 layers at the bench batch shape."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from promptttspp_amd import ops

dev = torch.device("cuda:0")
B, T, C, L = 19, 1580, 256, 20
torch.manual_seed(0)
yin = [torch.randn(B, T, C, device=dev).bfloat16() for _ in range(L)]
dcond = torch.randn(B, T, L * 2 * C, device=dev).bfloat16()
do = [torch.randn(B, T, 2 * C, device=dev).bfloat16() for _ in range(L)]
dw3 = [torch.zeros(2 * C, C, 3, device=dev) for _ in range(L)]
dw1 = [torch.zeros(2 * C, C, 1, device=dev) for _ in range(L)]
db = [torch.zeros(2 * C, device=dev) for _ in range(L)]


def timed(f, n=5):
    f(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3


def per_layer3():
    for l in range(L):
        d = 2 ** (l % 4)
        ops.conv1d_wgrad(yin[l], dcond[:, :, l * 2 * C:(l + 1) * 2 * C], C, 2 * C, 3, d, d, dw_out=dw3[l], db_out=db[l])


def per_layer1():
    for l in range(L):
        ops.conv1d_wgrad(yin[l], do[l], C, 2 * C, 1, 1, 0, dw_out=dw1[l], db_out=db[l])


p3 = [(yin[l], dcond[:, :, l * 2 * C:(l + 1) * 2 * C], dw3[l], db[l], 2 ** (l % 4), 2 ** (l % 4)) for l in range(L)]
p1 = [(yin[l], do[l], dw1[l], db[l], 1, 0) for l in range(L)]
fl3 = 2.0 * B * T * C * 2 * C * 3 * L / 1e12
fl1 = fl3 / 3
for name, f, fl in (("k=3 per layer (20 launches + 20 reduces)", per_layer3, fl3), ("k=3 batched (1 launch)", lambda: ops.conv1d_wgrad_batched(p3, C, 2 * C, 3), fl3),
                    ("k=1 per layer", per_layer1, fl1), ("k=1 batched", lambda: ops.conv1d_wgrad_batched(p1, C, 2 * C, 1), fl1)):
    us = timed(f)
    print(f"{name:45s} {us:9.1f} us  {fl / us * 1e6:7.1f} TFLOP/s")
