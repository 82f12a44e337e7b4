"""Micro-benchmark of the normalisation kernels at the bench step's sizes (direct C-ABI calls)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptttspp_amd import nn_ops, ops  # noqa: E402


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    print("env", {k: v for k, v in os.environ.items() if k.startswith("PTPP_")})
    for rows, C in [(544000, 32), (136000, 32), (34000, 64), (8500, 64), (5000, 256), (27200, 256)]:
        x = torch.randn(rows, C, device=dev).to(dt)
        dy = torch.randn(rows, C, device=dev).to(dt)
        mean = torch.zeros(C, device=dev)
        rstd = torch.ones(C, device=dev)
        g = torch.ones(C, device=dev)
        b = torch.zeros(C, device=dev)
        y = torch.empty_like(x)
        sums = torch.empty(2 * C, device=dev)
        dx = torch.empty_like(x)
        L = nn_ops._lib.load()
        P = ops._ptr
        t1 = timeit(lambda: nn_ops.col_reduce(x, mean))
        t2 = timeit(lambda: L.ptpp_bn_act_fwd(P(x), P(mean), P(rstd), P(g), P(b), P(y), rows, C, 1, 1, ops._stream()))
        t3 = timeit(lambda: L.ptpp_bn_act_bwd(P(x), P(dy), P(mean), P(rstd), P(g), P(b), P(sums), P(dx), rows, C, 1, 1, 1,
                                              *ops.reduction_scratch(dev), ops._stream()))
        mb = rows * C * 2 / 1e6
        print(f"bn rows={rows} C={C} ({mb:.1f} MB/tensor): col_reduce {t1:.1f} us, fwd {t2:.1f} us, bwd(reduce+apply) {t3:.1f} us")
    for B, T, C in [(32, 850, 256), (32, 100, 256), (32, 850, 512)]:
        x = torch.randn(B, T, C, device=dev).to(dt)
        dy = torch.randn(B, T, C, device=dev).to(dt)
        g = torch.ones(C, device=dev)
        b = torch.zeros(C, device=dev)
        y, mean, rstd, xs = ops.layernorm_fwd(x, g, b, 1e-5, save_stats=True, save_sum=True)
        dg = torch.zeros(C, device=dev)
        db = torch.zeros(C, device=dev)
        t1 = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-5, save_stats=True, save_sum=True))
        t2 = timeit(lambda: ops.layernorm_bwd(dy, xs, g, mean, rstd, dgamma_out=dg, dbeta_out=db))
        print(f"ln B={B} T={T} C={C} ({B*T*C*2/1e6:.1f} MB/tensor): fwd {t1:.1f} us, bwd {t2:.1f} us")


if __name__ == "__main__":
    main()
