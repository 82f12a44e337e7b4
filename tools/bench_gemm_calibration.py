"""What the SHAPES of the hot path allow: hipBLASLt (torch.matmul, bf16, f32 accumulation) on the exact implicit-GEMM shapes
of the frame-level and phone-level convolutions, next to the in-tree conv kernel on the same shape (im2col-free: the
library number is the pure GEMM with the taps already folded into K, i.e. an UPPER bound of what a library conv could do).
Diagnostic only -- the library is not in the product path.   python tools/bench_gemm_calibration.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptttspp_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
# name, M (rows), Cin, Cout, ks, dil, (B, T) of the conv launch with M = B * T
SHAPES = [("DiffNet dilated conv 256->512 k3", 30000, 256, 512, 3, 2, (52, 577)),
          ("DiffNet conv + proj as ONE K (4352)->256 dgrad-like", 30000, 4352, 256, 1, 1, (52, 577)),
          ("DiffNet output projection 256->512 k1", 30000, 256, 512, 1, 1, (52, 577)),
          ("DiffNet dgrad 512->256 k3", 30000, 512, 256, 3, 2, (52, 577)),
          ("frame prior 256->256 k17", 30000, 256, 256, 17, 1, (52, 577)),
          ("phone FFN 256->1024 k9", 2850, 256, 1024, 9, 1, (19, 150)),
          ("phone FFN 1024->256 k9", 2850, 1024, 256, 9, 1, (19, 150))]


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


def main():
    torch.manual_seed(0)
    print(f"{'shape':52s} {'M':>6s} {'N':>5s} {'K':>5s} | hipBLASLt us  TF/s  of 2.5PF | in-tree conv us  TF/s  of 2.5PF")
    out = {}
    for name, M, cin, cout, ks, dil, (B, T) in SHAPES:
        K = cin * ks
        a = torch.randn(M, K, device=dev).bfloat16()
        w = torch.randn(K, cout, device=dev).bfloat16()
        c = torch.empty(M, cout, device=dev, dtype=torch.bfloat16)
        t_lib = timeit(lambda: torch.matmul(a, w, out=c))
        fl = 2.0 * M * K * cout
        line = f"{name:52s} {M:6d} {cout:5d} {K:5d} | {t_lib:9.1f} {fl / t_lib * 1e-6:7.1f} {fl / t_lib * 1e-6 / 2500:7.3f}"
        if cin % 64 == 0 and cin <= 1024:
            x = torch.randn(B, T, cin, device=dev).bfloat16()
            wp = ops.pack_conv_weight(torch.randn(cout, cin, ks, device=dev) * 0.05, torch.bfloat16)
            y = torch.empty(B, T, cout, device=dev, dtype=torch.bfloat16)
            bias = torch.zeros(cout, device=dev)
            pad = dil * (ks - 1) // 2
            t_own = timeit(lambda: ops.conv1d(x, wp, bias, cout, ks=ks, dil=dil, pad=pad, out=y))
            fl2 = 2.0 * B * T * K * cout
            line += f" | {t_own:9.1f} {fl2 / t_own * 1e-6:7.1f} {fl2 / t_own * 1e-6 / 2500:7.3f}"
            out[name] = (round(fl / t_lib * 1e-6, 1), round(fl2 / t_own * 1e-6, 1))
        else:
            out[name] = (round(fl / t_lib * 1e-6, 1), None)
        print(line, flush=True)
    return out


if __name__ == "__main__":
    main()
