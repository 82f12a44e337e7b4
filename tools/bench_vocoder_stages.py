"""Where BigVGAN's time goes, stage by stage (BASELINE config 4: 64 x 1000 frames), and what the wide stages' two
launch kinds cost ALONE on one stream (the generator runs the three AMP blocks of a wide stage on three streams, so
rocprof's per-kernel durations there are inflated by the sharing)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.fill import fill_state_dict  # noqa: E402  (synthetic weights only)
from promptttspp_amd import ops  # noqa: E402
from promptttspp_amd.vocoders import BigVGAN  # noqa: E402

dev = torch.device("cuda:0")
B, T = int(os.environ.get("B", 64)), int(os.environ.get("T", 1000))
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[os.environ.get("DTYPE", "bf16")]
m = BigVGAN(80, 512, [6, 5, 4, 2], [12, 10, 8, 4], [3, 7, 11], [[1, 3, 5]] * 3)
fill_state_dict(m, seed=5, overrides={"weight_g": 0.4})
m = m.to(dev).eval().set_compute_dtype(dt)
m.fuse_wide_layers = os.environ.get('WIDE', '1') == '1'
m.wide_streams = os.environ.get('WIDE_STREAMS', '0') == '1'
x = torch.clamp(-5.5 + 2.1 * torch.randn(B, 80, T, device=dev), -11.5, 2.0)


def timed(f, n=3):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        r = f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r


with torch.no_grad():
    ms_all, _ = timed(lambda: m(x))
    print(f"whole generator: {ms_all:.2f} ms")
    pk = m._prepare()
    h = ops.bct_to_btc(x, m.compute_dtype)
    ms, h = timed(lambda: m._conv(h, pk["pre"]))
    print(f"conv_pre: {ms:.3f} ms")
    inv = 1.0 / m.num_kernels
    for s, (up, blocks) in enumerate(zip(pk["ups"], pk["mrfs"])):
        u = m.upsample_rates[s]
        Bh, Th, _ = h.shape
        ms, h = timed(lambda: m._conv(h, up).view(Bh, Th * u, up.cout // u))
        C = h.shape[-1]
        print(f"stage {s}: upsample x{u} -> C={C} T={h.shape[1]}: {ms:.3f} ms")
        hh = h.contiguous()
        ms, h = timed(lambda: m._mrf(hh, blocks, inv))
        fl = sum(2 * 2 * Bh * Th * u * C * C * l[1].ks for blk in blocks for l in blk) / 1e12
        print(f"stage {s}: MRF (9 AMP layers): {ms:.3f} ms   {fl / ms * 1e3:.0f} TFLOP/s")
    ms, _ = timed(lambda: m._post(h, pk))
    print(f"act_post + conv_post + tanh: {ms:.3f} ms")
