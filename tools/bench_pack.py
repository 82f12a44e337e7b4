"""The per-step re-pack of every cached weight operand (ptpp_pack_conv_weights_batched, functional.repack_all): what the
launch table holds (elements by pack mode and tap count) and how long the launch takes on the bench model."""
import collections
import sys

import torch

sys.path.insert(0, "/root/repo")
import bench  # noqa: E402
from promptttspp_amd import config, ops  # noqa: E402
import promptttspp_amd.functional as PF  # noqa: E402

dev = torch.device("cuda:0")
config.set_compute_dtype(torch.bfloat16)
model = bench.build_model(dev)
model.train()
batches = bench.make_batches(0, 1, 4, 30000, dev)
red, opt, sched = bench.train_setup(model, 1)
for b in batches[:3]:
    bench.train_step(model, b, red, opt, sched)
torch.cuda.synchronize()
rp = PF._repack
tab = rp["table"].cpu().numpy()
hist = collections.Counter()
for src, dst, cout, cin, ks, mode, dcode, innerp, off, blk in tab:
    hist[(int(mode), int(ks))] += int(cout) * int(cin) * int(ks)
tot = sum(hist.values())
print(f"{len(tab)} table rows, {rp['blocks']} blocks, {tot / 1e6:.1f} M packed elements "
      f"({sum(p.numel() for p in model.parameters()) / 1e6:.1f} M parameters)")
for (mode, ks), v in sorted(hist.items(), key=lambda kv: -kv[1]):
    print(f"  mode {mode} ks {ks:2d}: {v / 1e6:7.2f} M elements ({100 * v / tot:4.1f} %)")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(3):
    ops.pack_conv_weights_batched(rp["table"], rp["n"], rp["map"], rp["blocks"])
torch.cuda.synchronize()
N = 20
ev[0].record()
for _ in range(N):
    ops.pack_conv_weights_batched(rp["table"], rp["n"], rp["map"], rp["blocks"])
ev[1].record()
torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) * 1e3 / N
print(f"re-pack launch: {us:.1f} us  ({(tot * 4 + tot * 2) / us / 1e6:.2f} TB/s of f32 reads + bf16 writes)")
