"""Who waits for whom at the stream joins of a training step: for every join, the time between the waiting stream reaching it
and the branch stream finishing its work (positive = the waiting stream idled that long).  Forward joins are probed inside the
model (models/.../model.py JOIN_PROBE); the end-of-backward join (functional.sync_wgrad_stream: weight-gradient side stream and
the branch streams' backward) is probed here."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from promptttspp_amd import config  # noqa: E402
from promptttspp_amd import functional as PF  # noqa: E402
from promptttspp_amd.models.prompttts_mdn_v2_final import model as M  # noqa: E402

dev = torch.device("cuda:0")
config.set_compute_dtype(torch.bfloat16)
model = bench.build_model(dev).train()
batches = bench.make_batches(0, 1, 12, 30000, dev)
red, opt, sched = bench.train_setup(model, 1)
for b in batches[:6]:
    bench.train_step(model, b, red, opt, sched)
torch.cuda.synchronize()

orig_sync = PF.sync_wgrad_stream
probe = []


def sync_probe():
    cur = torch.cuda.current_stream()
    streams = [("weight-gradient side stream", PF._direct["side"])] + [(f"branch stream {i} (backward)", s) for i, s in enumerate(PF._grad_streams)]
    for name, s in streams:
        if s is None or s == cur:
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        e1.record(s)
        probe.append((name + " -> end of backward", e0, e1))
    orig_sync()


PF.sync_wgrad_stream = sync_probe
M.JOIN_PROBE = probe
acc = {}
starts = []
for b in batches[6:]:
    probe.clear()
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    bench.train_step(model, b, red, opt, sched)
    s1.record()
    torch.cuda.synchronize()
    acc.setdefault("step (events on the main stream)", []).append(s0.elapsed_time(s1))
    for name, e0, e1 in probe:
        acc.setdefault(name, []).append(e0.elapsed_time(e1))
        acc.setdefault(name + " [main reached the join at, ms from step start]", []).append(s0.elapsed_time(e0))
for k, v in acc.items():
    v = sorted(v)
    print(f"{k:95s} median {v[len(v) // 2]:7.2f} ms   (min {v[0]:6.2f}, max {v[-1]:6.2f})")
