"""Which cached packed operands does a training step actually read?  Every cached operand is re-packed after every optimiser step
(functional.repack_all); an entry that no launch asks for any more is pure re-pack traffic (round 6 found the Conformer's
feed-forward weights packed in four forms, two of them unread: profiles/r06_experiments.md section 6)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from promptttspp_amd import config  # noqa: E402
from promptttspp_amd import functional as PF  # noqa: E402

dev = torch.device("cuda:0")
config.set_compute_dtype(torch.bfloat16)
model = bench.build_model(dev)
model.train()
batches = bench.make_batches(0, 1, 6, 30000, dev)
red, opt, sched = bench.train_setup(model, 1)
for b in batches[:3]:
    bench.train_step(model, b, red, opt, sched)
torch.cuda.synchronize()
seen = collections.Counter()
orig = PF._packed_entry


def spy(ws, dtype, mode):
    ent = orig(ws, dtype, mode)
    seen[id(ent)] += 1
    return ent


PF._packed_entry = spy
for b in batches[3:5]:
    bench.train_step(model, b, red, opt, sched)
torch.cuda.synchronize()
PF._packed_entry = orig
names = {id(p): n for n, p in model.named_parameters()}
tot = unread = 0
rows = []
for key, ent in PF._pack_cache.items():
    ws = ent.srcs()
    if ws is None:
        continue
    n = ent.wp.numel()
    tot += n
    if not seen[id(ent)]:
        unread += n
        rows.append((n, ent.mode, [names.get(id(w), "?") for w in ws][:2]))
print(f"{len(PF._pack_cache)} cached operands, {tot / 1e6:.1f} M packed elements; never requested in two steps: {unread / 1e6:.1f} M")
for n, mode, nm in sorted(rows, reverse=True)[:30]:
    print(f"  {n / 1e6:8.2f} M  mode {mode}  {nm}")
by_mode = collections.Counter()
for ent in PF._pack_cache.values():
    by_mode[(ent.mode, bool(ent.late))] += ent.wp.numel()
print("by (mode, late):", {k: round(v / 1e6, 1) for k, v in sorted(by_mode.items())})
