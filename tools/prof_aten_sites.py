"""Which Python lines of the package issue the torch-native (aten) ops of a training step: a TorchDispatchMode that records,
for every non-view aten op on device tensors, the innermost promptttspp_amd / bench.py frame (native autograd nodes of the
backward have none: they are listed under their op name)."""
import collections
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, "/root/repo")
import bench  # noqa: E402
from promptttspp_amd import config  # noqa: E402

dev = torch.device("cuda:0")
config.set_compute_dtype(torch.bfloat16)
model = bench.build_model(dev)
model.train()
batches = bench.make_batches(0, 1, 6, 30000, dev)
red, opt, sched = bench.train_setup(model, 1)
for b in batches[:4]:
    bench.train_step(model, b, red, opt, sched)
torch.cuda.synchronize()
VIEWS = ("view", "reshape", "transpose", "permute", "slice", "select", "expand", "unsqueeze", "squeeze", "as_strided", "detach", "alias",
         "t.default", "split", "unbind", "_unsafe_view", "empty", "narrow", "chunk", "size", "stride", "is_", "sym_", "lift", "unfold",
         "_local_scalar", "set_", "record_stream", "new_empty", "empty_like", "empty_strided")
sites = collections.defaultdict(collections.Counter)
bwd_seq = []


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEWS):
            flat = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
            if any(t.is_cuda for t in flat):
                fr = [f for f in traceback.extract_stack()[:-1] if "/root/repo/promptttspp_amd" in f.filename or f.filename.endswith("bench.py")]
                site = f"{fr[-1].filename.split('/root/repo/')[-1]}:{fr[-1].lineno} {fr[-1].name}" if fr else "(no package frame)"
                sites[site][name.replace("aten.", "")] += 1
                if not fr or fr[-1].name == "train_step":  # backward of torch-native nodes: the sequence, with shapes
                    bwd_seq.append((name.replace("aten.", ""), [tuple(t.shape) for t in flat][:3]))
        return func(*args, **(kwargs or {}))


with Spy():
    bench.train_step(model, batches[4], red, opt, sched)
torch.cuda.synchronize()
tot = sum(sum(c.values()) for c in sites.values())
print(f"{tot} non-view aten ops on device tensors in the step")
for s, c in sorted(sites.items(), key=lambda kv: -sum(kv[1].values()))[:60]:
    print(f"{sum(c.values()):4d}  {s[:95]:95s} {dict(c.most_common(4))}")

print("-- the torch-native launches of the backward, in issue order (op, shapes of the first tensor arguments):")
for i, (n, sh) in enumerate(bwd_seq):
    print(f"  {i:3d} {n:28s} {sh}")
