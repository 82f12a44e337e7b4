"""Which call sites still pack weights per step (ops.pack_conv_weight) after the batched re-pack."""
import collections
import sys
import traceback

import torch

sys.path.insert(0, "/root/repo")
import bench  # noqa: E402
from promptttspp_amd import config, ops  # noqa: E402

dev = torch.device("cuda:0")
config.set_compute_dtype(torch.bfloat16)
model = bench.build_model(dev)
model.train()
batches = bench.make_batches(0, 1, 4, 30000, dev)
red, opt, sched = bench.train_setup(model, 1)
for b in batches[:3]:
    bench.train_step(model, b, red, opt, sched)
torch.cuda.synchronize()
seen = collections.Counter()
orig = ops.pack_conv_weight


def spy(w, dtype, mode=0):
    fr = [f for f in traceback.extract_stack()[:-1] if "/root/repo/promptttspp_amd" in f.filename]
    seen[" <- ".join(f"{f.filename.split('promptttspp_amd/')[1]}:{f.lineno}" for f in fr[-3:]), tuple(w.shape), mode] += 1
    return orig(w, dtype, mode)


ops.pack_conv_weight = spy
import promptttspp_amd.functional as PF  # noqa: E402

bench.train_step(model, batches[3], red, opt, sched)
torch.cuda.synchronize()
for k, v in seen.most_common(40):
    print(v, k)
