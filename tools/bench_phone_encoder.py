"""The phone-level Conformer encoder of the bench model alone: forward + backward on one bench batch's phone tensor
(kernel table under rocprofv3: tools/prof_summary.py)."""
import sys

import torch

sys.path.insert(0, "/root/repo")
import bench  # noqa: E402
from promptttspp_amd import config  # noqa: E402

dev = torch.device("cuda:0")
config.set_compute_dtype(torch.bfloat16)
model = bench.build_model(dev)
model.train()
batch = bench.make_batches(0, 1, 1, 30000, dev)[0]
red, opt, sched = bench.train_setup(model, 1)
for _ in range(2):
    bench.train_step(model, batch, red, opt, sched)
enc = model.encoder
cap = {}
orig = enc.forward_cl


def spy(x, *a):
    cap["args"] = (x.detach(),) + a
    return orig(x, *a)


enc.forward_cl = spy
bench.train_step(model, batch, red, opt, sched)
enc.forward_cl = orig
x0, *rest = cap["args"]
print("encoder input", tuple(x0.shape), x0.dtype, "lengths", rest[0].tolist() if torch.is_tensor(rest[0]) else rest[0])
g = torch.randn_like(orig(x0, *rest))
N = 20
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for it in range(N + 3):
    x = x0.clone().requires_grad_(True)
    ev[0].record()
    y = orig(x, *rest)
    ev[1].record()
    y.backward(g)
    from promptttspp_amd import functional as PF
    PF.sync_wgrad_stream()
    ev[2].record()
    torch.cuda.synchronize()
    if it >= 3:
        tf += ev[0].elapsed_time(ev[1])
        tb += ev[1].elapsed_time(ev[2])
print(f"forward {tf / N:.3f} ms, backward (incl. weight gradients joined) {tb / N:.3f} ms")
