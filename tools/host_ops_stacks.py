"""Which source lines of this repository issue the small aten kernels of one training step:
torch.profiler with stacks, aggregated per (aten op, innermost repo frame)."""
import collections
import sys

import torch

sys.path.insert(0, "/root/repo")
import bench  # noqa: E402
from promptttspp_amd import config  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

dev = torch.device("cuda:0")
config.set_compute_dtype(torch.bfloat16)
model = bench.build_model(dev)
model.train()
batches = bench.make_batches(0, 1, 4, 30000, dev)
red, opt, sched = bench.train_setup(model, 1)
for b in batches[:3]:
    bench.train_step(model, b, red, opt, sched)
torch.cuda.synchronize()
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

by_line = collections.Counter()
by_op = collections.Counter()
SKIP = ("view", "reshape", "slice", "select", "transpose", "permute", "expand", "unsqueeze", "squeeze", "as_strided",
        "detach", "alias", "t.default", "empty", "_unsafe_view", "size", "stride", "is_", "record_stream", "unbind",
        "split", "chunk", "_local_scalar", "lift_fresh", "narrow", "unfold", "numel")


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in SKIP):
            f = sys._getframe(0).f_back
            frame = "native backward node / engine"
            while f is not None:
                fn = f.f_code.co_filename
                if fn.startswith("/root/repo/") and "/tools/" not in fn and not fn.endswith("bench.py"):
                    frame = f"{fn[len('/root/repo/'):]}:{f.f_lineno} ({f.f_code.co_name})"
                    break
                f = f.f_back
            by_line[frame, name] += 1
            by_op[name] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    bench.train_step(model, batches[3], red, opt, sched)
    torch.cuda.synchronize()
print("aten ops dispatched in one step (views excluded):", sum(by_op.values()))
for k, v in by_op.most_common(40):
    print(f"  {v:5d}  {k}")
print("by innermost repository frame:")
for (fr, op), v in by_line.most_common(110):
    print(f"  {v:5d}  {op:38s} {fr}")
