"""cProfile of the training step's host side, sorted by cumulative time (callers of the launch wrappers)."""
import cProfile, pstats, sys, io, time
import torch
sys.path.insert(0, "/root/repo")
import bench
from promptttspp_amd import config

dev = torch.device("cuda:0")
config.set_compute_dtype(torch.bfloat16)
model = bench.build_model(dev)
model.train()
batches = bench.make_batches(0, 1, 8, 30000, dev)
red, opt, sched = bench.train_setup(model, 1)
for b in batches[:4]:
    bench.train_step(model, b, red, opt, sched)
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in batches[4:8]:
    bench.train_step(model, b, red, opt, sched)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host {1e3 * (t1 - t0) / 4:.2f} ms/step, total {1e3 * (time.perf_counter() - t0) / 4:.2f} ms/step (no profiler)")
pr = cProfile.Profile()
pr.enable()
for b in batches[4:8]:
    bench.train_step(model, b, red, opt, sched)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(70)
print(s.getvalue()[:14000])
