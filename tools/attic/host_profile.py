"""cProfile of the training step's HOST side (the step is launch-bound): top functions by own time,
plus torch.profiler's list of aten ops by count."""
import cProfile, pstats, sys, io
import torch
sys.path.insert(0, "/root/repo")
import bench
from promptttspp_amd import config

dev = torch.device("cuda:0")
config.set_compute_dtype(torch.bfloat16)
model = bench.build_model(dev)
model.train()
batches = bench.make_batches(0, 1, 4, 30000, dev)
red, opt, sched = bench.train_setup(model, 1)
for b in batches[:3]:
    bench.train_step(model, b, red, opt, sched)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for b in batches[:3]:
    bench.train_step(model, b, red, opt, sched)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    bench.train_step(model, batches[3], red, opt, sched)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60)[:12000])
