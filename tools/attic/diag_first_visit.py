"""cProfile of training steps on batch shapes seen for the FIRST time (where does the extra host time go?)"""
import cProfile, pstats, io, sys
import torch
sys.path.insert(0, "/root/repo")
import bench
from promptttspp_amd import config

dev = torch.device("cuda:0")
config.set_compute_dtype(torch.bfloat16)
model = bench.build_model(dev).train()
batches = bench.make_batches(0, 1, 14, 30000, dev)
red, opt, sched = bench.train_setup(model, 1)
for b in batches[:6]:
    bench.train_step(model, b, red, opt, sched)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for b in batches[6:14]:
    bench.train_step(model, b, red, opt, sched)
    torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25); print(s.getvalue()[:6000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(40); print(s.getvalue()[:9000])
