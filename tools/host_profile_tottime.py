import cProfile, pstats, sys, io, time, os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from promptttspp_amd import config
dev = torch.device("cuda:0")
config.set_compute_dtype(torch.bfloat16)
model = bench.build_model(dev); model.train()
batches = bench.make_batches(0, 1, 8, 30000, dev)
red, opt, sched = bench.train_setup(model, 1)
for b in batches[:4]:
    bench.train_step(model, b, red, opt, sched)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for b in batches[4:8]:
    bench.train_step(model, b, red, opt, sched)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
