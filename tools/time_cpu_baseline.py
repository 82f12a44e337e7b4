import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter("ignore")
import torch
import bench
print("cpu_count", os.cpu_count(), flush=True)
dev = torch.device("cuda:0")
from promptttspp_amd import config
config.set_compute_dtype(torch.bfloat16)
model = bench.build_model(dev).train()
batches = bench.make_batches(0, 1, 2, 30000, dev)
for nthr in (16, 32, 64):
    os.environ["PTPP_CPU_THREADS"] = str(nthr)
    t0 = time.time()
    r = bench.cpu_baseline(model, batches[1])
    print(nthr, "threads:", r, f"total {time.time() - t0:.1f}s", flush=True)
