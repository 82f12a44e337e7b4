"""per-step host time / total time / allocator growth over distinct batches (is the step launch-bound,
GPU-bound, or stalled by allocator growth?)"""
import sys, time
import torch
sys.path.insert(0, "/root/repo")
import bench
from promptttspp_amd import config

dev = torch.device("cuda:0")
import os
if os.environ.get("PTPP_DP_FORCE_COLLECTIVES"):  # one rank over RCCL: the multi-rank machinery on a 1-GPU box
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29556")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
config.set_compute_dtype(torch.bfloat16)
model = bench.build_model(dev).train()
batches = bench.make_batches(0, 1, 26, 30000, dev)
red, opt, sched = bench.train_setup(model, 1)
seg = lambda: torch.cuda.memory_stats().get("segment.all.allocated", 0)
for i, b in enumerate(batches):
    torch.cuda.synchronize(); s0 = seg(); t0 = time.perf_counter()
    bench.train_step(model, b, red, opt, sched)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"step {i:2d} B={b[0].shape[0]:3d} Tp={b[0].shape[1]:4d} Tf={b[3].shape[2]:5d} host {1e3*(t1-t0):6.1f} ms total {1e3*(t2-t0):6.1f} ms "
          f"new segments {seg()-s0:3d} reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB", flush=True)
# same batch repeated, no per-step sync
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): bench.train_step(model, batches[5], red, opt, sched)
torch.cuda.synchronize(); print(f"same batch x10 async: {1e2*(time.perf_counter()-t0):.1f} ms/step")
torch.cuda.synchronize(); t0 = time.perf_counter()
for b in batches[6:16]: bench.train_step(model, b, red, opt, sched)
torch.cuda.synchronize(); print(f"10 distinct batches async (2nd visit): {1e2*(time.perf_counter()-t0):.1f} ms/step")
