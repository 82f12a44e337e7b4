"""Host time of the training step with the autograd engine's worker thread on (default) and off
(torch.autograd.set_multithreading_enabled(False): backward runs on the calling thread -- the
custom Functions' Python backward then needs no GIL hand-off per node)."""
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
import bench  # noqa: E402
from promptttspp_amd import config  # noqa: E402

dev = torch.device("cuda:0")
config.set_compute_dtype(torch.bfloat16)
model = bench.build_model(dev).train()
batches = bench.make_batches(0, 1, 12, 30000, dev)
red, opt, sched = bench.train_setup(model, 1)
for b in batches[:4]:
    bench.train_step(model, b, red, opt, sched)
torch.cuda.synchronize()


def run(tag):
    hs, ts = [], []
    for b in batches:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.train_step(model, b, red, opt, sched)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        hs.append(t1 - t0)
        ts.append(t2 - t0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches:
        bench.train_step(model, b, red, opt, sched)
    torch.cuda.synchronize()
    print(f"{tag}: host {1e3 * sum(hs) / len(hs):.1f} ms, step (synced) {1e3 * sum(ts) / len(ts):.1f} ms, "
          f"async {1e3 * (time.perf_counter() - t0) / len(batches):.1f} ms/step")


for rep in range(2):
    run("engine thread ")
    with torch.autograd.set_multithreading_enabled(False):
        run("calling thread")
