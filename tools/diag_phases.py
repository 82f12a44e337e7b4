"""GPU-side phases of the training step on the MAIN stream, from events recorded without any host sync inside the run
(the host stays ahead as in bench.py): forward | backward | join of the gradient streams (red.finish) | optimizer."""
import sys, time
import torch
sys.path.insert(0, "/root/repo")
import bench
from promptttspp_amd import config, ops

dev = torch.device("cuda:0")
config.set_compute_dtype(torch.bfloat16)
model = bench.build_model(dev)
model.train()
batches = bench.make_batches(0, 1, 8, 30000, dev)
red, opt, sched = bench.train_setup(model, 1)
N = 16
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(N)]
host = []
for i in range(4):
    bench.train_step(model, batches[i % 8], red, opt, sched)
torch.cuda.synchronize()
t00 = time.perf_counter()
for i in range(N):
    b = batches[i % 8]
    h0 = time.perf_counter()
    with ops.pinned_stream():
        ev[i][0].record()
        red.zero_grad()
        out = model(b)
        ev[i][1].record()
        h1 = time.perf_counter()
        with torch.autograd.set_multithreading_enabled(False):
            out["loss"].backward()
        ev[i][2].record()
        h2 = time.perf_counter()
        red.finish()
        ev[i][3].record()
        opt.step()
        ev[i][4].record()
    sched.step()
    host.append((h1 - h0, h2 - h1, time.perf_counter() - h2))
t_host = time.perf_counter() - t00
torch.cuda.synchronize()
t_all = time.perf_counter() - t00
print(f"host {1e3 * t_host / N:.2f} ms/step, wall {1e3 * t_all / N:.2f} ms/step")
print("step   fwd    bwd   join   opt  | to next start | host fwd bwd rest")
for i in range(4, N):
    d = [ev[i][k].elapsed_time(ev[i][k + 1]) for k in range(4)]
    nxt = ev[i][4].elapsed_time(ev[i + 1][0]) if i + 1 < N else float("nan")
    print(f"{i:3d}  {d[0]:6.2f} {d[1]:6.2f} {d[2]:6.2f} {d[3]:6.2f} | {nxt:6.2f} | {1e3*host[i][0]:5.2f} {1e3*host[i][1]:5.2f} {1e3*host[i][2]:5.2f}")
