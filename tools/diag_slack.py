"""Which resource bounds the training step?  Inject 0.5 ms of (a) main-stream GPU spin, (b) host sleep into the forward
(before the decoder) or the backward (decoder's backward hook) and measure how much of it shows up in the step time."""
import sys, time
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
SPIN = int(0.5e-3 * 2.1e9)
mode = {"v": None}
dec = model.decoder
orig = dec.forward_cl

class Hook(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)
    @staticmethod
    def backward(ctx, g):
        if mode["v"] == "gpu_bwd":
            torch.cuda._sleep(SPIN)
        elif mode["v"] == "host_bwd":
            time.sleep(0.5e-3)
        return g

def patched(cond, mel, lengths):
    if mode["v"] == "gpu_fwd":
        torch.cuda._sleep(SPIN)
    elif mode["v"] == "host_fwd":
        time.sleep(0.5e-3)
    noise, pred = orig(cond, mel, lengths)
    return noise, Hook.apply(pred)

dec.forward_cl = patched

def run(n=24):
    for i in range(4):
        bench.train_step(model, batches[i % 8], red, opt, sched)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        bench.train_step(model, batches[i % 8], red, opt, sched)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e3 * th / n, 1e3 * (time.perf_counter() - t0) / n

# calibrate the spin
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record(); torch.cuda._sleep(SPIN); e1.record(); torch.cuda.synchronize()
print(f"spin kernel = {e0.elapsed_time(e1):.3f} ms")
for m in (None, "gpu_fwd", "gpu_bwd", "host_fwd", "host_bwd", None):
    mode["v"] = m
    h, w = run()
    print(f"{str(m):10s} host {h:6.2f} ms/step   wall {w:6.2f} ms/step")
