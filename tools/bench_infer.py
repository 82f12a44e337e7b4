"""100-step diffusion sampler latency (the decoder half of infer_batch) with and without the HIP-graph
replay, on synthetic conditioning of explicit size (a random-init duration head predicts unbounded lengths)."""
import sys, time
import torch
sys.path.insert(0, "/root/repo")
import bench
from promptttspp_amd import config

dev = torch.device("cuda:0")
config.set_compute_dtype(torch.bfloat16)
model = bench.build_model(dev).eval()
dec = model.decoder
for B, Tf in ((1, 500), (8, 800), (32, 1000)):
    cond = torch.randn(B, Tf, 256, device=dev).bfloat16()
    for ug in (False, True):
        with torch.no_grad():
            for _ in range(2):
                mel = dec.inference_cl(cond, use_graph=ug)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 3
            for _ in range(n):
                mel = dec.inference_cl(cond, use_graph=ug)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        frames = B * Tf
        print(f"B={B:3d} Tf={Tf:5d} graph={ug}: {1e3*dt:8.1f} ms per batch ({1e3*dt/dec.K_step:6.2f} ms/step), "
              f"{frames/dt:10.0f} frames/s, RTF(sampler) {dt/(frames*0.01):.5f}", flush=True)
