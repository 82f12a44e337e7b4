"""diagnostic: per-parameter difference between direct and autograd gradient accumulation"""
import sys
import torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_hip_acoustic as T
from promptttspp_amd import config, functional as PF
from promptttspp_amd.parallel import FlatGradReducer

dev = torch.device("cuda:0")
config.set_compute_dtype(torch.float32)
m, g = T._model(dev)
m.train()
m.decoder.injected = {"t": g["t"], "noise": g["noise"]}
names = [n for n, p in m.named_parameters() if p.requires_grad]
params = [p for p in m.parameters() if p.requires_grad]
def run():
    m.decoder.injected = {"t": g["t"], "noise": g["noise"]}
    PF.manual_seed(7); torch.manual_seed(3)
    out = m(T._batch(g, dev)); out["loss"].backward(); return float(out["loss"])
PF.enable_direct_grads(False)
l0 = run(); ref = [p.grad.clone() if p.grad is not None else None for p in params]
for p in params: p.grad = None
l1 = run(); ref2 = [p.grad.clone() if p.grad is not None else None for p in params]
for p in params: p.grad = None
red = FlatGradReducer(params); red.zero_grad()
l2 = run()
print("losses", l0, l1, l2)
bad = 0
for n, p, r, r2 in zip(names, params, ref, ref2):
    if r is None: continue
    s = float(r.abs().max()) + 1e-12
    e = float((p.grad - r).abs().max()) / s
    e2 = float((r2 - r).abs().max()) / s
    if e > 2e-5 or e2 > 2e-5:
        bad += 1
        if bad < 60: print(f"{n:70s} direct-vs-auto {e:.2e}  auto-vs-auto {e2:.2e}")
print("bad", bad, "of", len(params))
