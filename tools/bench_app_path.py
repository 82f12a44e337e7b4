"""BASELINE config 5: prompt -> waveform for 32 prompts (infer_batch with the tamed synthetic duration head of
the test fixtures, 100-step sampler, low-pass F0, F0-aware BigVGAN), bf16 decoder / vocoder."""
import sys, time
import torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_hip_acoustic as T
from oracle.fill import fill_state_dict
from promptttspp_amd import config
from promptttspp_amd.hydra_lite import compose, instantiate
from promptttspp.utils.model import lowpass_filter

dev = torch.device("cuda:0")
model, g = T._model(dev)          # synthetic weights with a tamed duration head
model.eval()
conf = "/root/repo/egs/proposed/bin/conf"
voc = instantiate(compose(conf, "demo", []).vocoder)
fill_state_dict(voc, seed=5, overrides={"weight_g": 0.4})
voc = voc.to(dev).eval()
config.set_compute_dtype(torch.bfloat16)
voc.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
B = 32
Tp = torch.randint(40, 121, (B,))
ph = torch.zeros(B, int(Tp.max()), dtype=torch.long)
for b in range(B):
    ph[b, : Tp[b]] = torch.randint(3, 90, (int(Tp[b]),))
ph, pl = ph.to(dev), Tp.to(dev)
ids = torch.randint(1000, 30000, (B, 24), device=dev)
prm = (ids, torch.ones_like(ids))

import os
parts = [0.0, 0.0, 0.0]


def run():
    with torch.no_grad():
        torch.cuda.synchronize(); a = time.perf_counter()
        mel, cf0, vuv, flen = model.infer_batch(ph, pl, style_prompt=prm, use_max=True, noise_scale=0.5, return_f0=True)
        torch.cuda.synchronize(); b = time.perf_counter()
        f0 = lowpass_filter(cf0, 100, cutoff=20).exp()
        f0[vuv < 0.5] = 0
        torch.cuda.synchronize(); c = time.perf_counter()
        wav = voc(mel, f0)
        torch.cuda.synchronize(); d = time.perf_counter()
        parts[0] += b - a; parts[1] += c - b; parts[2] += d - c
    return mel, flen, wav

voc.fuse_wide_layers = os.environ.get("WIDE", "1") == "1"
for _ in range(2):
    mel, flen, wav = run()
parts[:] = [0.0, 0.0, 0.0]
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 3
for _ in range(n):
    mel, flen, wav = run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
fr = int(flen.sum())
print("infer_batch %.1f ms, low-pass + gating %.1f ms, vocoder %.1f ms" % tuple(1e3 * v / n for v in parts))
print(f"app path, 32 prompts, Tp~U(40,120): padded mel {tuple(mel.shape)}, {fr} valid frames = {fr*0.01:.1f} s audio: "
      f"{1e3*dt:.1f} ms per batch, RTF {dt/(fr*0.01):.5f}, finite={bool(torch.isfinite(wav).all())}")
