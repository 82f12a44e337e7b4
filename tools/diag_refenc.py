"""Run-to-run and path-to-path differences of the reference encoder forward (training mode, bf16)."""
import sys
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
from promptttspp_amd import config
from promptttspp_amd import functional as PF
from promptttspp_amd.modules.reference_encoder import ReferenceEncoder

dev = torch.device("cuda:0")
B, T = 19, 1500
torch.manual_seed(3)
ref = ReferenceEncoder().to(dev)
rng = np.random.default_rng(5)
mel = torch.from_numpy((1.5 * rng.standard_normal((B, 80, T))).astype(np.float32)).to(dev)
lens = torch.tensor([max(40, T - 37 * i) for i in range(B)], device=dev)
state = {k: v.clone() for k, v in ref.state_dict().items()}
runs = []
with config.use_dtype(torch.bfloat16):
    for drv in (True, True, False, False, True, False):
        ref.load_state_dict(state)
        ref.train()
        PF.STACK_DRIVERS = drv
        with torch.no_grad():
            y = ref(mel, lens)
        torch.cuda.synchronize()
        runs.append((drv, y.clone(), {k: v.clone() for k, v in ref.state_dict().items() if "running" in k}))
for i in range(len(runs)):
    for j in range(i + 1, len(runs)):
        a, b = runs[i], runs[j]
        dy = float((a[1] - b[1]).abs().max())
        ds = {k: float((a[2][k] - b[2][k]).abs().max() / (b[2][k].abs().max() + 1e-12)) for k in a[2]}
        first = next((k for k in ds if ds[k] > 0), None)
        print(f"run {i}({'drv' if a[0] else 'launch'}) vs {j}({'drv' if b[0] else 'launch'}): dy {dy:.3e}  first differing stat {first} "
              f"{ds.get(first, 0):.2e}  max stat diff {max(ds.values()):.2e}")
