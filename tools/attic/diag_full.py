import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
warnings.simplefilter("ignore")
import numpy as np, torch
from conftest import load_golden, rel_err, key_shapes
import test_hip_acoustic as T
from test_oracle_golden_am import synth_sd, TAME, TAME_OFF
from oracle import ref_torch as R
from promptttspp_amd import config
dev = torch.device("cuda:0")
config.set_compute_dtype(torch.float32)
m, g = T._model(dev)
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
    for a in ("dropout_rate", "positional_dropout_rate", "p_dropout", "p"):
        if isinstance(getattr(mod, a, None), float): setattr(mod, a, 0.0)
m.train()
m.decoder.injected = {"t": g["t"], "noise": g["noise"]}
out = m(T._batch(g, dev))
which = sys.argv[1] if len(sys.argv) > 1 else "loss"
out[which].backward()
sd = synth_sd(key_shapes(g["keys"]), 100, TAME, TAME_OFF)
names = [n for n, p in m.named_parameters() if p.requires_grad and not n.startswith("prompt_encoder.bert")]
for n in names: sd[n] = sd[n].clone().requires_grad_()
batch = (g["phon"], g["dur"], g["plen"], g["mel"], g["cf0"], g["vuv"], g["flen"], g["ids"], g["am"])
lo = R.model_forward(sd, batch, g["t"], g["noise"], train_bn=True)[which]
gr = torch.autograd.grad(lo, [sd[n] for n in names], allow_unused=True)
P = dict(m.named_parameters())
rows = []
for n, go in zip(names, gr):
    gp = P[n].grad
    if go is None and (gp is None or float(gp.abs().max()) == 0.0):
        continue
    if go is None or gp is None:
        rows.append((float("nan"), n, go is None, gp is None)); continue
    rows.append((rel_err(gp.cpu(), go), n, float(go.abs().max())))
bad = [r for r in rows if not (r[0] < 1e-4)]
for r in bad[:6] + bad[-6:]: print(r)
print("n params", len(rows), "bad", sum(1 for r in rows if not (r[0] < 1e-4)))
