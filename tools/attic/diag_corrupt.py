import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
warnings.simplefilter("ignore")
import torch
import test_hip_acoustic as T
from promptttspp_amd import config, ops
from promptttspp_amd.modules import variance_adaptor as VA
dev = torch.device("cuda:0")
config.set_compute_dtype(torch.float32)
m, g = T._model(dev)
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
    for a in ("dropout_rate", "positional_dropout_rate", "p_dropout", "p"):
        if isinstance(getattr(mod, a, None), float): setattr(mod, a, 0.0)
m.train()
rec = []
orig_conv, orig_ln = ops.conv1d, ops.layernorm_fwd
def conv(x, *a, **k):
    y = orig_conv(x, *a, **k)
    rec.append(("conv", len(rec), x, x.clone(), y, y.clone()))
    return y
def ln(x, *a, **k):
    out = orig_ln(x, *a, **k)
    rec.append(("ln", len(rec), x, x.clone(), out[0], out[0].clone()))
    for nm, t in zip(("mean", "rstd", "xsum"), out[1:]):
        if t is not None:
            rec.append(("ln_" + nm, len(rec), t, t.clone(), t, t.clone()))
    return out
ops.conv1d, ops.layernorm_fwd = conv, ln
m.decoder.injected = {"t": g["t"], "noise": g["noise"]}
out = m(T._batch(g, dev))
torch.cuda.synchronize()
bad = 0
for kind, i, x, xc, y, yc in rec:
    ex, ey = not torch.equal(x, xc), not torch.equal(y, yc)
    if ex or ey:
        bad += 1
        print("CORRUPT", kind, i, tuple(x.shape), tuple(y.shape), "x" if ex else "", "y" if ey else "",
              int((x != xc).sum()), int((y != yc).sum()))
print("records", len(rec), "corrupted", bad)
