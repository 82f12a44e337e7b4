import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
warnings.simplefilter("ignore")
import torch
from conftest import load_golden, rel_err
import test_hip_acoustic as T
from promptttspp_amd import config
dev = torch.device("cuda:0")
with config.use_dtype(torch.float32):
    m, g = T._model(dev)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        for a in ("dropout_rate", "positional_dropout_rate", "p_dropout", "p"):
            if isinstance(getattr(mod, a, None), float):
                setattr(mod, a, 0.0)
    m.train()
    m.decoder.injected = {"t": g["t"], "noise": g["noise"]}
    out = m(T._batch(g, dev))
    print({k: (float(v), float(g["tr_" + k])) for k, v in out.items()})
    out["loss"].backward()
    params = dict(m.named_parameters())
    for key in [k for k in g if k.startswith("g:")]:
        name = key[2:]
        gr = params[name].grad.detach().cpu()
        if gr.numel() > 70000:
            gr = gr.flatten()[:: max(1, gr.numel() // 4096)][:4096]
        print(f"{rel_err(gr, g[key].reshape(gr.shape)):.3e}  {name}")
