"""overfit one batch: every loss term must fall (sanity of the whole backward + optimiser path)"""
import sys, torch
sys.path.insert(0, "/root/repo")
import bench
from promptttspp_amd import config
from promptttspp_amd.optim import FusedAdamW
from promptttspp_amd.parallel import FlatGradReducer
dev = torch.device("cuda:0")
config.set_compute_dtype(torch.bfloat16 if len(sys.argv) < 2 else torch.float32)
model = bench.build_model(dev).train()
batch = bench.make_batches(0, 1, 1, 8000, dev)[0]
params = [p for p in model.parameters() if p.requires_grad]
red = FlatGradReducer(params); opt = FusedAdamW(params, lr=3e-4, betas=(0.9, 0.98), weight_decay=0.0, max_grad_norm=1.0)
dn = model.decoder.denoise_fn
for it in range(201):
    red.zero_grad(); out = model(batch); out["loss"].backward(); red.finish()
    if it % 50 == 0:
        print(it, {k: round(float(v), 4) for k, v in out.items()},
              "| out_proj w %.3e g %.3e | skip_proj g %.3e | layer19 out g %.3e | in_proj g %.3e | gnorm %.3f" % (
                  float(dn.output_projection.weight.norm()), float(dn.output_projection.weight.grad.norm()),
                  float(dn.skip_projection.weight.grad.norm()), float(dn.residual_layers[19].output_projection.weight.grad.norm()),
                  float(dn.input_projection.weight.grad.norm()), float(sum(p.grad.double().pow(2).sum() for p in params) ** 0.5)), flush=True)
    opt.step()
