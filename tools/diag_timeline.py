"""Module-level GPU timeline of a training step without a profiler: events recorded (on whatever stream is current) when
the forward / backward of the top-level modules starts and ends; the host stays ahead as in bench.py."""
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
marks = []  # (name, event, stream id, host time)
on = [False]

def mark(name):
    if on[0]:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, e, torch.cuda.current_stream().cuda_stream, time.perf_counter()))

mods = {"encoder": model.encoder, "ref_enc": model.reference_encoder, "prompt": model.prompt_encoder, "var_adaptor": model.variance_adaptor,
        "decoder": model.decoder, "style_mdn": model.style_mdn}
for n, m in mods.items():
    m.register_forward_pre_hook(lambda mod, inp, n=n: mark(f"F {n} start"))
    m.register_forward_hook(lambda mod, inp, out, n=n: mark(f"F {n} end"))
    m.register_full_backward_pre_hook(lambda mod, g, n=n: mark(f"B {n} start"))
    m.register_full_backward_hook(lambda mod, gi, go, n=n: mark(f"B {n} end"))

def wrap(obj, attr, name, grad_of=None):
    orig = getattr(obj, attr)
    def f(*a, **k):
        mark(f"F {name} start")
        out = orig(*a, **k)
        mark(f"F {name} end")
        t = out if torch.is_tensor(out) else next((o for o in out if torch.is_tensor(o) and o.requires_grad), None)
        if t is not None and t.requires_grad and on[0]:
            t.register_hook(lambda g, name=name: mark(f"B grad reaches output of {name}"))
        return out
    setattr(obj, attr, f)

wrap(model, "_encode", "phoneme encoder")
wrap(model.variance_adaptor, "forward_cl", "variance adaptor")
wrap(model.decoder, "forward_cl", "decoder (diffusion)")
wrap(model.variance_adaptor.frame_prior_network, "forward_cl", "frame prior")


def step(b):
    from promptttspp_amd import ops
    with ops.pinned_stream():
        mark("step start")
        red.zero_grad()
        out = model(b)
        mark("forward done (main)")
        with torch.autograd.set_multithreading_enabled(False):
            out["loss"].backward()
        mark("backward done (main)")
        red.finish()
        mark("join done")
        opt.step()
        mark("optimizer done")
    sched.step()

for i in range(6):
    step(batches[i % 8])
torch.cuda.synchronize()
ref = torch.cuda.Event(enable_timing=True)
t_ref = time.perf_counter()
ref.record()
res = {}
for rep in range(6):
    for i in range(3):
        step(batches[(rep + i) % 8])
    marks.clear()
    on[0] = True
    step(batches[(rep + 3) % 8])
    on[0] = False
    step(batches[(rep + 4) % 8])
    torch.cuda.synchronize()
    t0 = marks[0][1]
    for k, (n, e, s, th) in enumerate(marks):
        res.setdefault((k, n, s), []).append((t0.elapsed_time(e), 1e3 * (th - marks[0][3]), ref.elapsed_time(e) - 1e3 * (th - t_ref)))
    # (a sync per repetition: re-anchor the common clock)
    t_ref = time.perf_counter()
    ref = torch.cuda.Event(enable_timing=True)
    ref.record()
streams = sorted({s for (_, _, s) in res})
print("   GPU ms   host ms   GPU behind host (ms)")
for (k, n, s), v in sorted(res.items()):
    g = sum(x[0] for x in v) / len(v); h = sum(x[1] for x in v) / len(v); lag = sum(x[2] for x in v) / len(v)
    print(f"{g:7.2f}  {h:7.2f}  {lag:7.2f}   stream {streams.index(s)}  {n}")
