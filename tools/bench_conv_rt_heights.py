import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from promptttspp_amd import ops
dev = torch.device("cuda:0"); ops.CONV_RT_MIN_ROWS = 1
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
torch.manual_seed(0)
for B, T, cin, ks, dil in ((65, 459, 256, 17, 1), (65, 459, 512, 3, 2), (33, 904, 256, 17, 1), (76, 392, 256, 5, 1), (19, 1550, 256, 17, 1)):
    x = torch.randn(B, T, cin, device=dev).bfloat16()
    w = torch.randn(256, cin, ks, device=dev) * (cin * ks) ** -0.5
    b = torch.zeros(256, device=dev)
    ws = ops.pack_conv_weight(w, torch.bfloat16, 3)
    y = torch.empty(B, T, 256, device=dev, dtype=torch.bfloat16)
    pad = dil * (ks - 1) // 2
    f = lambda: ops.conv1d(x, None, b, 256, ks=ks, dil=dil, pad=pad, wstream=ws, out=y)
    res = {}
    for bm in ("160", "128", "96", ""):
        if bm: os.environ["PTPP_CONV_RT_BM"] = bm
        else: os.environ.pop("PTPP_CONV_RT_BM", None)
        res[bm or "auto"] = min(timeit(f) for _ in range(3))
    print(f"B {B} T {T} cin {cin} ks {ks}: " + "  ".join(f"bm {k}: {v:6.1f} us" for k, v in res.items()), "  blocks@128", B * ((T + 127) // 128), "@160", B * ((T + 159) // 160))
