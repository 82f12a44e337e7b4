import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from promptttspp_amd import ops
dev = torch.device("cuda:0"); ops.CONV_RT_MIN_ROWS = 1
def run(B, T, cin, ks, dil, masked):
    g = torch.Generator().manual_seed(5)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    x, w, b = r(B, T, cin).bfloat16(), r(256, cin, ks, sc=(cin * ks) ** -0.5), r(256, sc=0.1)
    pad = dil * (ks - 1) // 2
    lengths = torch.tensor([max(1, T - 37 * i) for i in range(B)], device=dev, dtype=torch.int32) if masked else None
    kw = dict(ks=ks, dil=dil, pad=pad, lengths=lengths, in_mask=masked)
    ws = ops.pack_conv_weight(w, torch.bfloat16, 3)
    os.environ["PTPP_CONV_RT_BM"] = "128"
    ref = ops.conv1d(x, None, b, 256, wstream=ws, **kw)
    os.environ["PTPP_CONV_RT_BM"] = "160"
    got = ops.conv1d(x, None, b, 256, wstream=ws, **kw)
    torch.cuda.synchronize()
    bad = (ref != got)
    print(f"B{B} T{T} cin{cin} ks{ks} dil{dil} masked{masked}: mismatches {int(bad.sum())} of {bad.numel()}, maxdiff {float((ref.float()-got.float()).abs().max()):.4f}")
    if bad.any():
        bb = bad.any(dim=2)
        for bi in range(min(B,3)):
            rows = bb[bi].nonzero().flatten()
            print("   batch", bi, "bad rows", rows[:12].tolist(), "...", rows[-6:].tolist(), "count", rows.numel(), " bad channels", int(bad[bi].any(dim=0).sum()))
run(2, 459, 256, 3, 1, False)
run(2, 459, 256, 5, 1, False)
run(2, 459, 256, 17, 1, False)
run(3, 459, 256, 17, 1, True)
run(2, 700, 512, 3, 8, True)
