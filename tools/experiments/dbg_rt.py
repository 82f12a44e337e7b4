import sys, torch
sys.path.insert(0, "/root/repo")
from promptttspp_amd import ops
dev = torch.device("cuda:0")
for (B, T, cin, ks, dil, act, masked, use_res) in [(20, 450, 256, 17, 1, None, True, False), (20, 450, 256, 17, 1, None, False, False), (20, 450, 256, 5, 1, "relu", False, False), (9, 1000, 512, 3, 8, None, True, True)]:
    g = torch.Generator().manual_seed(1)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    x, w, b, res = r(B, T, cin).bfloat16(), r(256, cin, ks, sc=(cin * ks) ** -0.5), r(256, sc=0.1), r(B, T, 256).bfloat16()
    pad = dil * (ks - 1) // 2
    lengths = torch.tensor([max(1, T - 37 * i) for i in range(B)], device=dev, dtype=torch.int32) if masked else None
    kw = dict(ks=ks, dil=dil, pad=pad, act=act, lengths=lengths, in_mask=masked, out_mask=masked and act is not None, res=res if use_res else None, res_scale=0.7071 if use_res else 1.0)
    ref = ops.conv1d(x, ops.pack_conv_weight(w, torch.bfloat16), b, 256, **kw)
    got = ops.conv1d(x, None, b, 256, wstream=ops.pack_conv_weight(w, torch.bfloat16, 3), **kw)
    torch.cuda.synchronize()
    d = (ref.float() - got.float()).abs()
    bad = (ref != got)
    print((B, T, cin, ks, dil, act, masked), "equal", bool(torch.equal(ref, got)), "max", float(d.max()), "nbad", int(bad.sum()), "of", bad.numel())
    if bad.any():
        idx = bad.nonzero()
        print("  first bad", idx[:5].tolist(), "rows bad per b:", [int(bad[i].any(dim=1).sum()) for i in range(min(B, 6))], "t range", int(idx[:, 1].min()), int(idx[:, 1].max()), "ch range", int(idx[:, 2].min()), int(idx[:, 2].max()))
