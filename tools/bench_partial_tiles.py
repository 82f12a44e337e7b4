import os, sys, torch
sys.path.insert(0, "/root/repo")
from promptttspp_amd import ops
dev = torch.device("cuda:0")
print("PTPP_CONV_RULE2", os.environ.get("PTPP_CONV_RULE2"))
for (B, T) in [(96, 310), (76, 392), (83, 360), (59, 505), (47, 637)]:
    for (cin, cout, ks, dil) in [(256, 512, 3, 2), (256, 512, 1, 1), (512, 256, 1, 1)]:
        x = torch.randn(B, T, cin, device=dev).bfloat16()
        w = torch.randn(cout, cin, ks, device=dev) * 0.02
        wp = ops.pack_conv_weight(w, torch.bfloat16)
        y = torch.empty(B, T, cout, device=dev, dtype=torch.bfloat16)
        f = lambda: ops.conv1d(x, wp, None, cout, ks=ks, dil=dil, pad=dil * (ks // 2), out=y)
        for _ in range(3): f()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): f()
        e.record(); torch.cuda.synchronize()
        print(f"B={B} T={T} {cin}->{cout} k={ks}: {a.elapsed_time(e) / 20 * 1e3:7.1f} us")
