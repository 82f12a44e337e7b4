import sys, torch
sys.path.insert(0, "/root/repo")
from promptttspp_amd import ops, functional as PF
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, T) in [(2, 64), (3, 90), (1, 128), (2, 65)]:
    C, M = 256, 80
    s = (torch.randn(B, T, C, device=dev) * 0.5).bfloat16()
    ws, bs = torch.randn(C, C, 1, device=dev) * 0.06, torch.randn(C, device=dev) * 0.1
    wo, bo = torch.randn(M, C, 1, device=dev) * 0.06, torch.randn(M, device=dev) * 0.1
    wi, bi = torch.randn(C, M, 1, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
    x = torch.randn(B, T, M, device=dev)
    noise = torch.randn(B, T, M, device=dev)
    t = torch.full((B,), 37, device=dev, dtype=torch.long)
    K = 100
    tabs = [torch.rand(K, device=dev) + 0.5 for _ in range(4)] + [-torch.rand(K, device=dev)]
    ds0 = torch.randn(B, C, device=dev)
    dt = torch.bfloat16
    h = ops.conv1d(s, ops.pack_conv_weight(ws, dt), bs, C, act="relu")
    eps = ops.conv1d(h, ops.pack_conv_weight(wo, dt), bo, M)
    x1 = ops.ddpm_step(x, eps.contiguous(), noise, t, *tabs)
    h0 = ops.conv1d(x1.to(dt), ops.pack_conv_weight(wi, dt), bi, C, act="relu")
    _, yin0 = ops.diffnet_post_fwd(None, h0, None, ds0, init=True)
    gx, gh0, gy = ops.sampler_head(s, ops.pack_conv_weight(ws, dt), bs, ops.pack_conv_weight(wo, dt), bo, x, noise, t, *tabs,
                                   win_p=ops.pack_conv_weight(wi, dt), win_b=bi, ds0=ds0)
    torch.cuda.synchronize()
    # stage-wise: recompute eps from the fused x? compare outputs
    print(B, T, "x equal", torch.equal(gx, x1), float((gx - x1).abs().max()), "h0", torch.equal(gh0, h0), float((gh0.float() - h0.float()).abs().max()),
          "yin0", torch.equal(gy, yin0), float((gy.float() - yin0.float()).abs().max()))
    # which stage: h via a head call with identity? compare eps through x: if x equal then GEMM1+2 fine
    # exact IEEE sequence in torch f32 (no contraction): which kernel deviates?
    tb = 37
    a, bq, k1, k2 = tabs[0][tb], tabs[1][tb], tabs[2][tb], tabs[3][tb]
    sg = torch.exp(0.5 * tabs[4][tb])
    ev = eps.float()
    x0 = (a * x - bq * ev).clamp(-1, 1)
    ref = (k1 * x0 + k2 * x) + sg * noise
    print("   vs torch: launches", float((x1 - ref).abs().max()), int((x1 != ref).sum()), " head", float((gx - ref).abs().max()), int((gx != ref).sum()),
          " sg", float(sg), " eps equal through h?", torch.equal(eps, eps))
    def tm(f, n=50):
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            f()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    wsp, wop, wip = ops.pack_conv_weight(ws, dt), ops.pack_conv_weight(wo, dt), ops.pack_conv_weight(wi, dt)
    def seven():
        h = ops.conv1d(s, wsp, bs, C, act="relu")
        eps = ops.conv1d(h, wop, bo, M)
        x1 = ops.ddpm_step(x, eps, noise, t, *tabs)
        h0 = ops.conv1d(x1.to(dt), wip, bi, C, act="relu")
        ops.diffnet_post_fwd(None, h0, None, ds0, init=True)
    print("   seven launches %.1f us, head %.1f us" % (tm(seven), tm(lambda: ops.sampler_head(s, wsp, bs, wop, bo, x, noise, t, *tabs, win_p=wip, win_b=bi, ds0=ds0))))
