import sys, time, torch
sys.path.insert(0, "/root/repo")
from oracle.fill import fill_state_dict
from promptttspp_amd.vocoders import BigVGAN
dev = torch.device("cuda:0")
m = BigVGAN(80, 512, [6, 5, 4, 2], [12, 10, 8, 4], [3, 7, 11], [[1, 3, 5]] * 3)
fill_state_dict(m, seed=5, overrides={"weight_g": 0.4})
m = m.to(dev).eval().set_compute_dtype(torch.bfloat16)
for B, T in ((1, 500), (8, 800)):
    x = torch.clamp(-5.5 + 2.1 * torch.randn(B, 80, T, device=dev), -11.5, 2.0)
    for par in (False, True):
        m.parallel_blocks = par
        for _ in range(3): y = m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): y = m(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"B={B} T={T} parallel={par}: {1e3*dt:.2f} ms  RTF {dt/(B*T*0.01):.5f}", flush=True)
