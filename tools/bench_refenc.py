import time, torch, torch.nn as nn
dev = torch.device("cuda:0")
def stack():
    convs = []
    ch = [1, 128, 128, 256, 256, 512, 512]
    for i in range(6):
        convs += [nn.Conv2d(ch[i], ch[i + 1], 3, stride=2, padding=1, bias=False), nn.BatchNorm2d(ch[i + 1]), nn.ReLU(inplace=True)]
    return nn.Sequential(*convs).to(dev)
x = torch.randn(41, 1, 725, 80, device=dev)
for name, dt, cl in (("f32 nchw", torch.float32, False), ("f32 nhwc", torch.float32, True), ("bf16 nchw", torch.bfloat16, False), ("bf16 nhwc", torch.bfloat16, True)):
    m = stack()
    if cl: m = m.to(memory_format=torch.channels_last)
    xi = x.to(memory_format=torch.channels_last) if cl else x
    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dt == torch.bfloat16):
            y = m(xi)
        y.float().sum().backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); print(name, f"{(time.perf_counter() - t0) / 5 * 1e3:.2f} ms fwd+bwd")
# depthwise conv1d (conformer conv module)
d = nn.Conv1d(256, 256, 7, padding=3, groups=256).to(dev)
h = torch.randn(61, 256, 98, device=dev, requires_grad=True)
for _ in range(3): d(h).sum().backward()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): d(h).sum().backward()
torch.cuda.synchronize(); print("depthwise f32", f"{(time.perf_counter() - t0) / 10 * 1e3:.2f} ms fwd+bwd")
g = nn.GRU(1024, 256, 1, batch_first=True).to(dev)
hs = torch.randn(41, 12, 1024, device=dev, requires_grad=True)
for _ in range(3): g(hs)[0].sum().backward()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): g(hs)[0].sum().backward()
torch.cuda.synchronize(); print("gru f32", f"{(time.perf_counter() - t0) / 10 * 1e3:.2f} ms fwd+bwd")
