"""Stand-alone timing of ptpp_dwconv1d_wgrad at phone-level size."""
import sys, torch
sys.path.insert(0, "/root/repo")
from promptttspp_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()
B, T, C, ks = 19, 200, 256, 7
lengths = torch.tensor([max(20, T - 9 * i) for i in range(B)], device=dev, dtype=torch.int32)
for dt in (torch.bfloat16, torch.float32):
    u = torch.randn(B, T, C, device=dev).to(dt)
    dy = torch.randn(B, T, C, device=dev).to(dt)
    dw = torch.zeros(C, ks, device=dev)
    db = torch.zeros(C, device=dev)
    st = ops._stream()
    f = lambda: _lib.check(lib.ptpp_dwconv1d_wgrad(u.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), lengths.data_ptr(), B, T, C, ks,
                                                    ops.dtype_code(dt), st), "dw")
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        f()
    e.record()
    torch.cuda.synchronize()
    print(dt, f"{a.elapsed_time(e) / 200 * 1e3:.1f} us per call")
