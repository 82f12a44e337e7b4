"""micro-benchmark of ptpp_conv1d_wgrad / ptpp_conv1d_fwd on the training step's dominant shapes.
Calls the C ABI directly in a tight loop (pre-built argument blocks) so GPU time, not Python, is measured."""
import ctypes
import sys
import torch
sys.path.insert(0, "/root/repo")
from promptttspp_amd import _lib, ops
from promptttspp_amd.ops import ConvArgs, _ptr, _stream

dev = torch.device("cuda:0")
B, T = 19, 1580  # ~30000 rows, like one bench batch
shapes = [  # (Cin, Cout, ks, dil)
    (256, 512, 3, 1), (256, 512, 3, 4), (256, 512, 1, 1), (512, 256, 1, 1), (256, 256, 1, 1), (80, 256, 1, 1),
    (256, 80, 1, 1), (256, 1024, 1, 1), (1024, 256, 1, 1), (256, 256, 3, 1), (512, 256, 3, 2), (256, 256, 5, 1),
    (256, 10240, 1, 1),
]
which = sys.argv[1] if len(sys.argv) > 1 else "wgrad"
only = int(sys.argv[2]) if len(sys.argv) > 2 else -1
lib = _lib.load()
import os
ws = None if os.environ.get("NO_WS") else ops.workspace(dev)
for si, (cin, cout, ks, dil) in enumerate(shapes):
    if only >= 0 and si != only:
        continue
    x = torch.randn(B, T, cin, device=dev).bfloat16()
    dy = torch.randn(B, T, cout, device=dev).bfloat16()
    pad = dil * (ks - 1) // 2
    st = _stream()
    if which == "wgrad":
        dw = torch.zeros(cout, cin, ks, device=dev); db = torch.zeros(cout, device=dev)
        ops.conv1d_wgrad(x, dy, cin, cout, ks, dil, pad, dw_out=dw, db_out=db)
        j = ks - 1; sh = j * dil - pad
        xs = torch.zeros_like(x)
        if sh > 0: xs[:, : T - sh] = x[:, sh:]
        elif sh < 0: xs[:, -sh:] = x[:, : T + sh]
        else: xs = x
        ref = torch.einsum("btc,btd->cd", dy.float(), xs.float())
        err = float((dw[:, :, j] - ref).abs().max() / ref.abs().max())
        bref = dy.float().sum((0, 1))
        berr = float((db - bref).abs().max() / bref.abs().max())
        f = lambda: lib.ptpp_conv1d_wgrad(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), None, B, T, cin, cout, ks, dil, pad,
                                          cin, cout, 0, 1, _ptr(ws), ws.numel() if ws is not None else 0, st)
    else:
        w = torch.randn(cout, cin, ks, device=dev) * 0.05
        bias = torch.randn(cout, device=dev)
        wp = ops.pack_conv_weight(w, torch.bfloat16)
        y = ops.conv1d(x, wp, bias, cout, ks=ks, dil=dil, pad=pad)
        ref = torch.nn.functional.conv1d(x[:2].float().transpose(1, 2), w.bfloat16().float(), bias, padding=pad, dilation=dil).transpose(1, 2)
        err = float((y[:2].float() - ref).abs().max() / ref.abs().max()); berr = 0.0
        a = ConvArgs()
        a.x, a.wp, a.y, a.bias = x.data_ptr(), wp.data_ptr(), y.data_ptr(), bias.data_ptr()
        a.B, a.T, a.Cin, a.Cout, a.ks, a.dil, a.pad = B, T, cin, cout, ks, dil, pad
        a.ldx, a.ldy, a.out_scale, a.dtype = cin, cout, 1.0, 1
        f = lambda: lib.ptpp_conv1d_fwd(ctypes.byref(a), st)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / n
    fl = 2.0 * B * T * cin * cout * ks
    print(f"{which} cin={cin:5d} cout={cout:5d} ks={ks} dil={dil}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  err={err:.1e} berr={berr:.1e}", flush=True)
