"""One DiffNet residual layer: the one-launch kernel (csrc/diffnet_layer.hip) against the two launches it replaces,
same box, same process, interleaved rounds (us per layer and TFLOP/s of the 2 x 256 x (3 x 512 + 512) FLOP per frame).
  python tools/bench_diffnet_layer.py [B T]      default: the training bucket (19 x 1550) and the sampler batch (32 x 411)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from promptttspp_amd import functional as PF  # noqa: E402
from promptttspp_amd import ops  # noqa: E402

C = 256
dev = torch.device("cuda:0")


def case(B, T, dil, save, masked):
    g = torch.Generator().manual_seed(3)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    x = r(B, T, C).bfloat16()
    yin = r(B, T, C).bfloat16()
    cond_all = r(B, T, 20 * 2 * C, sc=0.7).bfloat16()
    cond = cond_all[:, :, 3 * 2 * C:4 * 2 * C]
    dil_w, dil_b, out_w, out_b = r(2 * C, C, 3, sc=0.04), r(2 * C, sc=0.1), r(2 * C, C, 1, sc=0.06), r(2 * C, sc=0.1)
    perm = PF._gate_perm(2 * C, dev)
    wp2 = ops.pack_conv_weight(dil_w, torch.bfloat16, 2)
    wo = ops.pack_conv_weight(out_w, torch.bfloat16)
    ws = ops.diffnet_pack_wstream([wp2], [wo], C)
    bp = dil_b[perm].contiguous()
    dnext = r(B, C).float()
    skip = torch.zeros(B, T, C, device=dev)
    lengths = torch.tensor([T - (7 * i) % 60 for i in range(B)], device=dev, dtype=torch.int32) if masked else None
    gbuf = torch.empty_like(x)
    abuf = torch.empty((B, T, 2 * C), device=dev, dtype=torch.bfloat16)

    def two():
        if save:
            ops.conv1d_gate_fwd_save(yin, wp2, bp, C, 3, dil, dil, cond, gbuf, abuf, lengths=lengths)
        else:
            ops.conv1d(yin, wp2, bp, 2 * C, ks=3, dil=dil, pad=dil, act="gate", res=cond, out=gbuf)
        ops.conv1d_diffnet_post(gbuf, wo, out_b, x, skip, dnext, init=False, lengths=lengths, out_mask=lengths is not None)

    def one():
        ops.diffnet_layer_fwd(yin, x, cond, ws[0], bp, out_b, dnext, skip, dil, False, lengths=lengths, save=save)

    # the form the training step and the sampler launch: conditioner projection inside (80-stage stream)
    condx = r(B, T, 256).bfloat16()
    cwp = ops.pack_conv_weight(r(2 * C, 256, 1, sc=0.05), torch.bfloat16, 2)
    wsc = ops.diffnet_pack_wstream([wp2], [wo], C, cond_wps=[cwp])

    def one_cond():
        ops.diffnet_layer_fwd(yin, x, None, wsc[0], bp, out_b, dnext, skip, dil, False, lengths=lengths, save=save, condx=condx)

    one.cond = one_cond

    def dbg(mode, stamps):
        import ctypes

        from promptttspp_amd import _lib

        xn, yn = torch.empty_like(x), torch.empty_like(x)
        args = _lib.DiffNetLayerArgs()
        args.yin, args.x, args.cond, args.wstream = yin.data_ptr(), x.data_ptr(), cond.data_ptr(), ws[0].data_ptr()
        args.dil_b, args.out_b, args.skip, args.xn = bp.data_ptr(), out_b.data_ptr(), skip.data_ptr(), xn.data_ptr()
        args.dnext, args.yin_next = dnext.data_ptr(), yn.data_ptr()
        args.a_out = abuf.data_ptr() if save else None
        args.g_out = gbuf.data_ptr() if save else None
        args.lengths = ops.i32(lengths, dev).data_ptr() if lengths is not None else None
        args.B, args.T, args.C, args.dil, args.ldc, args.init, args.dtype = B, T, C, dil, cond.stride(1), 0, ops.dtype_code(x.dtype)
        _lib.check(_lib.load().ptpp_diffnet_layer_fwd_dbg(ctypes.byref(args), mode, stamps.data_ptr(), ops._stream()), "dbg")

    one.dbg = dbg
    return two, one


def phases(B, T, dil, save, masked):
    """per-block clock stamps of the one-launch kernel, with parts of its work switched off"""
    _, one = case(B, T, dil, save, masked)
    nblk = B * ((T + 127) // 128)
    if B * ((T + 63) // 64) <= 256:
        nblk = B * ((T + 63) // 64)
    names = {1: "full", 3: "no MFMA", 5: "no weight stream", 7: "no MFMA, no weights", 9: "no epilogue traffic", 15: "nothing but barriers + x windows", 33: "no s_barrier", 65: "no fragment LDS reads", 47: "nothing, no s_barrier",
             79: "nothing, no LDS reads", 111: "nothing, no barrier, no LDS reads", 143: "nothing, W reads in 128-B-row pattern",
             129: "full, W reads in 128-B-row pattern",
             201: "1x8 global-weights form: full", 203: "1x8 GW: no MFMA", 209: "1x8 GW: no epilogue traffic", 211: "1x8 GW: no MFMA, no epilogue traffic"}
    print(f"-- phases, B {B} T {T} dil {dil} {'train' if save else 'infer'} ({nblk} blocks): us  prologue | dilated conv | gate epilogue | projection | tail || block, launch")
    for mode, name in names.items():
        if mode >= 200 and nblk != B * ((T + 127) // 128):
            continue
        st = torch.zeros((nblk, 6), device=dev, dtype=torch.int64)
        try:
            for _ in range(3):
                one.dbg(mode, st)
        except Exception:  # (not every mode is built for the 64-row instantiation)
            continue
        torch.cuda.synchronize()
        t = st.cpu().double() / 100.0  # 100 MHz -> us
        d = (t[:, 1:] - t[:, :-1]).median(dim=0).values
        blk = (t[:, 5] - t[:, 0]).median()
        span = t[:, 5].max() - t[:, 0].min()
        print(f"   {name:34s} {d[0]:6.1f} | {d[1]:6.1f} | {d[2]:6.1f} | {d[3]:6.1f} | {d[4]:6.1f} || {blk:6.1f}, {span:6.1f}", flush=True)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def fine(B, T, dil, save, masked, mode=17):
    """s_memtime stamps inside steps 16..19 of waves 0 and 5: cycles since the step's first stamp"""
    _, one = case(B, T, dil, save, masked)
    nblk = B * ((T + 127) // 128)
    st = torch.zeros((nblk * 6 + nblk * 64,), device=dev, dtype=torch.int64)
    for _ in range(3):
        one.dbg(mode, st)
    torch.cuda.synchronize()
    f = st[nblk * 6:].view(nblk, 2, 4, 8)[:, :, :, :6].cpu().double()
    base = f[:, 0:1, 0:1, 0:1]
    rel = (f - base).median(dim=0).values  # (wave, step, stamp)
    print(f"-- fine stamps (mode {mode}), B {B} T {T}: cycles from wave 0's arrival at step 16; columns: arrive | waited | barrier passed | DMA issued | LDS requested | MFMAs issued")
    for w in range(2):
        for s_ in range(4):
            print(f"   wave {0 if w == 0 else 5} step {16 + s_}: " + " ".join(f"{rel[w, s_, k]:7.0f}" for k in range(6)))


def chains(B, T, save, L=20, delay_us=30.0):
    """L layer launches in sequence (one chain over the whole batch) against TWO half-batch chains on two streams, the second
    one started ``delay_us`` later: every block of a launch walks the same phases in lock step (MFMA passes, then HBM-bound
    epilogues), so two chains half a layer apart keep both the matrix cores and the memory system busy."""
    def mk(Bh):
        two, one = case(Bh, T, 8, save, False)
        return one
    full = mk(B)
    h1, h2 = mk(B // 2), mk(B - B // 2)
    s2 = torch.cuda.Stream()

    def run_full():
        for _ in range(L):
            full()

    def run_split(delay):
        main = torch.cuda.current_stream()
        s2.wait_stream(main)
        with torch.cuda.stream(s2):
            if delay > 0:
                torch.cuda._sleep(int(delay * 2100))  # cycles at ~2.1 GHz
            from promptttspp_amd import ops as _o
            with _o.unpinned():
                for _ in range(L):
                    h2()
        for _ in range(L):
            h1()
        main.wait_stream(s2)

    for name, f in (("one chain", run_full), ("two half-batch chains, no delay", lambda: run_split(0.0)),
                    (f"two half-batch chains, second delayed {delay_us:.0f} us", lambda: run_split(delay_us)),
                    (f"two half-batch chains, second delayed {2 * delay_us:.0f} us", lambda: run_split(2 * delay_us))):
        ts = []
        for _ in range(5):
            f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / L)
        print(f"B {B} T {T} {'train' if save else 'infer'} {name:52s}: {min(ts):7.1f} us per layer", flush=True)


def main():
    if os.environ.get("PTPP_BENCH_CHAINS"):
        chains(19, 1550, True)
        chains(19, 1550, False)
        chains(32, 590, False)
        return
    if os.environ.get("PTPP_BENCH_FINE"):
        fine(19, 1550, 8, False, False, 17)
        fine(19, 1550, 8, False, False, 23)
        return
    if os.environ.get("PTPP_BENCH_PHASES"):
        phases(19, 1550, 8, True, True)
        phases(19, 1550, 8, False, False)
        phases(32, 411, 8, False, False)
        return
    shapes = [(19, 1550), (32, 411)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]))]
    for B, T in shapes:
        flop = 2.0 * B * T * C * (3 * 2 * C + 2 * C)
        for save, masked in ((True, True), (False, False)):
            for dil in (1, 8):
                two, one = case(B, T, dil, save, masked)
                res = {"two": [], "one": [], "cond": []}
                for _ in range(5):
                    res["two"].append(timeit(two))
                    res["one"].append(timeit(one))
                    res["cond"].append(timeit(one.cond))
                t2, t1 = min(res["two"]), min(res["one"])
                print(f"B {B:3d} T {T:5d} dil {dil} {'train' if save else 'infer'}: two launches {t2:7.1f} us ({flop / t2 * 1e-6:6.1f} TF/s)"
                      f"   one launch {t1:7.1f} us ({flop / t1 * 1e-6:6.1f} TF/s)   x{t2 / t1:.2f}"
                      f"   with the conditioner inside {min(res['cond']):7.1f} us", flush=True)


if __name__ == "__main__":
    main()
