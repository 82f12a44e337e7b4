#!/usr/bin/env python
"""Headline benchmark: PromptTTS++ training step (prompttts_mdn_v2_wo_erg_final, bf16,
dataset.max_tokens=30000 per GPU) on synthetic LibriTTS-R-shaped batches, plus the BigVGAN
24 kHz vocoder real-time factor, on N MI355X GPUs (one process per GPU, RCCL over xGMI).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line:  metric = mel-frames/sec (valid frames of all ranks per
second of optimiser steps: forward + backward + gradient exchange + clip + AdamW +
Noam step, train mode with dropout), `roofline` for the dominant kernel family (the
MFMA implicit-GEMM conv), `cpu_baseline` (the oracle = CPU restatement of the reference,
timed on the host cores on a bounded sample), and the BigVGAN RTF as extra keys.
"""
import argparse
import json
import math
import os

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory (promptttspp_amd/__init__.py); before the HIP runtime starts
import random
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
warnings.simplefilter("ignore")

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
# HBM traffic comes from PMC passes (bench.py cannot run rocprofv3 on itself): the committed summaries of
# `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this command / of tools/bench_vocoder.py, written by
# tools/pmc_traffic.py with the gfx950 corrections of MI355X_MICROARCH.md; the JSON line names the file it read.
TRAFFIC_TRAIN = os.path.join(ROOT, "profiles", "r06_hbm_traffic_train.json")
TRAFFIC_VOC = os.path.join(ROOT, "profiles", "r06_hbm_traffic_bigvgan.json")


def measured_traffic(path, kernel_substr=None):
    """(bytes, provenance) from a committed PMC summary: mean bytes per launch of the kernel whose name contains
    ``kernel_substr``, or the total per pass of the profiled workload when it is None; (None, reason) if absent."""
    try:
        d = json.load(open(path))
    except OSError as e:
        return None, f"{os.path.basename(path)} missing ({e.__class__.__name__})"
    src = f"profiles/{os.path.basename(path)} ({d.get('collected', 'PMC passes')}; command: {d.get('command', '?')})"
    if kernel_substr is None:
        return d.get("total_traffic_bytes_per_pass"), src
    for name, v in d["kernels"].items():
        if kernel_substr in name:
            return v["traffic_bytes"], src + f"; mean over {v['launches']} launches of {name[:70]}"
    return None, src + f"; no kernel matching {kernel_substr!r}"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preheat", type=int, default=60,
                    help="untimed steps on the warm-up batches BEFORE the W warm-up steps (a count, so that every rank runs the same "
                         "collectives): a box that comes out of idle needs ~1 s of load to reach its steady clocks (the first bench "
                         "process on a fresh box measured 3-10 %% slow in 2 of 6 trials, a second one never)")
    ap.add_argument("--max-tokens", type=int, default=30000)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-vocoder", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--voc-batch", type=int, default=64)
    ap.add_argument("--voc-frames", type=int, default=1000)
    ap.add_argument("--no-app", action="store_true", help="skip the prompt -> waveform leg (BASELINE config 5)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------
def build_model(dev):
    from promptttspp_amd import hydra_lite as H

    cfg = H.load_node(os.path.join(ROOT, "egs", "proposed", "bin", "conf", "model", "prompttts_mdn_v2_wo_erg_final.yaml"))
    from promptttspp_amd.modules.prompt_encoder import allow_random_bert

    torch.manual_seed(1234)  # identical random-init weights on every rank
    with allow_random_bert():  # random-init weights of the named architecture (no network for checkpoints)
        return H.instantiate(cfg).to(dev)


def make_batches(rank, world, n, max_tokens, dev):
    """Reference DP sharding (trainers/tts.py:122-143): pack the length-sorted corpus into
    global batches of max_tokens*W with size multiple W, shuffle with seed 42, rank r takes
    x[r::W] of each global batch -> ~max_tokens padded frames per GPU."""
    from promptttspp_amd.datasets.prompttts import PromptTTSCollator
    from promptttspp_amd.datasets.synthetic import SyntheticLibriTTSR
    from promptttspp_amd.datasets.utils import batch_by_size

    ds = SyntheticLibriTTSR(num_utts=20000, seed=1234)
    gb = batch_by_size(ds.ordered_indices(), ds.num_tokens, max_tokens=max_tokens * world, required_batch_size_multiple=world)
    gb = [b for b in gb if len(b) % world == 0 and len(b) >= world]
    random.Random(42).shuffle(gb)
    coll = PromptTTSCollator()
    out = []
    for b in gb[:n]:
        items = coll([ds[i] for i in b[rank::world]])[2:]
        items = [tuple(t.to(dev) for t in x) if isinstance(x, tuple) else x.to(dev) for x in items]
        out.append(items)
    return out


def train_setup(model, world):
    from promptttspp_amd.optim import FusedAdamW
    from promptttspp_amd.parallel import FlatGradReducer
    from promptttspp_amd.utils.lr_scheduler import NoamLR

    params = [p for p in model.parameters() if p.requires_grad]
    red = FlatGradReducer(params)
    red.broadcast_parameters(model)
    opt = FusedAdamW(params, lr=1e-3, betas=(0.9, 0.98), weight_decay=0.0, max_grad_norm=1.0)  # conf/optimizer/adamw.yaml + clip 1.0
    opt.stable_grads = True  # p.grad are views of the reducer's flat buffer for the whole run
    sched = NoamLR(opt, warmup_steps=4000)
    return red, opt, sched


def train_step(model, batch, red, opt, sched):
    from promptttspp_amd import ops

    with ops.pinned_stream():  # as trainers/tts.py does: one stream lookup per step, not one per launch
        red.zero_grad()
        out = model(batch)
        # backward on the calling thread: most nodes are Python autograd Functions, and the engine's worker
        # thread would take the GIL from this (idle) one for each of them (tools/diag_backward_thread.py)
        with torch.autograd.set_multithreading_enabled(False):
            out["loss"].backward()
        red.finish()
        opt.step()
    sched.step()
    return out


# ------------------------------------------------------------------------------------------
def box_calibration(dev):
    """What THIS box delivers on two library operations that involve none of this repository's kernels: a 2 GiB device
    copy (GB/s, read + write) and a bf16 8192^3 matmul through hipBLASLt (TFLOP/s).  The boxes of the pool differ by
    ~20 % on identical code; these two numbers let a reader tell a slow box from a slow commit."""
    n = 1 << 30
    x = torch.empty(n, device=dev, dtype=torch.bfloat16)
    y = torch.empty_like(x)
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    for _ in range(2):
        y.copy_(x)
        torch.matmul(a, b)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(10):
        y.copy_(x)
    e[1].record()
    for _ in range(10):
        torch.matmul(a, b)
    e[2].record()
    torch.cuda.synchronize()
    out = {"copy_gbs": round(10 * 2 * 2 * n / (e[0].elapsed_time(e[1]) * 1e-3) / 1e9, 1),
           "hipblaslt_bf16_8192_tflops": round(10 * 2 * 8192 ** 3 / (e[1].elapsed_time(e[2]) * 1e-3) / 1e12, 1),
           "device": torch.cuda.get_device_name(dev)}
    del x, y, a, b
    # what the hot path's own GEMM SHAPES allow on this box (library GEMM with the taps folded into K: an upper bound for a
    # conv of that shape; tools/bench_gemm_calibration.py has the full table): the DiffNet dilated conv and a phone-level FFN conv
    for key, (M, K, N) in (("hipblaslt_bf16_30000x768x512_tflops", (30000, 768, 512)),
                           ("hipblaslt_bf16_2850x2304x1024_tflops", (2850, 2304, 1024))):
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            torch.matmul(a, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            torch.matmul(a, b)
        e1.record()
        torch.cuda.synchronize()
        out[key] = round(20 * 2.0 * M * K * N / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1)
    return out


def conv_roofline(model, batch, red, opt, sched, dtype_name):
    """Instrumented extra step: HIP events around every launch of the implicit-GEMM conv
    family (forward, data-gradient) on torch's current stream; algorithmic FLOPs =
    2 * valid_rows * Cin * Cout * ks per launch."""
    from promptttspp_amd import ops

    recs = []
    fams = []  # per record: the kernel family the launch takes (for algorithmic_bytes_by_family)
    orig = ops.conv1d

    def timed(x, wp, bias, cout, ks=1, dil=1, pad=0, lengths=None, **kw):
        Bn, T = x.shape[0], x.shape[1]
        if ks == 1 and not kw.get("in_mask") and not kw.get("out_mask"):
            Bn, T = 1, Bn * T  # mask-free linear layers run over the flattened rows (ptpp_conv1d_fwd_ws)
        # the launches that take the 128 x 128-tile instantiation (tile choice of conv1d_cl.hip::launch_tiles)
        big = (x.dtype == torch.bfloat16 and cout > 64 and x.shape[2] % 64 == 0 and Bn * T > 128
               and not (T <= 96 or (T % 128 != 0 and T % 128 <= 64 and T < 512))
               and Bn * ((T + 127) // 128) * ((cout + 127) // 128) >= 192)  # smaller grids: other tiles / split-K
        if not big:
            return orig(x, wp, bias, cout, ks=ks, dil=dil, pad=pad, lengths=lengths, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = orig(x, wp, bias, cout, ks=ks, dil=dil, pad=pad, lengths=lengths, **kw)
        e1.record()
        rows = float(lengths.sum()) if lengths is not None and (kw.get("out_mask") or kw.get("in_mask")) else x.shape[0] * x.shape[1]
        es = x.element_size()
        nbytes = x.shape[0] * x.shape[1] * (x.shape[2] + (cout // 2 if kw.get("act") == "gate" else cout)
                                            + (cout if kw.get("res") is not None else 0)) * es + cout * x.shape[2] * ks * es
        recs.append((e0, e1, 2.0 * rows * x.shape[2] * cout * ks, nbytes))
        fams.append(f"conv1d_rt_gw_kernel<8,{ks},0,0>" if (kw.get("wstream") is not None and ops.conv1d_rt_ok(x, cout, ks, dil, kw.get("act"), kw.get("res2"), kw.get("drop_p", 0.0)))
                    else "conv1d_glds_kernel<2,4,2,2,2,false,0>")
        if ks == 1 and cout >= 4096 and lengths is None:
            dn_fwd.add(len(recs) - 1)  # the (B, T, L * 2C) conditioner projection of all DiffNet layers (timed path: inside the layers)
        return y

    dn_fwd = set()  # records of the DiffNet training forward on the launch-by-launch path (2 per layer + the conditioner GEMM)
    orig_post = ops.conv1d_diffnet_post

    def timed_post(g, wp, bias, x, skip, dnext, init, lengths=None, out_mask=False, **kw):
        # the DiffNet output projection with its fused tail: same kernel family, same accounting (2C output channels)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig_post(g, wp, bias, x, skip, dnext, init, lengths=lengths, out_mask=out_mask, **kw)
        e1.record()
        rows = float(lengths.sum()) if lengths is not None and out_mask else g.shape[0] * g.shape[1]
        C = x.shape[2]  # bytes: g, x in; xn, yin out (bf16); the f32 skip rows read and written; the weights
        nbytes = g.shape[0] * g.shape[1] * ((g.shape[2] + 3 * C) * 2 + 2 * C * 4) + 2 * C * g.shape[2] * 2
        recs.append((e0, e1, 2.0 * rows * g.shape[2] * 2 * C, nbytes))
        fams.append("diffnet post (launch-by-launch path)")
        dn_fwd.add(len(recs) - 1)
        return r

    orig_gbwd = ops.conv1d_gate_bwd

    def timed_gbwd(do, wpt, a, da, lengths=None):
        # the output projection's data gradient with the fused gate backward: 2C -> C channels; bytes: do and a in, da out
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig_gbwd(do, wpt, a, da, lengths=lengths)
        e1.record()
        rows, c2 = do.shape[0] * do.shape[1], do.shape[2]
        valid = float(lengths.sum()) if lengths is not None else rows
        recs.append((e0, e1, 2.0 * valid * c2 * (c2 // 2), rows * 3 * c2 * 2 + c2 * (c2 // 2) * 2))
        fams.append("conv1d_rt_gw_kernel<8,1,0,1> (gate backward)")
        return r

    orig_gsave = ops.conv1d_gate_fwd_save

    def timed_gsave(x, wp, bias, C, ks, dil, pad, res, g, a, lengths=None):
        # the dilated conv with the gate in its epilogue and the pre-activation kept: C -> 2C channels; bytes: x and the
        # conditioner slice in, a (2C) and g (C) out
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig_gsave(x, wp, bias, C, ks, dil, pad, res, g, a, lengths=lengths)
        e1.record()
        rows = x.shape[0] * x.shape[1]
        valid = float(lengths.sum()) if lengths is not None else rows
        recs.append((e0, e1, 2.0 * valid * x.shape[2] * 2 * C * ks, rows * (x.shape[2] + 2 * C + 2 * C + C) * 2 + 2 * C * x.shape[2] * ks * 2))
        fams.append("diffnet gate conv (launch-by-launch path)")
        dn_fwd.add(len(recs) - 1)
        return r

    from promptttspp_amd import functional as PF

    ops.conv1d = timed
    ops.conv1d_diffnet_post = timed_post
    ops.conv1d_gate_bwd = timed_gbwd
    ops.conv1d_gate_fwd_save = timed_gsave
    # the timed steps issue whole stacks through the C-side drivers (one call per DiffNet stack / predictor stack); the
    # instrumented step takes the per-launch path -- the same kernels with the same arguments in the same order (bit-identical,
    # tests/test_stack_drivers.py) -- so that every launch can be bracketed by its own pair of events
    drivers = PF.STACK_DRIVERS
    PF.STACK_DRIVERS = False
    try:
        # keep the device busy while the host enqueues the step, so that the events bracket kernel
        # execution and not launch gaps (the instrumented step is host-bound)
        torch.cuda._sleep(int(0.08 * 2.4e9))
        train_step(model, batch, red, opt, sched)
        torch.cuda.synchronize()
    finally:
        PF.STACK_DRIVERS = drivers
        ops.conv1d = orig
        ops.conv1d_diffnet_post = orig_post
        ops.conv1d_gate_bwd = orig_gbwd
        ops.conv1d_gate_fwd_save = orig_gsave
    # the dominant kernel = the LDS-DMA conv kernel these launches take (csrc/conv1d_glds.h; rocprof:
    # conv1d_glds_kernel<2, 4, 2, 2, 2>, 64 x 128 tiles, and <4, 4, 2, 2, 2>, 128 x 128 tiles, for the few launches
    # with >= 1536 tiles; profiles/r04_train_step.md is the rocprofv3 summary of the training leg of this command);
    # launches of the conv family with smaller tiles / split-K are listed there, not averaged in here
    ridge = MFMA_BF16_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)

    def summarize(entries):
        """entries: (ms, flop, algorithmic bytes) per launch -> (total ms, total flop, by_bound): the launches split by the roof
        that applies to each (arithmetic intensity of its ALGORITHMIC bytes against the 2.5 PF / 8 TB/s ridge of 312 FLOP/B):
        the 1 x 1 projections (64-100 FLOP/B) can never approach the MFMA peak."""
        split = {}
        for ms, f, nb in entries:
            k = "mfma" if f / nb >= ridge else "hbm"
            d = split.setdefault(k, {"launches": 0, "ms": 0.0, "flop": 0.0, "bytes": 0.0})
            d["launches"] += 1; d["ms"] += ms; d["flop"] += f; d["bytes"] += nb
        out = {}
        for k, d in split.items():
            if k == "mfma":
                a_ = d["flop"] / (d["ms"] * 1e-3) / 1e12
                out[k] = {"launches": d["launches"], "achieved": round(a_, 1), "unit": "TFLOP/s",
                          "frac": round(a_ / MFMA_BF16_PEAK_TFLOPS, 4), "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2)}
            else:
                a_ = d["bytes"] / (d["ms"] * 1e-3) / 1e9
                out[k] = {"launches": d["launches"], "achieved": round(a_, 1), "unit": "GB/s (algorithmic)",
                          "frac": round(a_ / HBM_PEAK_GBS, 4), "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2)}
        return sum(e[0] for e in entries), sum(e[1] for e in entries), out

    all_entries = [(a.elapsed_time(b), f, nb) for a, b, f, nb in recs]
    tot_ms, tot_flop, by_bound = summarize(all_entries)
    ach = tot_flop / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
    peak = MFMA_BF16_PEAK_TFLOPS if dtype_name == "bf16" else 157.3
    traffic, traffic_src = measured_traffic(TRAFFIC_TRAIN, "conv1d_glds_kernel<2, 4, 2, 2, 2") \
        if dtype_name == "bf16" else (None, "no PMC pass for the f32 mode")
    kname = "conv1d_glds_kernel / conv1d_rt_gw_kernel / diffnet_layer_kernel <bf16> (implicit-GEMM conv family: LDS-DMA 64x128 / 128x128 " \
            "tiles; row tiles with the weight fragments straight from global memory on a 1 x 8 wave grid for the 256-channel layers " \
            "and the one-launch DiffNet layer)" if dtype_name == "bf16" else "conv1d_cl_kernel<f32>, 128x128 tiles"
    # The timed steps run the DiffNet forward as ONE launch per layer (csrc/diffnet_layer.hip, issued by the C-side stack
    # driver: the launch-by-launch step above takes the two launches it replaces).  One more step through the drivers with an
    # event pair around the driver call: 20 launches of that kernel (+ one elementwise launch), priced per launch.
    layer = None
    if dtype_name == "bf16" and drivers:
        lrec = []
        orig_drv = PF._diffnet_stack_forward_driver

        def timed_drv(h0, cond_all, dsteps, weights, lengths, *rest, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig_drv(h0, cond_all, dsteps, weights, lengths, *rest, **kw)
            e1.record()
            # (no host read here: the step stays fully enqueued behind the spin kernel -- tools/prof_step_tail.py reads this step as
            #  the device's own timeline; the valid-row count is taken after the synchronize below)
            rows = lengths if lengths is not None else h0.shape[0] * h0.shape[1]
            cx = kw.get("condx")
            lrec.append((e0, e1, len(weights), rows, h0.shape[0] * h0.shape[1], h0.shape[2], 0 if cx is None else cx.shape[2]))
            return r

        PF._diffnet_stack_forward_driver = timed_drv
        try:
            torch.cuda._sleep(int(0.08 * 2.4e9))
            train_step(model, batch, red, opt, sched)
            torch.cuda.synchronize()
        finally:
            PF._diffnet_stack_forward_driver = orig_drv
        if lrec:
            e0, e1, L, rows, padded, C, Cc = lrec[0]
            rows = float(rows.sum()) if torch.is_tensor(rows) else rows
            us = 1e3 * e0.elapsed_time(e1) / L
            # dilated conv k3 C -> 2C + output projection C -> 2C (+ the conditioner projection Cc -> 2C where the layer does it)
            flop = 2.0 * rows * C * (3 * 2 * C + 2 * C) + 2.0 * rows * Cc * 2 * C
            # algorithmic bytes per launch: yin, x, conditioner (its 2C slice, or the Cc-channel input) in; a (2C), g, xn, yin' out
            # (bf16); skip f32 read + written
            nbytes = padded * C * (2 + 2 + 4 + 2 + 2 + 2) + padded * (2 * Cc if Cc else 4 * C) + padded * C * 8
            layer_flop, layer_bytes = flop, nbytes
            lay_tf = flop / (us * 1e-6) / 1e12
            layer = {"kernel": "diffnet_layer_kernel (one launch per DiffNet residual layer, training forward"
                               + (", conditioner projection inside)" if Cc else ")"), "launches": L,
                     "avg_launch_us": round(us, 2), "achieved": round(lay_tf, 1), "unit": "TFLOP/s",
                     "frac": round(lay_tf / MFMA_BF16_PEAK_TFLOPS, 4),
                     "algorithmic_gbs": round(nbytes / (us * 1e-6) / 1e9, 1), "hbm_frac": round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                     "note": "event pair around ptpp_diffnet_stack_fwd of a driver-path step / layers; both roofs are quoted: the "
                             "launch alternates matrix passes with HBM-bound epilogues (DESIGN.md sections 5e, 5f.2)"}
    alg_by_fam = {}
    for (e0_, e1_, f_, nb_), fam in zip(recs, fams):
        d_ = alg_by_fam.setdefault(fam, [0, 0.0])
        d_[0] += 1
        d_[1] += nb_
    alg_by_fam = {k: {"launches": v[0], "mean_bytes_per_launch": round(v[1] / v[0], 1)} for k, v in alg_by_fam.items()}
    if layer is not None:
        alg_by_fam["diffnet_layer_kernel<5,true,0,8,true,true,2>"] = {"launches": layer["launches"], "mean_bytes_per_launch": round(layer_bytes, 1)}
    per_launch = {"achieved": round(ach, 2), "frac": round(ach / peak, 4), "launches": len(recs), "by_bound": by_bound,
                  "note": "every launch of the instrumented step as issued there: the DiffNet forward as two launches per layer + one "
                          "(B, T, L * 2C) conditioner GEMM (the form of rounds 1-3, kept for continuity)"}
    if layer is not None and dn_fwd:
        # AS TIMED: the steps `value` is measured on run the DiffNet forward as ONE launch per layer (conditioner projection
        # inside) -- those launches, measured above through the driver, take the place of the launch-by-launch records
        ent = [e for i, e in enumerate(all_entries) if i not in dn_fwd] + [(layer["avg_launch_us"] * 1e-3, layer_flop, layer_bytes)] * layer["launches"]
        tot_ms, tot_flop, by_bound = summarize(ent)
        ach = tot_flop / (tot_ms * 1e-3) / 1e12
        n_launch = len(ent)
    else:
        n_launch = len(recs)
    return {"bound": "mfma", "kernel": kname + ": frame-level fwd + dgrad launches of one step",
            "diffnet_layer": layer, "launch_by_launch_path": per_launch,
            "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "traffic": traffic, "traffic_source": traffic_src,
            # `traffic` above is the 64 x 128-tile kernel's; the two other families the `kernel` string names, from the same passes
            "traffic_by_family": {k: measured_traffic(TRAFFIC_TRAIN, sub)[0] for k, sub in
                                  (("conv1d_glds_kernel<2,4,2,2,2,false,0>", "conv1d_glds_kernel<2, 4, 2, 2, 2, false, 0"),
                                   ("conv1d_rt_gw_kernel<8,3,0,0>", "conv1d_rt_gw_kernel<8, 3, 0, 0"),
                                   ("conv1d_rt_gw_kernel<8,17,0,0>", "conv1d_rt_gw_kernel<8, 17, 0, 0"),
                                   ("conv1d_rt_gw_kernel<8,1,0,1> (gate backward)", "conv1d_rt_gw_kernel<8, 1, 0, 1"),
                                   ("diffnet_layer_kernel<5,true,0,8,true,true,2>", "diffnet_layer_kernel<5, true, 0, 8, true, true, 2"))}
            if dtype_name == "bf16" else None,
            # the byte model of the same families on THIS run's batch, mean per launch (bytes a launch must move: inputs once,
            # outputs once, weights once) -- traffic_by_family / this = the re-read factor of each kernel
            "algorithmic_bytes_by_family": alg_by_fam,
            "launches": n_launch, "avg_launch_us": round(1e3 * tot_ms / max(n_launch, 1), 2),
            "flop_per_step": tot_flop, "by_bound": by_bound,
            "note": "instrumented extra step on the timed batch with the longest utterances, issued launch by launch (the timed "
                    "steps issue the same kernels through the C-side stack drivers, except the DiffNet forward: there ONE launch "
                    "per layer, measured through the driver and substituted here -- launch_by_launch_path keeps the old mix); "
                    "`traffic` is read from the committed PMC "
                    "pass named in traffic_source, not measured in this run; "
                    "all launches are priced against the MFMA peak here for continuity with round 1; 40 of them are the DiffNet 1x1 "
                    "projections whose epilogues now also do the work of the elementwise kernels they replaced (residual / skip "
                    "update, gate backward), i.e. they got longer while the step got shorter; by_bound prices each launch "
                    "against the roof its arithmetic intensity selects"}


def cpu_baseline(model, batch):
    """The oracle (CPU restatement of the reference, fp32 PyTorch) on the host cores: forward + backward + clip + AdamW
    on a BOUNDED sample of one bench batch, by the protocol of SURVEY section 8d: a quarter of the batch, 2 warm-ups, median
    of 5 timed steps, at the thread count that a full sweep (every power of two up to the host's logical cores, on 2
    utterances) finds fastest -- all counts and times are in the `sample` string."""
    import statistics

    from oracle import ref_torch as R

    ncores = os.cpu_count() or 1
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    train_names = [k for k, p in model.named_parameters() if p.requires_grad]
    for k in train_names:
        sd[k].requires_grad_()
    opt = torch.optim.AdamW([sd[k] for k in train_names], lr=1e-3, betas=(0.9, 0.98), weight_decay=0.0)

    def sample(n):
        phon, dur, plen, mel, cf0, vuv, energy, flen, (ids, am) = [x if isinstance(x, tuple) else x[:n].cpu() for x in batch]
        ids, am = ids[:n].cpu(), am[:n].cpu()
        Tp, Tf = int(plen.max()), int(flen.max())
        cb = (phon[:, :Tp], dur[:, :, :Tp], plen, mel[:, :, :Tf], cf0[:, :, :Tf], vuv[:, :, :Tf], flen, ids, am)
        g = torch.Generator().manual_seed(0)
        return cb, torch.randint(0, 100, (n,), generator=g), torch.randn(n, 80, Tf, generator=g), int(flen.sum())

    def step(cb, t, noise):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = R.model_forward(sd, cb, t, noise, train_bn=True)["loss"]
        loss.backward()
        torch.nn.utils.clip_grad_norm_([sd[k] for k in train_names], 1.0)
        opt.step()
        return time.perf_counter() - t0

    forced = os.environ.get("PTPP_CPU_THREADS")
    # ascending thread counts; the sweep stops at the first count that is slower than the best so far (beyond a few dozen
    # threads this workload of many small ops only adds synchronisation: 0.35 / 0.59 / 1.30 s per step at 16 / 32 / 64
    # threads in round 2 -- and a 256-thread step did not finish within the bench's time budget)
    # sweep of the thread count ("N = all host cores" -- but this workload of many small ops gets SLOWER beyond
    # a few dozen threads), ascending powers of two up to the host's logical cores on 2 utterances
    cands = [int(forced)] if forced else sorted({c for c in (4, 8, 16, 32, 64, 128, 256, ncores) if c <= ncores})
    sweep, skipped = {}, {}
    cb2, t2, n2, _ = sample(min(2, batch[0].shape[0]))
    worse = 0
    for c in cands:
        if worse >= 2:
            # two counts in a row slower than the best: past the optimum the step time only grows with the thread count (round 6's
            # full sweep: 0.63 / 0.91 / 1.91 / 6.7 / 823 s per step at 16 / 32 / 64 / 128 / 256 threads -- the last one alone held
            # the bench for 14 minutes), so the remaining counts are listed, not run
            skipped[c] = "not run (past the optimum)"
            continue
        torch.set_num_threads(c)
        step(cb2, t2, n2)
        sweep[c] = step(cb2, t2, n2)
        log(f"cpu baseline sweep: {c} threads {sweep[c]:.2f}s")
        worse = worse + 1 if sweep[c] > 1.05 * min(sweep.values()) else 0
    nthr = min(sweep, key=sweep.get)
    torch.set_num_threads(nthr)
    # the section 8(d) protocol at the best count: a quarter of the batch, 2 warm-ups, median of 5 timed steps
    n = max(1, min(batch[0].shape[0], (batch[0].shape[0] + 3) // 4))
    cb, t, noise, frames = sample(n)
    for _ in range(2):
        step(cb, t, noise)
    times = [step(cb, t, noise) for _ in range(5)]
    med = statistics.median(times)
    log(f"cpu baseline: {nthr} threads, steps {times}")
    return {"value": round(frames / med, 1), "unit": "mel-frames/sec", "cores": nthr, "kind": "port",
            "sample": f"{n} utterances (a quarter of the bench batch of {batch[0].shape[0]}) / {frames} valid frames, fp32, dropout off, "
                      f"2 warm-ups + 5 timed steps, median {med:.2f} s/step (all {[round(x, 2) for x in times]}); thread count = best "
                      f"of a full sweep on 2 utterances: { {c: round(v, 2) for c, v in sweep.items()} } s/step"
                      + (f"; {skipped}" if skipped else "")
                      + f"; host has {ncores} logical cores"}


def _tame_gain(voc):
    with torch.no_grad():
        for name, p in voc.named_parameters():
            if name.endswith("weight_g"):
                p.mul_(0.4)  # keep the random-init generator off the tanh rails
    return voc


def vocoder_leg(dev, batch, frames, dtype):
    """BASELINE config 4: BigVGAN 24 kHz, batch x frames mel frames.  Returns (seconds per batch, generator)."""
    from promptttspp_amd.vocoders import BigVGAN

    torch.manual_seed(7)
    voc = _tame_gain(BigVGAN(80, 512, [6, 5, 4, 2], [12, 10, 8, 4], [3, 7, 11], [[1, 3, 5]] * 3)).to(dev).eval().set_compute_dtype(dtype)
    x = torch.clamp(-5.5 + 2.1 * torch.randn(batch, 80, frames, device=dev), -11.5, 2.0)
    for _ in range(2):
        voc(x)
    torch.cuda.synchronize()
    iters = 3
    t0 = time.perf_counter()
    for _ in range(iters):
        voc(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters, voc


def vocoder_valu_roofline(voc_model, x):
    """The THIRD roof of the fused AMP-layer kernels (amp_fused_kernel, C = 64 / 32: 59 % of the generator's time): an instrumented
    forward with a HIP-event pair around every ops.amp_layer call, grouped by instantiation (C, kernel size), against the floor
    each phase of the kernel has on this chip -- the phases do not overlap in practice, so they add:
      valu   2 anti-aliased Snakes per layer; per channel-sample and Snake 24 FMAs of the two 12-tap FIRs + 8 other VALU
             instructions (alpha x, the sin's prescale, sin^2, + x / alpha fma, per upsampled value) at the measured 2.3 cycles per
             wave64 instruction and 2 v_sin at 8.2 (profiles/r05_valu_rate.txt): 90 cycles per 64 channel-samples per Snake;
             1024 SIMDs at 2.4 GHz
      mfma   2 convs: 2 * 2 C^2 ks FLOP per sample at the dense bf16 peak
      io     x read + y written once (2 B per element) at the HBM peak
    frac = (valu + mfma + io floor) / measured; valu_frac = valu floor / measured."""
    from promptttspp_amd import ops

    recs = []
    orig = ops.amp_layer

    def timed(xb, w1p, b1, w2p, b2, la1, la2, t1, t2, ks, dil, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(xb, w1p, b1, w2p, b2, la1, la2, t1, t2, ks, dil, **kw)
        e1.record()
        recs.append((e0, e1, xb.shape[0] * xb.shape[1], xb.shape[2], int(ks)))
        return r

    ops.amp_layer = timed
    try:
        voc_model(x)
        torch.cuda.synchronize()
    finally:
        ops.amp_layer = orig
    SIMDS, CLK, CYC_PER_64 = 1024, 2.4e9, 24 * 2.3 + 8 * 2.3 + 2 * 8.2
    groups = {}
    for e0, e1, rows, C, ks in recs:
        g = groups.setdefault((C, ks), {"launches": 0, "ms": 0.0, "rows": rows})
        g["launches"] += 1
        g["ms"] += e0.elapsed_time(e1)
    out, tot_ms, tot_floor, tot_valu = {}, 0.0, 0.0, 0.0
    for (C, ks), g in sorted(groups.items()):
        cs = g["rows"] * C
        valu = 1e3 * 2 * cs / 64 * CYC_PER_64 / (SIMDS * CLK)
        mfma = 1e3 * 2 * 2.0 * C * C * ks * g["rows"] / (MFMA_BF16_PEAK_TFLOPS * 1e12)
        io = 1e3 * 2 * cs * 2 / (HBM_PEAK_GBS * 1e9)
        ms = g["ms"] / g["launches"]
        out[f"C{C}_k{ks}"] = {"launches": g["launches"], "ms_per_launch": round(ms, 3), "floor_valu_ms": round(valu, 3),
                              "floor_mfma_ms": round(mfma, 3), "floor_io_ms": round(io, 3), "frac": round((valu + mfma + io) / ms, 3),
                              "valu_frac": round(valu / ms, 3)}
        tot_ms += g["ms"]
        tot_floor += (valu + mfma + io) * g["launches"]
        tot_valu += valu * g["launches"]
    if not out:
        return None
    return {"kernel": "amp_fused_kernel (one launch per AMP layer, C = 64 / 32)", "by_instantiation": out, "ms_per_forward": round(tot_ms, 3),
            "floor_ms_per_forward": round(tot_floor, 3), "frac": round(tot_floor / tot_ms, 3), "valu_frac": round(tot_valu / tot_ms, 3),
            "model": "phases add (no VALU / MFMA / HBM overlap inside a workgroup or between co-resident ones: DESIGN.md 5f.1); VALU: "
                     "90 cycles per 64 channel-samples per Snake (24 FMA + 8 other at 2.3 cycles, 2 v_sin at 8.2), 1024 SIMDs at 2.4 GHz"}


def _cpu_threads():
    ncores = os.cpu_count() or 1
    nthr = int(os.environ.get("PTPP_CPU_THREADS", min(ncores, 32)))
    torch.set_num_threads(nthr)
    return nthr, ncores


def vocoder_cpu_baseline(voc, frames):
    """The oracle's BigVGAN (CPU restatement of vocoders/bigvgan.py, fp32) on the host cores on a BOUNDED sample:
    2 utterances of `frames` frames (SURVEY section 8d asks for 8 x 10 s; 2 x 10 s keeps the default bench run short --
    the vocoder is batch-parallel, so the real-time factor carries over)."""
    from oracle import ref_torch as R

    nthr, ncores = _cpu_threads()
    sd = {k: v.detach().float().cpu() for k, v in voc.state_dict().items()}
    x = torch.clamp(-5.5 + 2.1 * torch.randn(2, 80, frames), -11.5, 2.0)
    with torch.no_grad():
        R.bigvgan(sd, x[:1, :, :100])  # warm-up
        t0 = time.perf_counter()
        R.bigvgan(sd, x)
        dt = time.perf_counter() - t0
    audio = 2 * frames * 0.01
    return {"value": round(dt / audio, 5), "unit": "RTF (s compute per s of 24 kHz audio; lower is better)", "cores": nthr,
            "kind": "port", "sample": f"2 x {frames} frames ({audio:.0f} s of audio) in {dt:.1f} s, fp32; host has {ncores} logical cores"}


def _tame_durations(model):
    """A random-init duration head predicts exp(mu + sigma^2/2) of anything: give it the statistics of a
    trained one (about 5 frames per phone) so that the synthetic prompts produce LibriTTS-R-shaped mels."""
    with torch.no_grad():
        ol = model.variance_adaptor.duration_predictor.out_layer
        ol.mu.weight.mul_(0.05)
        ol.mu.bias.fill_(1.6)
        ol.log_sigma.weight.mul_(0.05)
        ol.log_sigma.bias.fill_(-1.5)
    return model


def app_inputs(n, seed=0):
    import numpy as np

    r = np.random.default_rng(seed)
    tp = r.integers(40, 121, size=n)
    ph = torch.zeros(n, int(tp.max()), dtype=torch.long)
    for i, L in enumerate(tp):
        ph[i, :L] = torch.from_numpy(np.concatenate([[1], r.integers(3, 90, size=int(L) - 2), [2]]))
    Lp = r.integers(12, 49, size=n)
    ids = torch.zeros(n, int(Lp.max()), dtype=torch.long)
    am = torch.zeros_like(ids)
    for i, L in enumerate(Lp):
        ids[i, 0], ids[i, L - 1] = 101, 102
        ids[i, 1 : L - 1] = torch.from_numpy(r.integers(1000, 30000, size=int(L) - 2))
        am[i, :L] = 1
    return ph, torch.from_numpy(tp), ids, am


def app_leg(dev, dtype, n_prompts=32):
    """BASELINE config 5: prompt -> waveform for 32 prompts, phone sequences Tp ~ U{40..120}: infer_batch (use_max,
    noise_scale 0.5, 100-step sampler) -> zero-phase low-pass of log-F0 -> F0-aware BigVGAN.  bf16 decoder / vocoder,
    f32 duration / MDN heads.  Returns a dict incl. the sampler's share, and the models for the CPU baseline."""
    from promptttspp.utils.model import lowpass_filter
    from promptttspp_amd import hydra_lite as H
    from promptttspp_amd.modules.prompt_encoder import allow_random_bert

    conf = os.path.join(ROOT, "egs", "proposed", "bin", "conf")
    torch.manual_seed(11)
    with allow_random_bert():
        model = _tame_durations(H.instantiate(H.load_node(os.path.join(conf, "model", "prompttts_mdn_v2_wo_erg_final.yaml")))).to(dev).eval()
    voc = _tame_gain(H.instantiate(H.load_node(os.path.join(conf, "vocoder", "bigvgan_f0.yaml")))).to(dev).eval()
    voc.set_compute_dtype(dtype)
    ph, pl, ids, am = app_inputs(n_prompts)
    ph, pl, prm = ph.to(dev), pl.to(dev), (ids.to(dev), am.to(dev))
    tsamp = [0.0]
    dec = model.decoder
    orig = dec.inference_cl

    def timed_sampler(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = orig(*a, **k)
        torch.cuda.synchronize()
        tsamp[0] += time.perf_counter() - t0
        return out

    from promptttspp_amd import config as _cfg

    def run():
        with torch.no_grad(), _cfg.use_dtype(dtype):
            mel, cf0, vuv, flen = model.infer_batch(ph, pl, style_prompt=prm, use_max=True, noise_scale=0.5, return_f0=True)
            f0 = lowpass_filter(cf0, 100, cutoff=20).exp()
            f0[vuv < 0.5] = 0
            wav = voc(mel, f0)
        return mel, flen, wav

    for _ in range(2):
        run()
    dec.inference_cl = timed_sampler
    torch.cuda.synchronize()
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        mel, flen, wav = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    dec.inference_cl = orig
    frames = int(flen.sum())
    res = {"prompts": n_prompts, "ms_per_batch": round(1e3 * dt, 2), "valid_frames": frames, "audio_s": round(frames * 0.01, 1),
           "rtf": dt / (frames * 0.01), "sampler_ms": round(1e3 * tsamp[0] / n, 2), "padded_mel": list(mel.shape),
           "finite": bool(torch.isfinite(wav).all())}
    return res, model, voc


def app_cpu_baseline(model, voc, n_prompts=2):
    """The oracle's prompt -> waveform path on the host cores for a BOUNDED sample (2 prompts; SURVEY section 8d asks for
    4): encoder + prompt branch + variance adaptor + the 100-step sampler + low-pass + NSF source + F0-aware BigVGAN."""
    import numpy as np
    from scipy import signal

    from oracle import ref_torch as R

    nthr, ncores = _cpu_threads()
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    sdv = {k: v.detach().float().cpu() for k, v in voc.state_dict().items()}
    ph, pl, ids, am = app_inputs(n_prompts, seed=1)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        t0 = time.perf_counter()
        mel, cf0, vuv, flen, _ = R.model_infer_batch(
            sd, ph, pl, lambda b, t: torch.randn(b, 80, t, generator=g), lambda b, t: [torch.randn(b, 80, t, generator=g) for _ in range(100)],
            ids=ids, am=am, style_noise=torch.randn(n_prompts, 1, 256, generator=g), noise_scale=0.5)
        b, a = signal.butter(5, [20 / 50], "lowpass")
        f0 = torch.from_numpy(np.ascontiguousarray(R.filtfilt_zero_state(cf0.numpy(), b, a))).float().exp()
        f0[vuv < 0.5] = 0
        L = f0.shape[-1] * 240
        src = R.nsf_source(sdv, f0, torch.rand(n_prompts, 9, generator=g), torch.randn(n_prompts, L, 9, generator=g))
        R.bigvgan(sdv, mel, source=src)
        dt = time.perf_counter() - t0
    frames = int(flen.sum())
    return {"value": round(dt / (frames * 0.01), 4), "unit": "RTF (s compute per s of audio; lower is better)", "cores": nthr,
            "kind": "port", "sample": f"{n_prompts} prompts, {frames} frames ({frames * 0.01:.1f} s of audio) in {dt:.1f} s, fp32; "
                                      f"host has {ncores} logical cores"}


# ------------------------------------------------------------------------------------------
def log(msg):
    if os.environ.get("PTPP_BENCH_VERBOSE"):
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def self_launch(a):
    """``python bench.py --gpus N`` without a launcher (no WORLD_SIZE in the environment): start the N ranks here, one
    process per GPU, the way the reference's trainer spawns its workers from a plain ``python train.py``
    (/root/reference/promptttspp/trainers/tts.py:40-48).  Rank 0 inherits stdout (the ONE JSON line); the other ranks'
    stdout goes to stderr.  Returns the worst exit status; a rank that dies takes the others down with it."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for pr in list(pending):
                st = pr.poll()
                if st is None:
                    continue
                pending.remove(pr)
                if st != 0:
                    rc = rc or st
                    for other in pending:  # a dead rank would leave the others hanging in a collective
                        other.terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    # stdout carries exactly ONE line (the JSON): native libraries write banners there (RCCL prints its
    # version block on the first collective), so fd 1 points at stderr for the whole run and the result
    # goes to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    if os.environ.get("PTPP_BENCH_LAUNCH_PROBE"):
        # launcher check for a box without GPUs (tests/test_bench_launch.py): rendezvous + one collective over gloo,
        # rank 0 answers on the real stdout exactly as the benchmark does; nothing is measured
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29556")
        if os.environ["PTPP_BENCH_LAUNCH_PROBE"] == f"fail{rank}":
            sys.exit(3)  # the launcher must notice, stop the ranks left waiting in the rendezvous and report failure
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        print(f"probe rank {rank} of {world}")  # lands on stderr: fd 1 is redirected
        if rank == 0:
            os.write(real_stdout, (json.dumps({"probe": True, "n_gpus": world, "sum": float(t[0])}) + "\n").encode())
        dist.barrier()
        dist.destroy_process_group()
        return
    # PTPP_BENCH_SELFTEST=1: all ranks on device 0 over gloo -- exercises the N > 1 code path on a 1-GPU
    # box (RCCL refuses two ranks per device); never set for measurements
    selftest = bool(os.environ.get("PTPP_BENCH_SELFTEST"))
    if selftest:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from promptttspp_amd import functional as PF0

    PF0.create_side_stream(dev)  # before RCCL creates its streams (see the docstring)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if selftest:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm

    elif os.environ.get("PTPP_DP_FORCE_COLLECTIVES"):  # diagnostics: one rank over RCCL (see parallel.py)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    from promptttspp_amd import _lib, config
    from promptttspp_amd import functional as PF

    _lib.load()  # fail loudly if the HIP extension is missing
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    config.set_compute_dtype(dtype)
    PF.manual_seed(1000 + rank)  # different dropout streams per rank

    # MI355X has 288 GB: take a slab for the caching allocator up front so that the token-bucket batches
    # (a new (batch, length) shape almost every step) carve their tensors out of it instead of growing the
    # pool with hipMalloc calls inside the timed steps (PTPP_RESERVE_GIB=0 disables)
    gib = float(os.environ.get("PTPP_RESERVE_GIB", "24"))
    if gib > 0:
        slab = torch.empty(int(gib * (1 << 30)), device=dev, dtype=torch.uint8)
        del slab

    log("building model")
    model = build_model(dev).train()
    log("building batches")
    batches = make_batches(rank, world, a.steps + a.warmup + 1, a.max_tokens, dev)
    log(f"{len(batches)} batches, first: B={batches[0][0].shape[0]} Tp={batches[0][0].shape[1]} Tf={batches[0][3].shape[2]}")
    assert len(batches) >= a.steps + a.warmup + 1
    frame_counts = [int(b[7].sum()) for b in batches]  # host-side bookkeeping, outside the timed region
    red, opt, sched = train_setup(model, world)

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    if a.preheat > 0 and a.warmup > 0 and not os.environ.get("PTPP_BENCH_SELFTEST"):  # setup, not measurement: the same steps as the warm-up, until the clocks have settled
        tp = time.perf_counter()
        for k in range(a.preheat):
            train_step(model, batches[k % a.warmup], red, opt, sched)
        torch.cuda.synchronize()
        log(f"preheat: {a.preheat} untimed steps in {time.perf_counter() - tp:.2f} s")
    for i in range(a.warmup):
        train_step(model, batches[i], red, opt, sched)
        torch.cuda.synchronize()
        log(f"warmup step {i} done")
    if world > 1 or os.environ.get("PTPP_DP_FORCE_COLLECTIVES"):
        red.enable_timing()  # three event records per step on the main stream: what the exchange costs it (the "dp" object below)
    barrier()
    nalloc0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    t0 = time.perf_counter()
    frames = 0
    for i in range(a.steps):
        b = batches[a.warmup + i]
        out = train_step(model, b, red, opt, sched)
        frames += frame_counts[a.warmup + i]
    host_dt = time.perf_counter() - t0  # the host has ENQUEUED the K steps (no device sync inside a step): << dt = the GPU is the limit
    barrier()
    dt = time.perf_counter() - t0
    loss = float(out["loss"])
    log(f"timed region done: {dt:.3f}s for {a.steps} steps (host enqueue {host_dt:.3f}s), loss {loss:.4f}; device allocations inside it: "
        f"{torch.cuda.memory_stats(dev).get('num_device_alloc', 0) - nalloc0}, reserved {torch.cuda.memory_reserved(dev) / 2**30:.1f} GiB")

    dp = None
    if getattr(red, "_timing", None) is not None:
        dp = red.timing_summary()
    if world > 1:
        import torch.distributed as dist

        tt = torch.tensor([dt, float(frames)], device=dev, dtype=torch.float64)
        tmax, tmin = tt.clone(), tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        ex = torch.tensor([dp["exposed_allreduce_ms"] or 0.0, dp["join_gradient_streams_ms"] or 0.0], device=dev, dtype=torch.float64)
        exmax = ex.clone()
        dist.all_reduce(exmax, op=dist.ReduceOp.MAX)
        # per-rank wall time of the timed region (min / max over ranks, ms per step) and the exchange as the slowest rank saw it
        dp.update({"rank_ms_per_step_min": round(1e3 * float(tmin[0]) / a.steps, 3),
                   "rank_ms_per_step_max": round(1e3 * float(tmax[0]) / a.steps, 3),
                   "frames_per_rank_min": float(tmin[1]), "frames_per_rank_max": float(tmax[1]),
                   "exposed_allreduce_ms_max_over_ranks": round(float(exmax[0]), 4),
                   "join_gradient_streams_ms_max_over_ranks": round(float(exmax[1]), 4)})
        dt, frames = float(tmax[0]), float(tt[1])

    voc = None
    voc_model = None
    if not a.no_vocoder:
        vdt, voc_model = vocoder_leg(dev, a.voc_batch, a.voc_frames, dtype)
        if world > 1:
            import torch.distributed as dist

            v = torch.tensor([vdt], device=dev, dtype=torch.float64)
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
            vdt = float(v[0])
        audio_s = a.voc_batch * a.voc_frames * 0.01 * world
        # algorithmic bytes of SURVEY section 8d: 789 312 activation elements per mel frame (every conv reads its input and
        # writes its output once, Snake / residual / block mean fused away)
        alg_bytes = a.voc_batch * a.voc_frames * 789312 * (2 if a.dtype == "bf16" else 4)
        ach = alg_bytes / vdt / 1e9
        traffic, tsrc = measured_traffic(TRAFFIC_VOC) if (a.dtype == "bf16" and a.voc_batch == 64 and a.voc_frames == 1000) \
            else (None, "PMC summary is for 64 x 1000 frames bf16 only")
        # BASELINE config 4 is worded "fp16": the same generator with IEEE-half storage (PTPP_F16), timed beside the bf16 run
        f16 = valu_roof = None
        if a.dtype == "bf16" and world == 1:
            voc_model.set_compute_dtype(torch.float16)
            xx = torch.clamp(-5.5 + 2.1 * torch.randn(a.voc_batch, 80, a.voc_frames, device=dev), -11.5, 2.0)
            for _ in range(2):
                voc_model(xx)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                voc_model(xx)
            torch.cuda.synchronize()
            fdt = (time.perf_counter() - t0) / 3
            voc_model.set_compute_dtype(torch.bfloat16)
            f16 = {"ms_per_batch": round(1e3 * fdt, 3), "rtf": fdt / audio_s,
                   "roofline_frac": round(alg_bytes / fdt / 1e9 / HBM_PEAK_GBS, 4), "dtype": "f16"}
            valu_roof = vocoder_valu_roofline(voc_model, xx)
            del xx
        voc = {"rtf": vdt / audio_s, "ms_per_batch": 1e3 * vdt, "batch": a.voc_batch, "frames": a.voc_frames, "dtype": a.dtype,
               "f16": f16, "roofline_valu": valu_roof,
               "algorithmic_tflops": world * a.voc_batch * a.voc_frames * 444.5e6 / vdt / 1e12,
               "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes": alg_bytes, "traffic": traffic,
                            "traffic_source": tsrc, "per": "one forward of the whole generator on one GPU"}}

    log("vocoder leg done")
    # the instrumented step contains the gradient all-reduce: EVERY rank runs it, rank 0 reports
    # instrumented on the timed batch with the longest utterances (the 128 x 128-tile instantiation takes
    # every frame-level layer there; short-utterance batches route some launches to the 64 x 128 tiles)
    rb = max(range(a.warmup, a.warmup + a.steps), key=lambda i: batches[i][3].shape[2])
    roof = conv_roofline(model, batches[rb], red, opt, sched, a.dtype)
    if rank == 0:
        log(f"roofline pass done: {roof['achieved']} TFLOP/s")
        one = not (a.no_cpu_baseline or world > 1)  # the CPU baselines run on rank 0 at N = 1 only
        # every GPU leg first, the CPU baselines after them: the app leg measured right after the 32-thread CPU legs
        # came out 15-20 % slower (211 vs 179 ms) than on its own
        app = None
        if not a.no_app and world == 1:
            app, app_model, app_voc = app_leg(dev, dtype)
            log(f"app leg done: {app['ms_per_batch']} ms")
            if a.dtype == "bf16":
                # BASELINE config 5 is worded "fp16 mel decoder + fp32 MDN head": the same path with IEEE-half storage for the
                # sampler (one-launch DiffNet layers, sampler head, conditioner GEMM) and the vocoder, f32 conditioning path
                a16, m16, v16 = app_leg(dev, torch.float16)
                app["f16"] = {k: a16[k] for k in ("ms_per_batch", "sampler_ms", "rtf", "valid_frames", "finite")}
                del m16, v16
                log(f"app leg (f16) done: {a16['ms_per_batch']} ms")
        cpu = cpu_baseline(model, batches[a.warmup]) if one else None
        log("cpu baseline done")
        if voc is not None and one:
            voc["cpu_baseline"] = vocoder_cpu_baseline(voc_model, a.voc_frames)
            log("vocoder cpu baseline done")
        if app is not None and one:
            app["cpu_baseline"] = app_cpu_baseline(app_model, app_voc)
            log("app cpu baseline done")
        B = batches[a.warmup][0].shape[0]
        line = {
            "metric": "mel-frames/sec (train)", "value": round(frames / dt, 1), "unit": "mel-frames/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "train.py model=prompttts_mdn_v2_wo_erg_final, dataset.max_tokens=%d per GPU, synthetic "
                                   "LibriTTS-R-shaped utterances, fwd+bwd+clip+AdamW+Noam, train mode (dropout on)" % a.max_tokens,
                       "utts_per_gpu_batch": int(B), "parallelism": f"dp{world}", "final_loss": round(loss, 4),
                       "preheat_steps": a.preheat, "host_enqueue_ms_per_step": round(1e3 * host_dt / a.steps, 3),
                       "env": {"HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG")}},
            "roofline": roof, "cpu_baseline": cpu,
            "per_gpu_value": round(frames / dt / world, 1),
            "dp": dp,
            "bigvgan": voc,
            "app_path": app,
            "box": box_calibration(dev),
        }
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
