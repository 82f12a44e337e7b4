"""GPU, world_size 2 on ONE device over gloo: the data-parallel TRAINING path of the acoustic model
end to end -- weight-gradient kernels accumulating straight into the flat buffer, per-parameter use
counting, bucket notifications, asynchronous bucket all-reduces, finish() -- against plain single-process
autograd gradients of the two rank shards.  (RCCL refuses two ranks on one device, so the collective here
is gloo; bench.py / the trainer use backend "nccl" = RCCL with the same reducer code.)"""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port):
    import torch.distributed as dist

    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.dirname(here)):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import test_hip_acoustic as T
        from promptttspp_amd import config
        from promptttspp_amd import functional as PF
        from promptttspp_amd.parallel import FlatGradReducer

        dev = torch.device("cuda:0")
        torch.cuda.set_device(0)
        config.set_compute_dtype(torch.float32)
        model, g = T._model(dev)
        model.train()
        full = T._batch(g, dev)

        def shard(r):
            out = [x[r::world] if not isinstance(x, tuple) else tuple(t[r::world] for t in x) for x in full]
            return out, {"t": g["t"][r::world], "noise": g["noise"][r::world]}

        def run(r):
            b, inj = shard(r)
            model.decoder.injected = inj
            PF.manual_seed(1000 + r)
            torch.manual_seed(50 + r)
            model(b)["loss"].backward()

        params = [p for p in model.parameters() if p.requires_grad]
        if os.environ.get("PTPP_DP_DEBUG"):
            from collections import Counter

            CNT = Counter()
            _orig_hook = FlatGradReducer._hook

            def _counting(self, p):
                CNT[id(p)] += 1
                return _orig_hook(self, p)
            FlatGradReducer._hook = _counting
        red = FlatGradReducer(params, bucket_elems=8 * 1024 * 1024)  # several buckets
        assert len(red.buckets) >= 3 and PF.direct_grads_enabled()
        red.broadcast_parameters(model)
        red.zero_grad()
        if os.environ.get("PTPP_DP_DEBUG"):
            from collections import Counter

            cnt, order = Counter(), []
            names_by_id = {id(p): n for n, p in model.named_parameters()}
            orig = red._hook

            def spy(p):
                cnt[id(p)] += 1
                order.append(names_by_id[id(p)])
                orig(p)
            PF._direct["notify"] = spy
            red._hook = spy
        run(rank)
        if os.environ.get("PTPP_DP_DEBUG"):
            multi = [(names_by_id[i], c) for i, c in CNT.items() if c > 1]
            print(f"rank {rank} CNT total {sum(CNT.values())} distinct {len(CNT)} params {len(params)}", flush=True)
            never = [n for n, p in model.named_parameters() if p.requires_grad and cnt[id(p)] == 0]
            print(f"rank {rank} notified {len(cnt)} params; multiple: {multi[:10]}; never: {never[:10]} ({len(never)})", flush=True)
            print(f"rank {rank} first notified: {order[:6]} pending {red._pending}", flush=True)
        red.finish()
        torch.cuda.synchronize()
        dp = [p.grad.detach().clone() for p in params]

        PF.enable_direct_grads(False)
        ref = [torch.zeros_like(p) for p in params]
        per = []
        for r in range(world):
            for p in params:
                p.grad = None
            run(r)
            per.append([p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p) for p in params])
            for a, p in zip(ref, params):
                if p.grad is not None:
                    a += p.grad / world
        if os.environ.get("PTPP_DP_DEBUG"):
            names = [n for n, p in model.named_parameters() if p.requires_grad]
            for i in (0, 5, 7, 8, 100, 200, 300, len(params) - 1):
                print(f"rank {rank} {names[i][:50]:50s} |dp| {float(dp[i].norm()):.4e} |r0| {float(per[0][i].norm()):.4e} "
                      f"|r1| {float(per[1][i].norm()):.4e} |mean| {float(ref[i].norm()):.4e} launched={red._launched} ", flush=True)
        bad = []
        for i, (a, b_) in enumerate(zip(dp, ref)):
            scale = float(b_.abs().max()) + 1e-12
            # (f32 atomics in the BatchNorm statistics / reductions make runs differ at the 1e-4 level)
            if float((a - b_).abs().max()) > 1e-3 * scale + 1e-6:
                bad.append((i, tuple(a.shape), scale, float((a - b_).abs().max())))
        assert not bad, bad[:5]

        # one optimiser step on the exchanged gradients: every rank must hold bit-identical parameters afterwards
        # (DDP's invariant, trainers/tts.py:117,206-211), and the BatchNorm statistics after the opt-in buffer
        # broadcast must be rank 0's
        from promptttspp_amd.optim import FusedAdamW

        for p, gdp in zip(params, dp):
            p.grad = gdp
        FusedAdamW(params, lr=1e-3, max_grad_norm=1.0).step()
        red.broadcast_buffers(model)
        torch.cuda.synchronize()
        sig = torch.stack([p.detach().double().sum() for p in params] +
                          [b.detach().double().sum() for b in model.buffers() if b.is_floating_point()]).cpu()
        probe = torch.cat([p.detach().flatten()[:64].float() for p in params]).cpu()
        sigs, probes = [torch.zeros_like(sig) for _ in range(world)], [torch.zeros_like(probe) for _ in range(world)]
        dist.all_gather(sigs, sig)
        dist.all_gather(probes, probe)
        names_all = [n for n, p in model.named_parameters() if p.requires_grad] + \
            ["buf:" + n for n, b in model.named_buffers() if b.is_floating_point()]
        # bit for bit, with and without the weight-gradient side stream.  (Round 2 bounded the side-stream leg at 1e-6 after one
        # run that differed in the last bits; what differed then was the clip factor -- the squared-gradient sum used f32
        # atomics until ptpp_grad_sumsq_det -- not the exchange: test_no_gradient_kernel_writes_a_bucket_after_its_collective_
        # was_issued shows that no gradient kernel of either stream runs after its bucket's collective.)
        for r in range(1, world):
            diff = sigs[0] != sigs[r]
            bad = [(names_all[i], float(sigs[0][i]), float(sigs[r][i])) for i in diff.nonzero().flatten().tolist()]
            assert not bad, f"rank {r} diverged from rank 0: {len(bad)} tensors, e.g. {bad[:6]}"
            assert torch.equal(probes[0], probes[r])
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _run_two_ranks():
    ctx = torch.multiprocessing.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


def test_dp_training_step_two_ranks_one_device():
    _run_two_ranks()


def test_dp_training_step_two_ranks_with_the_weight_gradient_side_stream():
    """The configuration multi-rank RCCL runs use: weight gradients on the side stream, every bucket
    all-reduce joining it first.  Forced here over gloo (slow with gloo, but the gradients must be the same)."""
    os.environ["PTPP_FORCE_ASYNC_WGRAD"] = "1"
    try:
        _run_two_ranks()
    finally:
        del os.environ["PTPP_FORCE_ASYNC_WGRAD"]


def test_no_gradient_kernel_writes_a_bucket_after_its_collective_was_issued():
    """DDP's invariant (every rank applies the SAME reduced gradients) needs every kernel that writes a bucket to be
    ordered before that bucket's collective.  One rank, the whole multi-rank machinery on (hooks, per-bucket launches from
    the weight-gradient side stream, finish()), the collective replaced by a snapshot taken on the stream and at the point
    the collective would run: after finish() every bucket must still equal its snapshot -- a difference is a gradient
    kernel (of either stream) that ran after "its" all-reduce, i.e. a contribution the other ranks would never see."""
    import torch.distributed as dist

    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.dirname(here)):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ["PTPP_DP_FORCE_COLLECTIVES"] = "1"
    os.environ["PTPP_FORCE_ASYNC_WGRAD"] = "1"
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        import test_hip_acoustic as T
        from promptttspp_amd import config
        from promptttspp_amd import functional as PF
        from promptttspp_amd.parallel import FlatGradReducer

        from promptttspp_amd.models.prompttts_mdn_v2_final import model as M

        assert M.BRANCH_STREAMS == "2"  # the default: prompt branch and reference encoder backward run on their own streams
        dev = torch.device("cuda:0")
        for dtype in (torch.float32, torch.bfloat16):
            config.set_compute_dtype(dtype)
            model, g = T._model(dev)
            model.train()
            batch = T._batch(g, dev)
            params = [p for p in model.parameters() if p.requires_grad]
            red = FlatGradReducer(params, bucket_elems=2 * 1024 * 1024)
            assert red.collective and len(red.buckets) >= 8 and PF._direct["async"]
            snaps = []

            def fake_reduce(t, snaps=snaps):
                snaps.append((t, t.clone()))  # on the stream, and at the point in it, where the all-reduce would be enqueued

            red._reduce = fake_reduce
            names = {id(p): n for n, p in model.named_parameters()}
            for rep in range(3):
                snaps.clear()
                red.zero_grad()
                model.decoder.injected = {"t": g["t"], "noise": g["noise"]}
                PF.manual_seed(5 + rep)
                model(batch)["loss"].backward()
                red.finish()
                torch.cuda.synchronize()
                assert len(snaps) == len(red.buckets)
                late = []
                for view, snap in snaps:
                    if not torch.equal(view, snap):
                        off = int((view != snap).nonzero()[0]) + view.storage_offset()
                        owner = [names[id(p)] for p in params if p.grad.storage_offset() <= off < p.grad.storage_offset() + p.numel()]
                        late.append((owner, int((view != snap).sum())))
                assert not late, f"{dtype}: gradient writes after the bucket's collective: {late[:8]}"
            PF.enable_direct_grads(False)
    finally:
        from promptttspp_amd import config, functional as PF

        PF.enable_direct_grads(False)
        config.set_compute_dtype(torch.float32)
        del os.environ["PTPP_DP_FORCE_COLLECTIVES"], os.environ["PTPP_FORCE_ASYNC_WGRAD"]
        dist.destroy_process_group()
