"""CPU, world_size 2, gloo: the data-parallel pieces that do not need a GPU --
the flat-bucket gradient reducer (mean all-reduce == single-process gradient of the
concatenated batch for a mean-normalised loss), parameter broadcast, and the reference's
rank sharding x[rank::W] of token-bucket batches (trainers/tts.py:122-143)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from promptttspp_amd.parallel import FlatGradReducer

        torch.manual_seed(100 + rank)  # different init per rank: broadcast must fix it
        model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 4))
        red = FlatGradReducer(model.parameters(), bucket_elems=100)  # several buckets
        assert len(red.buckets) >= 2
        red.broadcast_parameters(model)
        g = torch.Generator().manual_seed(7)
        x = torch.randn(8, 16, generator=g)
        y = torch.randn(8, 4, generator=g)
        xs, ys = x[rank::world], y[rank::world]
        for step in range(2):
            red.zero_grad()
            loss = ((model(xs) - ys) ** 2).mean()
            loss.backward()
            red.finish()
        if rank == 0:
            # plain numpy: torch tensors would travel as shared-memory handles that die with this process
            # (the flat buffer pads every bucket to a multiple of 64 floats: collect the gradient views, in its order)
            flat = torch.cat([p.grad.reshape(-1) for p in reversed(list(model.parameters()))])
            assert all(p.grad.data_ptr() >= red.flat.data_ptr() for p in model.parameters())
            assert all((b - a) % 64 == 0 for a, b, _ in red.buckets)
            out.put({"flat": flat.numpy().copy(), "params": [p.detach().numpy().copy() for p in model.parameters()],
                     "x": x.numpy().copy(), "y": y.numpy().copy()})
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_flat_grad_reducer_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    res = {k: ([torch.from_numpy(a) for a in v] if isinstance(v, list) else torch.from_numpy(v)) for k, v in res.items()}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference with rank 0's (broadcast) parameters on the full batch
    model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 4))
    with torch.no_grad():
        for p, v in zip(model.parameters(), res["params"]):
            p.copy_(v)
    loss = ((model(res["x"]) - res["y"]) ** 2).mean()
    loss.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in reversed(list(model.parameters()))])  # reducer layout: reverse order
    assert torch.allclose(res["flat"], ref, atol=1e-6)


def test_rank_sharding_of_token_bucket_batches():
    from promptttspp_amd.datasets.synthetic import SyntheticLibriTTSR
    from promptttspp_amd.datasets.utils import batch_by_size

    ds = SyntheticLibriTTSR(num_utts=1500, seed=1234)
    for W in (1, 2, 8):
        gb = batch_by_size(ds.ordered_indices(), ds.num_tokens, max_tokens=30000 * W, required_batch_size_multiple=W)
        gb = [b for b in gb if len(b) % W == 0]
        for b in gb[:: max(1, len(gb) // 10)]:
            shards = [b[r::W] for r in range(W)]
            assert sorted(i for s in shards for i in s) == sorted(b)          # a partition
            assert len({len(s) for s in shards}) == 1                            # equal utterance counts
            tok = [len(s) * max(ds.num_tokens(i) for i in s) for s in shards]
            assert max(tok) <= 30000 * 1.02                                      # ~max_tokens padded frames per GPU
            fr = [sum(ds.num_tokens(i) for i in s) for s in shards]
            assert (max(fr) - min(fr)) / max(fr) < 0.1                           # interleaving balances the work


def test_default_bucket_layout_tapers_towards_the_end_of_the_backward(monkeypatch):
    """The flat buffer is laid out in backward order, so the last buckets are the ones nothing hides: by default every bucket is
    half the size of the one before, down to PTPP_DP_BUCKET_MIN_MB.  Every parameter keeps one bucket, the
    buckets tile the buffer, each ends on a multiple of 64 floats; PTPP_DP_BUCKET_MIN_MB=0 and an explicit size stay constant."""
    from promptttspp_amd.parallel import FlatGradReducer

    ps = [torch.nn.Parameter(torch.zeros(50_000 + 13 * i)) for i in range(60)]  # 3.0 M floats, ragged sizes
    monkeypatch.setenv("PTPP_DP_BUCKET_MB", "4")
    monkeypatch.setenv("PTPP_DP_BUCKET_MIN_MB", "1")
    red = FlatGradReducer(ps, direct=False)
    mb = [(b - a) * 4 / 2 ** 20 for a, b, _ in red.buckets]
    assert len(mb) >= 4 and 4.0 <= mb[0] < 4.3 and 2.0 <= mb[1] < 2.3 and all(1.0 <= m < 1.3 for m in mb[2:-1]) and mb[-1] < 1.3, mb
    assert red.buckets[0][0] == 0 and all(red.buckets[i][1] == red.buckets[i + 1][0] for i in range(len(mb) - 1))
    assert red.buckets[-1][1] == red.flat.numel() and all((b - a) % 64 == 0 for a, b, _ in red.buckets)
    assert sum(n for _, _, n in red.buckets) == len(ps)
    for p in ps:  # gradients are views of the flat buffer, inside their bucket
        a, b, _ = red.buckets[red._bucket_of[id(p)]]
        o = (p.grad.data_ptr() - red.flat.data_ptr()) // 4
        assert a <= o and o + p.numel() <= b
    monkeypatch.setenv("PTPP_DP_BUCKET_MIN_MB", "0")
    const = [(b - a) * 4 / 2 ** 20 for a, b, _ in FlatGradReducer(ps, direct=False).buckets]
    assert all(4.0 <= m < 4.3 for m in const[:-1]), const
    monkeypatch.setenv("PTPP_DP_BUCKET_MIN_MB", "1")
    fixed = [(b - a) for a, b, _ in FlatGradReducer(ps, bucket_elems=500_000, direct=False).buckets]
    assert all(500_000 <= n < 560_000 for n in fixed[:-1]), fixed
