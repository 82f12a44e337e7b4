"""GPU: the C ABI's RCCL entry points (include/ptpp.h "Data-parallel gradient exchange"; reference call sites
promptttspp/trainers/tts.py:52-55,117).  A 1-GPU box can only form a communicator of one rank (RCCL refuses two
ranks on one device): the calls, the dlopen binding and the stream ordering are exercised; the multi-rank
arithmetic of the same reducer code is covered over gloo in test_dp_gpu.py / test_dp_gloo.py."""
import ctypes
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_native_rccl_communicator_single_rank(dev):
    from promptttspp_amd import _lib, ops
    from promptttspp_amd.parallel import NativeComm

    c = NativeComm(0, 1)
    try:
        for dt in (torch.float32, torch.bfloat16):
            x = torch.randn(1 << 20, device=dev).to(dt)
            ref = x.clone()
            c.allreduce_mean(x, ops._stream())   # mean over one rank: identity
            c.broadcast(x, 0, ops._stream())
            torch.cuda.synchronize()
            assert torch.equal(x, ref)
        lib = _lib.load()
        assert lib.ptpp_allreduce_mean(None, 4, _lib.F32, c.handle, None) == -1 and b"allreduce_mean" in lib.ptpp_last_error()
        assert lib.ptpp_allreduce_mean(x.data_ptr(), 4, 7, c.handle, None) == -1
    finally:
        c.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(port):
    import torch.distributed as dist

    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.dirname(here)):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PTPP_DP_FORCE_COLLECTIVES="1")
    from promptttspp_amd import functional as PF

    torch.cuda.set_device(0)
    PF.create_side_stream(torch.device("cuda", 0))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        from promptttspp_amd.parallel import FlatGradReducer

        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        lin = [torch.nn.Linear(256, 256).to(dev) for _ in range(6)]
        params = [p for m in lin for p in m.parameters()]
        x = torch.randn(64, 256, device=dev)

        def grads(**kw):
            for p in params:
                p.grad = None
            red = FlatGradReducer(params, bucket_elems=100_000, direct=False, **kw)
            assert len(red.buckets) >= 3
            red.zero_grad()
            h = x
            for m in lin:
                h = torch.tanh(m(h))
            h.square().mean().backward()
            red.finish()
            torch.cuda.synchronize()
            return torch.cat([p.grad.flatten().clone() for p in params]), red

        ref, _ = grads()                                  # torch.distributed all-reduce over RCCL
        for kw in (dict(algo="rs_ag"), dict(backend="native")):
            got, red = grads(**kw)
            assert torch.equal(got, ref), kw             # one rank: every exchange is the identity, bit for bit
            if red.native is not None:
                red.broadcast_parameters(lin[0])
                red.native.close()
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_reducer_exchange_variants_over_rccl_one_rank():
    """FlatGradReducer's three exchange variants (all-reduce, reduce-scatter + all-gather, the native
    communicator) with one rank over RCCL, hooks and bucket launches active (PTPP_DP_FORCE_COLLECTIVES)."""
    ctx = torch.multiprocessing.get_context("spawn")
    p = ctx.Process(target=_worker, args=(_free_port(),))
    p.start()
    p.join(timeout=600)
    assert p.exitcode == 0
