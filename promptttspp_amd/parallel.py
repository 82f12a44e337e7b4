"""Data-parallel gradient exchange for one-process-per-GPU training over RCCL/xGMI.

The reference wraps the model in DistributedDataParallel (trainers/tts.py:116-117):
a broadcast of parameters at construction and a bucketed mean all-reduce of ~74.6 M
f32 gradients (298 MB) per step.  Here the gradients of all trainable parameters
live in ONE flat f32 buffer (``p.grad`` are views), cut into a few LARGE buckets --
xGMI is point-to-point (7 links/GPU), so few big collectives beat many 25 MB ones --
and each bucket's all-reduce is issued from a post-accumulate hook as soon as the
last gradient of the bucket has been written, overlapping the rest of backward.
The flat buffer also gives the fused optimiser a single zero-fill and stable
pointers (no per-step pointer-table rebuilds).

With ``direct=True`` (default) the weight-gradient kernels accumulate straight into the
views of the flat buffer (functional.enable_direct_grads): no per-parameter zero-fill,
temporary or ``grad +=`` launch.  Autograd's post-accumulate hooks do not fire for
gradients written that way, so functional notifies ``_hook`` itself.
"""
import ctypes
import os

import torch
import torch.distributed as dist

_ALIGN = 64  # bucket lengths are multiples of this many floats, so a bucket splits evenly over <= 64 ranks


class NativeComm:
    """The C ABI's own RCCL communicator (include/ptpp.h: ptpp_comm_init / ptpp_allreduce_mean /
    ptpp_broadcast -- SURVEY section 8b): rank 0 draws the unique id, the other ranks receive it through the
    torch.distributed store of the already initialised default group, every rank joins.  Selected with
    ``PTPP_DP_BACKEND=native`` (default: torch.distributed's collectives, the same RCCL underneath)."""

    def __init__(self, rank, world, group=None):
        from . import _lib

        self.lib, self.rank, self.world = _lib.load(), rank, world
        idbuf = ctypes.create_string_buffer(128)
        if rank == 0:
            _lib.check(self.lib.ptpp_comm_unique_id(idbuf), "ptpp_comm_unique_id")
        if world > 1:
            box = [idbuf.raw if rank == 0 else None]
            src = dist.get_global_rank(group, 0) if group is not None else 0  # (src is a GLOBAL rank)
            dist.broadcast_object_list(box, src=src, group=group)
            idbuf = ctypes.create_string_buffer(box[0], 128)
        self.handle = ctypes.c_void_p()
        _lib.check(self.lib.ptpp_comm_init(rank, world, idbuf, ctypes.byref(self.handle)), "ptpp_comm_init")

    def allreduce_mean(self, t, stream):
        from . import _lib, ops

        _lib.check(self.lib.ptpp_allreduce_mean(t.data_ptr(), t.numel(), ops.dtype_code(t.dtype), self.handle, stream),
                   "ptpp_allreduce_mean")

    def broadcast(self, t, src, stream):
        from . import _lib, ops

        _lib.check(self.lib.ptpp_broadcast(t.data_ptr(), t.numel(), ops.dtype_code(t.dtype), src, self.handle, stream),
                   "ptpp_broadcast")

    def close(self):
        if self.handle:
            self.lib.ptpp_comm_destroy(self.handle)
            self.handle = ctypes.c_void_p()


class FlatGradReducer:
    """Knobs for the first multi-GPU sweeps (environment, so the driver's bench command line stays fixed):
    ``PTPP_DP_BUCKET_MB`` (bucket size, default 128), ``PTPP_DP_ALGO`` = ``allreduce`` (default) | ``rs_ag``
    (reduce-scatter + all-gather of each bucket, in place), ``PTPP_DP_BACKEND`` = ``torch`` (default) | ``native``
    (the C ABI's RCCL communicator), ``PTPP_DP_BROADCAST_BUFFERS=0`` (switch OFF DDP's per-forward broadcast of the
    BatchNorm running statistics, trainers/tts.py:117 ``broadcast_buffers=True`` default; ON by default since round 4, ONE concatenated collective per
    step; when switched off each rank keeps its own statistics and rank 0's are checkpointed -- DESIGN.md section 6)."""

    def __init__(self, params, bucket_elems=None, process_group=None, direct=True, algo=None, backend=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        taper_floor = 0
        if bucket_elems is None:
            bucket_elems = int(float(os.environ.get("PTPP_DP_BUCKET_MB", "128")) * (1 << 20)) // 4
            # Round 6: the buffer is laid out in backward order, so the LAST buckets are the ones whose all-reduce nothing hides
            # (the phone encoder's ~45 M parameters are differentiated in the last 3.3 ms of the step).  Tapered layout: every
            # bucket half the size of the one before, down to PTPP_DP_BUCKET_MIN_MB (default 32; 0 = constant size): 128, 64, 32,
            # 32, 32, 10 MB for this model instead of 128, 128, 42 -- six collectives, of which only the small last ones can be
            # exposed.  An explicit ``bucket_elems`` argument keeps a constant size.
            taper_floor = int(float(os.environ.get("PTPP_DP_BUCKET_MIN_MB", "32")) * (1 << 20)) // 4
        self.algo = algo or os.environ.get("PTPP_DP_ALGO", "allreduce")
        assert self.algo in ("allreduce", "rs_ag"), self.algo
        self.backend = backend or os.environ.get("PTPP_DP_BACKEND", "torch")
        assert self.backend in ("torch", "native"), self.backend
        self.native = None
        # PTPP_DP_FORCE_COLLECTIVES=1 (diagnostics): run the multi-rank machinery -- hooks, bucket all-reduces,
        # finish() -- at world size 1 too, e.g. one rank over RCCL to look at stream interplay on a 1-GPU box
        self.collective = self.world > 1 or (dist.is_available() and dist.is_initialized()
                                             and bool(os.environ.get("PTPP_DP_FORCE_COLLECTIVES")))
        dev = self.params[0].device
        # Backward produces gradients roughly in reverse registration order: lay the
        # buffer out in that order so buckets complete front to back.  Every bucket ends on a multiple of
        # _ALIGN floats (a few zero floats of padding) so that it splits evenly over the ranks (rs_ag).
        self.buckets = []  # [start, end, n_params]
        off, start, count = 0, 0, 0
        self._bucket_of = {}
        slots = []
        for p in reversed(self.params):
            n = p.numel()
            slots.append((p, off, n))
            self._bucket_of[id(p)] = len(self.buckets)
            off += n
            count += 1
            if off - start >= bucket_elems:
                off = (off + _ALIGN - 1) // _ALIGN * _ALIGN
                self.buckets.append([start, off, count])
                start, count = off, 0
                if taper_floor > 0:
                    bucket_elems = max(bucket_elems // 2, min(taper_floor, bucket_elems))
        if count:
            off = (off + _ALIGN - 1) // _ALIGN * _ALIGN
            self.buckets.append([start, off, count])
        self.flat = torch.zeros(off, device=dev, dtype=torch.float32)
        for p, o, n in slots:
            p.grad = self.flat[o : o + n].view_as(p)
        self._pending = [b[2] for b in self.buckets]
        self._works = []
        self._launched = [False] * len(self.buckets)
        self._seen = set()
        self._next = 0
        if self.collective:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)
            if self.backend == "native":
                assert dev.type == "cuda", "the native RCCL communicator needs device tensors"
                self.native = NativeComm(dist.get_rank(process_group), self.world, process_group)
        if direct and dev.type == "cuda":
            from . import functional as PF

            # The weight-gradient side stream (functional.wgrad_stream): at world size 1, and with several ranks
            # over RCCL, whose collectives are stream-ordered device work -- every bucket all-reduce is issued
            # FROM the side stream once it has caught up with the main one (_launch), so the main stream never
            # waits.  Measured with one rank over RCCL on the 1-GPU box (PTPP_DP_FORCE_COLLECTIVES=1): 25.7-27.2
            # ms/step against 28.2-28.5 on one stream -- provided the side stream exists BEFORE RCCL creates its
            # streams (functional.create_side_stream; created after them it shared the main stream's hardware
            # queue and nothing overlapped).  gloo stages device tensors through the host and blocks the
            # launching thread: same gradients (tests/test_dp_gpu.py) but 4.7x slower with the side stream, so
            # gloo runs stay on one stream (PTPP_FORCE_ASYNC_WGRAD=1 overrides).
            backend = dist.get_backend(process_group) if self.collective else None
            PF.enable_direct_grads(True, notify=self._hook if self.collective else None,
                                   async_wgrad=(not self.collective or backend == "nccl" or self.native is not None
                                                or bool(os.environ.get("PTPP_FORCE_ASYNC_WGRAD"))))

    # -- parameter broadcast (DDP constructor semantics) ------------------------------
    def broadcast_parameters(self, module, src=0):
        if self.world == 1:
            return
        with torch.no_grad():
            for t in list(module.parameters()) + list(module.buffers()):
                self._bcast(t, src)

    def _bcast(self, t, src):
        if self.native is not None and t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and t.is_contiguous():
            from . import ops

            self.native.broadcast(t, src, ops._stream())
        else:
            dist.broadcast(t, src, group=self.group)

    def broadcast_buffers(self, module, src=0):
        """DDP's ``broadcast_buffers=True`` (the reference's default, trainers/tts.py:117): before a forward every
        rank takes rank 0's BatchNorm running statistics.  On by default since round 4 like the reference's DDP (``PTPP_DP_BROADCAST_BUFFERS=0`` switches it off): it only changes
        what eval-mode BatchNorm would see on ranks other than 0, which never evaluate or checkpoint.  All
        statistics travel as ONE concatenated tensor (one collective per step instead of ~40)."""
        if self.world == 1:
            return
        # every floating-point buffer of the BatchNorm layers' owners (running statistics; the integer
        # num_batches_tracked counters advance identically on all ranks and are left alone)
        bufs = [b for m in module.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)
                for b in m.buffers(recurse=False) if b is not None and b.is_floating_point()]
        if not bufs:
            return
        with torch.no_grad():
            cat = torch.cat([b.reshape(-1).float() for b in bufs])
            self._bcast(cat, src)
            off = 0
            for b in bufs:
                b.copy_(cat[off : off + b.numel()].view_as(b))
                off += b.numel()

    def _reduce(self, t):
        """Sum (torch backend; finish() divides) or mean (native backend) of ``t`` over the ranks, in place, issued
        on the current stream; returns a Work handle or None."""
        if self.native is not None:
            self.native.allreduce_mean(t, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            return None
        if self.algo == "rs_ag" and self.world > 1 and t.numel() % self.world == 0 and t.is_cuda and \
                dist.get_backend(self.group) == "nccl":  # (gloo has no reduce-scatter)
            r = dist.get_rank(self.group)
            c = t.numel() // self.world
            mine = t[r * c : (r + 1) * c]
            dist.reduce_scatter_tensor(mine, t, op=dist.ReduceOp.SUM, group=self.group)
            return dist.all_gather_into_tensor(t, mine, group=self.group, async_op=True)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _launch(self, bi):
        a, b, _ = self.buckets[bi]
        self._launched[bi] = True
        side = None
        if self.flat.is_cuda:
            from . import functional as PF
            from . import ops

            ops.red_flush()  # the queued parameter-gradient sums of every stream, each on its producer's stream
            side = PF.side_stream_for_collective()
            if side is None:
                PF.sync_wgrad_stream()  # the collective orders itself after the CURRENT stream only
        if side is not None:
            # weight gradients run on the side stream: issue the collective FROM that stream (after it has
            # caught up with this one), so that the main stream -- the critical path -- never waits for it
            with torch.cuda.stream(side):
                w = self._reduce(self.flat[a:b])
        else:
            w = self._reduce(self.flat[a:b])
        if w is not None:
            self._works.append(w)

    def _hook(self, p):
        # A parameter can be reported twice in one step: by functional's direct-accumulation path when
        # its last weight-gradient kernel has been enqueued, and by autograd's post-accumulate hook
        # (which this PyTorch also fires for the None gradient those functions return).  Count once.
        pid = id(p)
        bi = self._bucket_of.get(pid)
        if bi is None or self._launched[bi] or pid in self._seen:
            return
        self._seen.add(pid)
        self._pending[bi] -= 1
        # collectives must be issued in the SAME order on every rank: buckets go out strictly in index
        # order (the buffer is laid out in backward order, so this costs little overlap)
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            self._launch(self._next)
            self._next += 1

    def close(self):
        """Release the native RCCL communicator (ncclCommDestroy); the trainer calls it at teardown."""
        if self.native is not None:
            self.native.close()
            self.native = None

    def zero_grad(self):
        self.flat.zero_()
        if self.flat.is_cuda:
            from . import functional as PF

            PF.reset_direct_uses()
        self._pending = [b[2] for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._works = []
        self._seen = set()
        self._next = 0

    # -- diagnostics of the exchange (bench.py --gpus N: the first 2/4/8-GPU run must tell where the time goes) ----------------
    def enable_timing(self, on=True):
        """Record three events per ``finish()`` on the calling stream: entry, gradient streams joined, collectives done.
        ``timing_summary()`` turns them into the time the MAIN stream spent joining and -- beyond that -- waiting for
        all-reduces that backward did not hide (``exposed``)."""
        self._timing = [] if (on and self.flat.is_cuda) else None

    def timing_summary(self, skip=0):
        ev = getattr(self, "_timing", None) or []
        torch.cuda.synchronize()
        join = [a.elapsed_time(b) for a, b, _ in ev[skip:]]
        exposed = [b.elapsed_time(c) for _, b, c in ev[skip:]]
        mean = lambda v: round(sum(v) / len(v), 4) if v else None
        be = (dist.get_backend(self.group) if self.collective else None)
        return {"join_gradient_streams_ms": mean(join), "exposed_allreduce_ms": mean(exposed),
                "exposed_allreduce_ms_max": round(max(exposed), 4) if exposed else None, "steps": len(join),
                "buckets": len(self.buckets), "bucket_mb": [round((b - a) * 4 / 2**20, 1) for a, b, _ in self.buckets],
                "backend": ("native-rccl" if self.native is not None else be), "algo": self.algo, "world": self.world}

    def finish(self):
        """Join the weight-gradient side stream, wait for the bucket all-reduces, turn sums into means."""
        tm = getattr(self, "_timing", None)
        if tm is not None:
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
        if self.flat.is_cuda:
            from . import functional as PF
            from . import ops

            ops.red_flush()
            PF.sync_wgrad_stream()
        if tm is not None:
            e1.record()
        if not self.collective:
            if tm is not None:
                e2.record()
                tm.append((e0, e1, e2))
            return
        for bi in range(self._next, len(self.buckets)):  # buckets with parameters that received no gradient
            if not self._launched[bi]:
                self._launch(bi)
        self._next = len(self.buckets)
        for w in self._works:
            w.wait()
        if tm is not None:
            e2.record()
            tm.append((e0, e1, e2))
        if self.native is None:
            self.flat.mul_(1.0 / self.world)
        elif self.flat.is_cuda:
            from . import functional as PF

            PF.sync_wgrad_stream()  # the native collectives are plain stream work on the side stream
