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
import os

import torch
import torch.distributed as dist


class FlatGradReducer:
    def __init__(self, params, bucket_elems=32 * 1024 * 1024, process_group=None, direct=True):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        # PTPP_DP_FORCE_COLLECTIVES=1 (diagnostics): run the multi-rank machinery -- hooks, bucket all-reduces,
        # finish() -- at world size 1 too, e.g. one rank over RCCL to look at stream interplay on a 1-GPU box
        self.collective = self.world > 1 or (dist.is_available() and dist.is_initialized()
                                             and bool(os.environ.get("PTPP_DP_FORCE_COLLECTIVES")))
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        # Backward produces gradients roughly in reverse registration order: lay the
        # buffer out in that order so buckets complete front to back.
        self.buckets = []  # [start, end, n_params]
        off, start, count = 0, 0, 0
        self._bucket_of = {}
        for p in reversed(self.params):
            n = p.numel()
            p.grad = self.flat[off : off + n].view_as(p)
            self._bucket_of[id(p)] = len(self.buckets)
            off += n
            count += 1
            if off - start >= bucket_elems:
                self.buckets.append([start, off, count])
                start, count = off, 0
        if count:
            self.buckets.append([start, off, count])
        self._pending = [b[2] for b in self.buckets]
        self._works = []
        self._launched = [False] * len(self.buckets)
        self._seen = set()
        self._next = 0
        if self.collective:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)
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
                                   async_wgrad=(not self.collective or backend == "nccl"
                                                or bool(os.environ.get("PTPP_FORCE_ASYNC_WGRAD"))))

    # -- parameter broadcast (DDP constructor semantics) ------------------------------
    def broadcast_parameters(self, module, src=0):
        if self.world == 1:
            return
        with torch.no_grad():
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t, src, group=self.group)

    def _launch(self, bi):
        a, b, _ = self.buckets[bi]
        self._launched[bi] = True
        side = None
        if self.flat.is_cuda:
            from . import functional as PF

            side = PF.side_stream_for_collective()
            if side is None:
                PF.sync_wgrad_stream()  # the collective orders itself after the CURRENT stream only
        if side is not None:
            # weight gradients run on the side stream: issue the collective FROM that stream (after it has
            # caught up with this one), so that the main stream -- the critical path -- never waits for it
            with torch.cuda.stream(side):
                self._works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

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

    def finish(self):
        """Join the weight-gradient side stream, wait for the bucket all-reduces, turn sums into means."""
        if self.flat.is_cuda:
            from . import functional as PF

            PF.sync_wgrad_stream()
        if not self.collective:
            return
        for bi in range(self._next, len(self.buckets)):  # buckets with parameters that received no gradient
            if not self._launched[bi]:
                self._launch(bi)
        self._next = len(self.buckets)
        for w in self._works:
            w.wait()
        self.flat.mul_(1.0 / self.world)
