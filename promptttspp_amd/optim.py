"""FusedAdamW: the trainer's ``clip_grad_norm_(1.0)`` + ``AdamW.step()``
(reference: promptttspp/trainers/tts.py:206-211, conf/optimizer/adamw.yaml) as two
HIP launches over ALL parameters with no host synchronisation
(``ptpp_grad_sumsq`` + ``ptpp_adamw_step``).  Same constructor as
``torch.optim.AdamW`` (params, lr, betas, eps, weight_decay) plus ``max_grad_norm``;
LR schedulers keep working through ``param_groups[i]['lr']``."""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from .ops import _ptr, _stream

_BLOCK = 4096  # elements per block, must match adamw.hip (CHUNK)
_SLOTS = 64  # PTPP_SUMSQ_SLOTS


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=0.0):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.max_grad_norm = float(max_grad_norm)
        self.stable_grads = False  # set True by callers whose p.grad tensors live for the whole run (see _table)
        # gradient norm summed in a fixed order (ptpp_grad_sumsq_det).  Kept on for single-GPU runs too: a run is then
        # reproducible bit for bit, and its checkpoints do not depend on whether it was started with 1 or N ranks; False
        # selects the one-launch atomic form (ptpp_grad_sumsq)
        self.deterministic_norm = os.environ.get("PTPP_ADAMW_ATOMIC_NORM", "") in ("", "0")
        self._tables = {}
        self._sumsq = None
        self._partials = {}
        self._lr_dev = {}

    def _bump_list(self, gi, group):
        """The parameters of the group that are updated (have a grad): one list object per group, rebuilt only
        when the table is (its identity tells functional.repack_all that the same set was bumped again)."""
        ent = self._tables.get(gi)
        if ent is not None and len(ent) > 7:
            return ent[7]
        return [p for p in group["params"] if p.grad is not None]

    def _table(self, gi, group):
        """Device pointer table for the group's parameters that have grads."""
        ent = self._tables.get(gi)
        if ent is not None and self.stable_grads:
            # fast validity check (this runs every step over ~700 parameters): same gradient tensor OBJECTS as
            # when the table was built.  Only sound when the gradients are long-lived tensors (the views of
            # FlatGradReducer's flat buffer) -- with per-step gradient tensors an id() can be recycled -- so
            # the owner of such gradients opts in through ``stable_grads``.
            ps0, gids = ent[5], ent[6]
            if len(ps0) == len(group["params"]) and all(id(p.grad) == g for p, g in zip(ps0, gids)) and \
                    ps0[0].data_ptr() == ent[0][0][0] and ps0[-1].data_ptr() == ent[0][-1][0]:
                return ent[1:5]
        ps = [p for p in group["params"] if p.grad is not None]
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in ps)
        if ent is not None and ent[0] == key:
            return ent[1:5]
        rows, blk, owners = [], 0, []
        for p in ps:
            assert p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous() and p.is_cuda
            st = self.state[p]
            if "exp_avg" not in st:
                st["exp_avg"] = torch.zeros_like(p)
                st["exp_avg_sq"] = torch.zeros_like(p)
            n = p.numel()
            rows.append([p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), n, blk])
            nb = (n + _BLOCK - 1) // _BLOCK
            owners.append(np.full(nb, len(rows) - 1, dtype=np.int32))
            blk += nb
        tab = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(ps[0].device)
        bmap = torch.from_numpy(np.concatenate(owners)).to(ps[0].device)  # block -> record
        fast = ps if len(ps) == len(group["params"]) else []  # the id() shortcut needs every parameter to have a grad
        self._tables[gi] = (key, tab, len(ps), blk, bmap, fast, [id(p.grad) for p in fast], list(ps))
        return tab, len(ps), blk, bmap

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        lib = _lib.load()
        live = [(gi, g) for gi, g in enumerate(self.param_groups) if any(p.grad is not None for p in g["params"])]
        if not live:
            return loss
        dev = next(p for p in live[0][1]["params"] if p.grad is not None).device
        if dev.type == "cuda":
            from . import ops

            ops.red_flush()  # parameter-gradient sums still queued (no-op after FlatGradReducer.finish())
        if self._sumsq is None:
            self._sumsq = torch.zeros(_SLOTS, device=dev, dtype=torch.float32)
        if self.max_grad_norm > 0:
            # global norm over every group: accumulate group sums into one scalar
            total = None
            for gi, g in live:
                tab, nt, nblk, bmap = self._table(gi, g)
                part = self._sumsq if total is None else torch.zeros_like(self._sumsq)
                if self.deterministic_norm:
                    scratch = self._partials.get(gi)
                    if scratch is None or scratch.numel() < nblk or scratch.device != dev:
                        scratch = self._partials[gi] = torch.empty(nblk, device=dev, dtype=torch.float32)
                    # per-block partials summed in a fixed order: identical clip factors on every data-parallel rank and
                    # from run to run, one small launch more than the atomic form
                    _lib.check(lib.ptpp_grad_sumsq_det(_ptr(tab), nt, _ptr(bmap), nblk, _ptr(part), _ptr(scratch), _stream()),
                               "ptpp_grad_sumsq_det")
                else:
                    part.zero_()
                    _lib.check(lib.ptpp_grad_sumsq(_ptr(tab), nt, _ptr(bmap), nblk, _ptr(part), _stream()), "ptpp_grad_sumsq")
                total = part if total is None else total.add_(part)
            if total is not self._sumsq:
                self._sumsq.copy_(total)
        for gi, g in live:
            tab, nt, nblk, bmap = self._table(gi, g)
            g["step"] = g.get("step", 0) + 1
            lr_dev = self._lr_dev.get(gi)
            if lr_dev is None:
                lr_dev = self._lr_dev[gi] = torch.zeros(1, device=dev, dtype=torch.float32)
            lr_dev.fill_(float(g["lr"]))
            b1, b2 = g["betas"]
            _lib.check(
                lib.ptpp_adamw_step(_ptr(tab), nt, _ptr(bmap), nblk, _ptr(self._sumsq), _ptr(lr_dev), float(b1), float(b2),
                                    float(g["eps"]), float(g["weight_decay"]), int(g["step"]), self.max_grad_norm,
                                    _stream()),
                "ptpp_adamw_step",
            )
            # the kernel writes the parameters through raw pointers: tell autograd (and every cache
            # keyed on Tensor._version -- the packed-weight caches of functional.py) that they changed
            torch.autograd.graph.increment_version(self._bump_list(gi, g))
        from . import functional as PF

        # one launch refreshes every cached packed operand of the updated weights
        PF.repack_all(bumped=self._bump_list(live[0][0], live[0][1]) if len(live) == 1 and self.stable_grads else None)
        return loss

    # -- checkpoints interchangeable with torch.optim.AdamW (the reference's optimiser) in both directions: torch keeps
    # a per-parameter ``state[p]["step"]`` tensor, this class one counter per group -----------------------------------
    def state_dict(self):
        sd = super().state_dict()
        for g in sd["param_groups"]:
            step = float(g.get("step", 0))
            for pid in g["params"]:
                if pid in sd["state"]:
                    sd["state"][pid] = dict(sd["state"][pid], step=torch.tensor(step))
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for g in self.param_groups:
            steps = [float(self.state[p].pop("step")) for p in g["params"] if "step" in self.state.get(p, {})]
            if steps and not g.get("step"):
                g["step"] = int(max(steps))  # a torch.optim.AdamW checkpoint: continue its bias correction
        self._tables.clear()  # the moment tensors were replaced: rebuild the pointer tables
        self._partials.clear()
        self._lr_dev.clear()

    def grad_norm(self):
        """sqrt of the last computed sum of squares (device tensor; no sync)."""
        return self._sumsq.sum().sqrt() if self._sumsq is not None else None
