"""The optimizer of the training steps (reference utils/utils.py:115-121: ``optim.Adam(params, lr, betas, weight_decay)``, stepped by
trainers/hvae_trainer.py:150-154 and train_2prior.py:405-410) with the whole update in ONE launch (csrc/optim.hip).

``Adam`` is a ``torch.optim.Optimizer`` with torch.optim.Adam's hyper-parameters, state keys (``step``, ``exp_avg``, ``exp_avg_sq``: a
state_dict moves between the two) and arithmetic (its single-tensor path, fp32 op for op; L2 weight decay; per-parameter step counts;
parameters without a gradient are skipped).  Differences in form: the step counts and the learning rate live in device memory (so a
captured step replays: call ``sync_lr()`` after changing ``group['lr']`` when steps are replayed rather than run), and the addresses of
{param, grad, exp_avg, exp_avg_sq, step, ema} of every tensor sit in a device table that is rewritten only when a gradient moved.
``ema_decay > 0``: the moving average of the weights that the reference's ``EMA`` wrapper keeps around every optimizer
(utils/ema.py:31-89, ``state[p]['ema']``; decay 0.9999 in every released config) is updated by the same launch -- the wrapper's
form is a stack, a multiply-add and an unstack per parameter shape after the step.  ``lion_amd.training.EMA`` around this optimizer
hands its decay over and keeps ``swap_parameters_with_ema``.  float32 HIP parameters only -- anything else is an error, not a
fallback."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


class _Plan:
    __slots__ = ("key", "table", "numel", "blockmap", "blocks", "host")


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, ema_decay=0.0):
        if not 0.0 <= ema_decay <= 1.0:
            raise ValueError(f"Invalid ema_decay: {ema_decay}")
        self.ema_decay = float(ema_decay)
        if not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"Invalid betas: {betas}")
        if not 0.0 <= weight_decay:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._plans = {}       # group index -> _Plan
        self._lr_dev = {}      # group index -> (device f32[1], the host value it holds)
        self._captured = []    # plans written inside a stream capture: their pinned tables are the source of the graph's copy nodes
        self._chunk = self._row = None

    # -- device-resident hyper-parameters ---------------------------------------------------------------------------
    def _lr(self, gi, group, dev):
        cur = self._lr_dev.get(gi)
        lr = float(group["lr"])
        if cur is None:
            cur = (torch.full((1,), lr, device=dev, dtype=torch.float32), lr)
            self._lr_dev[gi] = cur
        elif cur[1] != lr and not torch.cuda.is_current_stream_capturing():
            cur[0].fill_(lr)
            cur = (cur[0], lr)
            self._lr_dev[gi] = cur
        return cur[0]

    def sync_lr(self):
        """push every group's ``lr`` to the device (replayed steps read it there)"""
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.is_cuda]
            if ps:
                self._lr(gi, group, ps[0].device)

    # -- the pointer table ------------------------------------------------------------------------------------------------
    def _plan(self, gi, ps, grads, dev):
        ema = self.ema_decay > 0.0
        key = tuple((p.data_ptr(), g.data_ptr(), p.numel(), self.state[p]["ema"].data_ptr() if ema else 0) for p, g in zip(ps, grads))
        plan = self._plans.get(gi)
        if plan is not None and plan.key == key:
            return plan
        if self._chunk is None:
            self._chunk, self._row = int(_lib.load().lion_adam_chunk()), int(_lib.load().lion_adam_row())
        T = len(ps)
        table = np.empty((T, self._row), dtype=np.uint64)
        numel = np.empty((T,), dtype=np.int32)
        for i, (p, g) in enumerate(zip(ps, grads)):
            st = self.state[p]
            table[i] = (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(),
                        st["ema"].data_ptr() if ema else 0)
            numel[i] = p.numel()
        per = (numel.astype(np.int64) + self._chunk - 1) // self._chunk
        blocks = int(per.sum())
        blockmap = np.empty((blocks, 2), dtype=np.int32)
        blockmap[:, 0] = np.repeat(np.arange(T, dtype=np.int32), per)
        start = np.concatenate(([0], np.cumsum(per)[:-1]))
        blockmap[:, 1] = (np.arange(blocks, dtype=np.int64) - np.repeat(start, per)).astype(np.int32)
        # ONE pinned buffer -> ONE device buffer: [table | numel | blockmap] as bytes (a single copy node when captured)
        raw = np.concatenate((table.view(np.uint8).reshape(-1), numel.view(np.uint8).reshape(-1),
                              blockmap.view(np.uint8).reshape(-1)))
        host = torch.from_numpy(raw).pin_memory()
        devbuf = torch.empty(raw.size, dtype=torch.uint8, device=dev)
        devbuf.copy_(host, non_blocking=True)
        plan = _Plan()
        plan.key, plan.host, plan.blocks = key, host, blocks
        o1, o2 = table.nbytes, table.nbytes + numel.nbytes
        plan.table, plan.numel, plan.blockmap = devbuf[:o1], devbuf[o1:o2], devbuf[o2:]
        self._plans[gi] = plan
        if torch.cuda.is_current_stream_capturing():
            self._captured.append(plan)
        return plan

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            grads = []
            for p in ps:
                g = p.grad
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.device == dev):
                    raise RuntimeError("lion_amd.optim.Adam: float32, contiguous HIP parameters on one device only "
                                       f"(got {p.dtype}, {p.device}, contiguous={p.is_contiguous()})")
                if g.is_sparse or g.dtype != torch.float32 or g.device != dev:
                    raise RuntimeError("lion_amd.optim.Adam: dense float32 gradients on the parameter's device only")
                if not g.is_contiguous():
                    g = g.contiguous()
                    p.grad = g
                grads.append(g)
                st = self.state[p]
                if len(st) == 0 or "exp_avg" not in st:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=dev)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                elif not torch.is_tensor(st["step"]) or st["step"].device != dev:   # a state_dict from torch.optim.Adam(capturable=False)
                    st["step"] = torch.full((), float(st["step"]), dtype=torch.float32, device=dev)
                if self.ema_decay > 0.0 and "ema" not in st:
                    # the kernel starts the average from the updated parameter at the parameter's FIRST step (utils/ema.py:58-59);
                    # state that arrives with steps already taken but no average (a plain Adam checkpoint) starts from the
                    # parameter as it is
                    st["ema"] = p.detach().clone(memory_format=torch.contiguous_format)
            plan = self._plan(gi, ps, grads, dev)
            b1, b2 = group["betas"]
            _lib.check(lib.lion_adam_step(_lib.ptr(plan.table), _lib.ptr(plan.numel), _lib.ptr(plan.blockmap), plan.blocks,
                                          len(ps), _lib.ptr(self._lr(gi, group, dev)), float(b1), float(b2), float(group["eps"]),
                                          float(group["weight_decay"]), self.ema_decay, _lib.stream_ptr(dev)), "adam_step")
        return loss
