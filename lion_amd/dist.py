"""Data-parallel gradient step over RCCL / xGMI -- replaces ``utils.average_gradients``,
``utils.broadcast_params`` and ``utils.init_processes`` (utils/utils.py:717-770, :1129-1163).

The reference flattens ALL gradients into one buffer AFTER backward has finished and issues a
single all-reduce (no overlap, two extra full copies).  On an MI355X node every GPU has 7
point-to-point xGMI links, and a ring all-reduce of the 354 MB prior gradient is bound by one link
(~4 ms).  ``BucketedGradAverager`` instead:
  * packs parameters, in REVERSE registration order (the order backward produces gradients),
    into ~32 MiB flat buckets whose storage the ``.grad`` tensors alias (no copy in, no copy out);
  * launches each bucket's all-reduce from a side HIP stream as soon as its last gradient has been
    accumulated (``register_post_accumulate_grad_hook``), so communication overlaps the rest of the
    PVCNN backward;
  * pre-divides by the world size like the reference (:734-738) so the result is the mean.
``finish()`` (call before ``optimizer.step``) waits for the outstanding buckets.
Works with any ``torch.distributed`` backend ('nccl' is RCCL on ROCm; the CPU tests use 'gloo').
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_processes(rank: int, world_size: int, backend: str | None = None, master_addr: str = "127.0.0.1",
                   master_port: int = 6020):
    """env:// rendezvous, one process per GPU (utils/utils.py:1129-1163)."""
    os.environ.setdefault("MASTER_ADDR", master_addr)
    os.environ.setdefault("MASTER_PORT", str(master_port))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if torch.cuda.is_available():
        torch.cuda.set_device(rank % torch.cuda.device_count())
    dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world_size)
    dist.barrier()


def broadcast_params(params, is_distributed=True, src=0):
    """One flat broadcast instead of one message per tensor (utils/utils.py:767-770)."""
    if not is_distributed or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    params = [p for p in params]
    if not params:
        return
    flat = torch.cat([p.data.reshape(-1) for p in params])
    dist.broadcast(flat, src)
    off = 0
    with torch.no_grad():  # through the parameter: bumps p._version so derived (packed) weights are rebuilt
        for p in params:
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p))
            off += n
    from . import _wcache
    _wcache.invalidate_all()


def average_gradients(params, is_distributed=True):
    """Drop-in for utils.average_gradients (:717-748): mean of the gradients over ranks, one flat
    buffer, no overlap.  Kept for call sites that cannot use the bucketed averager."""
    if not is_distributed or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    grads = [p.grad for p in params if p.requires_grad and p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads]) / float(dist.get_world_size())
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


class BucketedGradAverager:
    def __init__(self, params, bucket_bytes: int = 32 << 20, overlap: bool = True):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        self.overlap = bool(overlap)   # hooks are armed at world size 1 too (same code path; nothing to reduce)
        self.buckets = []      # (flat tensor, [params])
        self._pending = {}     # bucket id -> grads still missing this step
        self._bucket_of = {}
        self._view_of = {}     # param -> its slice of the bucket
        self._seen = set()     # params whose gradient landed this step
        self._works = []
        self._hooks = []
        self._stream = None
        self._launched = set()
        # False: the hooks only keep the books (who got a gradient, aliasing); the buckets are reduced by finish() /
        # reduce_all().  GraphedTrainStep's split mode sets it: a backend whose collectives cannot be stream-captured
        # (gloo) must not be called from inside the captured backward.
        self.launch_in_hooks = True
        if not self.params:
            return
        dev = self.params[0].device
        if dev.type == "cuda":
            self._stream = torch.cuda.Stream(device=dev)
        cur, cur_bytes = [], 0
        for p in reversed(self.params):  # backward produces gradients roughly in this order
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                self._make_bucket(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._make_bucket(cur)
        if self.overlap:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._reset()

    def _make_bucket(self, plist):
        n = sum(p.numel() for p in plist)
        flat = torch.zeros(n, dtype=plist[0].dtype, device=plist[0].device)
        off = 0
        for p in plist:
            self._view_of[p] = flat[off:off + p.numel()].view_as(p)
            p.grad = self._view_of[p]  # .grad aliases the bucket: no pack/unpack copies
            off += p.numel()
            self._bucket_of[p] = len(self.buckets)
        self.buckets.append((flat, list(plist)))

    def _reset(self):
        self._pending = {i: len(pl) for i, (_, pl) in enumerate(self.buckets)}
        self._works = []
        self._seen = set()
        self._launched = set()

    def _launch(self, i):
        flat, _ = self.buckets[i]
        self._launched.add(i)
        if self.world == 1:
            return
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._stream):
                flat.div_(float(self.world))
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        else:
            flat.div_(float(self.world))
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        self._works.append(work)

    def _realias(self, p):
        """`p.grad` must live inside its bucket.  ``zero_grad(set_to_none=True)`` (torch.optim's default) or a
        ``p.grad = ...`` assignment breaks the aliasing: autograd then accumulates into a fresh tensor while the
        bucket that gets all-reduced stays zero and the ranks silently diverge.  Heal it: move the gradient into
        the bucket and point ``.grad`` back at the view."""
        view = self._view_of[p]
        g = p.grad
        if g is None or g.data_ptr() == view.data_ptr():
            return
        view.copy_(g)
        p.grad = view

    def _on_grad(self, p):
        i = self._bucket_of[p]
        if p in self._seen:
            raise RuntimeError("BucketedGradAverager: a parameter received a second gradient before finish(); "
                               "two backward passes per step (gradient accumulation) need overlap=False")
        self._seen.add(p)
        self._realias(p)
        self._pending[i] -= 1
        if self._pending[i] == 0 and self.launch_in_hooks:
            self._launch(i)

    def zero_grad(self):
        """zero the buckets and (re-)attach every ``.grad`` to its bucket view (parameters that got no gradient
        in the previous step were detached by ``finish``; a foreign ``zero_grad(set_to_none=True)`` detaches all)."""
        for flat, _ in self.buckets:
            flat.zero_()
        for p, view in self._view_of.items():
            if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                p.grad = view

    def finish(self):
        """Wait for (or, without overlap, perform) the averaging of every bucket.  Afterwards parameters that
        received no gradient in this step have ``grad = None``, like in the reference, whose averaging and
        optimizers skip them (utils/utils.py:725-727): Adam / EMA must not step them with zeros."""
        if not self.overlap:  # no hooks: find out who got a gradient, heal broken aliasing
            for p, view in self._view_of.items():
                g = p.grad
                if g is None:
                    continue
                if g.data_ptr() != view.data_ptr():
                    self._realias(p)
                    self._seen.add(p)
            touched = None    # without hooks "received a gradient" cannot be told from "stayed zero"
        else:
            touched = self._seen
        if self.world > 1:
            if not self.overlap:
                for i in range(len(self.buckets)):
                    self._launch(i)
            else:  # parameters that received no gradient this step leave their bucket pending
                for i in range(len(self.buckets)):
                    if i not in self._launched:
                        self._launch(i)
            for w in self._works:
                w.wait()
            if self._stream is not None:
                torch.cuda.current_stream().wait_stream(self._stream)
        if touched is not None:
            for p in self.params:
                if p not in touched:
                    p.grad = None
        self._reset()

    def reduce_all(self):
        """average every bucket now, on the current stream's timeline, and wait -- no bookkeeping (which parameters take
        part was fixed when the step was captured): what a replayed [forward + backward] graph is followed by when the
        collectives themselves are not capturable"""
        if self.world == 1:
            return
        works = []
        for flat, _ in self.buckets:
            flat.div_(float(self.world))
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
        for w in works:
            w.wait()

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
