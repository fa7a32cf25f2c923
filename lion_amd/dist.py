"""Data-parallel gradient step over RCCL / xGMI -- replaces ``utils.average_gradients``,
``utils.broadcast_params`` and ``utils.init_processes`` (utils/utils.py:717-770, :1129-1163).

The reference flattens ALL gradients into one buffer AFTER backward has finished and issues a
single all-reduce (no overlap, two extra full copies).  On an MI355X node every GPU has 7
point-to-point xGMI links, and a ring all-reduce of the 354 MB prior gradient is bound by one link
(~4 ms).  ``BucketedGradAverager`` instead:
  * starts every step with ``grad = None`` (``zero_grad``): autograd then hands each parameter the gradient tensor its
    backward node produced, as it is -- no zero-fill of the gradients and no ``grad += new`` kernel per parameter (867
    tiny launches per VAE step, ~4 ms of an 89 ms step, when ``.grad`` pre-exists);
  * packs parameters, in REVERSE registration order (the order backward produces gradients), into ~32 MiB flat
    buckets; as soon as the last gradient of a bucket has landed (``register_post_accumulate_grad_hook``) ONE
    multi-tensor copy moves the bucket's gradients into the flat buffer, ``.grad`` is pointed at the views, and the
    bucket's all-reduce is launched from a side HIP stream, so communication overlaps the rest of the PVCNN backward;
  * pre-divides by the world size like the reference (:734-738) so the result is the mean;
  * at world size 1 has nothing to move: no flat buffers are allocated and ``.grad`` stays what autograd produced.
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
        flat = torch.zeros(n, dtype=plist[0].dtype, device=plist[0].device) if self.world > 1 else None
        off = 0
        for p in plist:
            self._view_of[p] = flat[off:off + p.numel()].view_as(p) if flat is not None else None
            off += p.numel()
            self._bucket_of[p] = len(self.buckets)
        self.buckets.append((flat, list(plist)))

    def _reset(self):
        self._pending = {i: len(pl) for i, (_, pl) in enumerate(self.buckets)}
        self._works = []
        self._seen = set()
        self._launched = set()

    def _bind(self, i):
        """move the gradients bucket i's parameters hold into its flat buffer (one multi-tensor copy) and point ``.grad``
        at the views; gradients already living there (a caller that kept ``.grad`` and zeroed it in place) are left alone.
        Parameters without a gradient keep ``grad = None``; their slice of the buffer is reduced with whatever it held
        and never read."""
        flat, plist = self.buckets[i]
        if flat is None:
            return
        src, dst, who = [], [], []
        for p in plist:
            g, view = p.grad, self._view_of[p]
            if g is None or g.data_ptr() == view.data_ptr():
                continue
            src.append(g.detach())
            dst.append(view)
            who.append(p)
        if not who:
            return
        with torch.no_grad():
            torch._foreach_copy_(dst, src)
        for p in who:
            p.grad = self._view_of[p]

    def bind_all(self):
        """every gradient produced so far into its bucket -- the tail of a captured [forward + backward] graph whose
        collectives run eagerly behind it (GraphedTrainStep's split mode): a bucket with a parameter that got no gradient
        never completes inside the hooks"""
        for i in range(len(self.buckets)):
            self._bind(i)

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

    def _on_grad(self, p):
        i = self._bucket_of[p]
        if p in self._seen:
            raise RuntimeError("BucketedGradAverager: a parameter received a second gradient before finish(); "
                               "two backward passes per step (gradient accumulation) need overlap=False")
        self._seen.add(p)
        self._pending[i] -= 1
        if self._pending[i] == 0:
            self._bind(i)
            if self.launch_in_hooks:
                self._launch(i)

    def zero_grad(self):
        """``grad = None`` for every parameter: the backward's own gradient tensors are adopted as they are (module
        docstring) -- nothing to zero, nothing to accumulate into."""
        for p in self.params:
            p.grad = None

    def finish(self):
        """Wait for (or, without overlap, perform) the averaging of every bucket.  Afterwards parameters that
        received no gradient in this step have ``grad = None``, like in the reference, whose averaging and
        optimizers skip them (utils/utils.py:725-727): Adam / EMA must not step them with zeros; at world size > 1
        every other ``.grad`` is a view of its bucket."""
        if self.world > 1:
            for i in range(len(self.buckets)):
                if i not in self._launched:   # no hooks, hooks told not to launch, or a parameter of the bucket got no gradient
                    self._bind(i)
                    self._launch(i)
            for w in self._works:
                w.wait()
            if self._stream is not None:
                torch.cuda.current_stream().wait_stream(self._stream)
        if self.overlap:   # a caller that kept .grad alive across steps (foreign zero_grad(set_to_none=False)): drop the untouched
            for p in self.params:
                if p not in self._seen:
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
