"""hipGraph capture of one denoiser evaluation.

One PVCNN2Prior forward is ~740 kernel launches (14 PVConv x [voxelize, 2 convs, folds, devoxelize,
point MLP] + 4 SA + 4 FP stages); at ~4-5 us of host time per launch the eager step is host-bound
for several milliseconds.  Every operator of liblion_hip.so is capture-safe by contract (caller's
stream, no allocation, no synchronisation), and the torch-side allocations land in the graph's private
pool, so the whole forward is captured once per (model, batch shape) and replayed per step; only the
inputs (x, t, condition) are copied into static buffers.
"""
from __future__ import annotations

import torch


class GraphedDenoiser:
    """callable with the denoisers' signature: model(x=..., t=..., condition_input=..., clip_feat=...)."""

    def __init__(self, model, x, t, condition_input=None, clip_feat=None, warmup: int = 2):
        self.model = model
        self.x = x.detach().clone()
        self.t = t.detach().clone()
        self.cond = None if condition_input is None else condition_input.detach().clone()
        self.clip = None if clip_feat is None else clip_feat.detach().clone()
        self.mixed_prediction = getattr(model, "mixed_prediction", False)
        self.mixing_logit = getattr(model, "mixing_logit", None)
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(warmup):  # first calls set function attributes, pack weights, fill caches
                model(x=self.x, t=self.t, condition_input=self.cond, clip_feat=self.clip)
        torch.cuda.current_stream(x.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = model(x=self.x, t=self.t, condition_input=self.cond, clip_feat=self.clip)

    def eval(self):
        return self

    def train(self, mode=True):
        return self

    def __call__(self, x, t, condition_input=None, clip_feat=None, **kwargs):
        self.x.copy_(x)
        self.t.copy_(t)
        if self.cond is not None and condition_input is not None and condition_input.data_ptr() != self.cond.data_ptr():
            self.cond.copy_(condition_input)
        if self.clip is not None and clip_feat is not None and clip_feat.data_ptr() != self.clip.data_ptr():
            self.clip.copy_(clip_feat)
        self.graph.replay()
        return self.out
