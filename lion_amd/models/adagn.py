"""Adaptive group normalisation (reference models/adagn.py:19-65):
    y = GroupNorm8(x) * (W s + b)[:C] + (W s + b)[C:]
Parameters: ``norm.{weight,bias}`` (GroupNorm(8, C)), ``emd.{weight,bias}`` (Linear(style, 2C))."""
import contextlib

import torch
import torch.nn as nn

from .dense import dense
from .. import _wcache

# Inference-time batching of the style projections: every AdaGN of a network projects the SAME style
# vector with its own Linear(style, 2C).  StylePlan concatenates those weights once and evaluates all
# projections of a forward pass with ONE GEMM ([B, D] x [D, sum 2C]) instead of one tiny GEMM per
# layer (75 launches per denoiser step in the released local prior); AdaGN.affine then returns
# strided views into that result.  Nothing is cached across forward passes.
_ACTIVE = None  # (style tensor, {id(module): (factor view, bias view)})
# Round 6 -- the same batching under autograd (training): one cat of the projection weights, ONE Linear and one split per forward
# instead of a Linear per AdaGN; backward = one cat of the slice gradients + the Linear's two GEMMs + narrow views onto the
# parameters' gradients, instead of (2 GEMMs + a bias sum + an accumulation into the style gradient) x ~140 layers per VAE step.
# Only modules that were actually CALLED in an earlier forward of this root are batched (an unused module must keep grad = None,
# as in the reference): the first training forward runs unbatched and records them.  LION_TRAIN_STYLE_PLAN=0 switches it off.
TRAIN_BATCHED = __import__("os").environ.get("LION_TRAIN_STYLE_PLAN", "1") != "0"
_RECORD = None  # set of module ids that called affine() during a recording forward


class StylePlan:
    def __init__(self, root):
        self.mods = [m for m in root.modules() if isinstance(m, AdaGN)]
        self._key, self._w, self._b = None, None, None
        self._used = None      # ids of the modules a training forward of this root calls (recorded by the first one)

    def _weights(self):
        key = (_wcache.generation(),) + tuple((m.emd.weight.data_ptr(), m.emd.weight._version, m.emd.bias._version)
                                              for m in self.mods)
        if key != self._key:
            self._w = torch.cat([m.emd.weight.detach() for m in self.mods], 0).contiguous()
            self._b = torch.cat([m.emd.bias.detach() for m in self.mods], 0).contiguous()
            self._key = key
        return self._w, self._b

    @contextlib.contextmanager
    def projected(self, style):
        """all AdaGN.affine(style) calls inside the block are served from one GEMM."""
        global _ACTIVE, _RECORD
        if not self.mods or style.dim() != 2 or any(m.style_dim != style.shape[1] for m in self.mods):
            yield
            return
        if torch.is_grad_enabled():
            if not TRAIN_BATCHED:
                yield
                return
            if self._used is None:          # first training forward: plain per-layer projections, recording who asks
                prev_r, _RECORD = _RECORD, set()
                try:
                    yield
                finally:
                    self._used, _RECORD = _RECORD, prev_r
                return
            mods = [m for m in self.mods if id(m) in self._used]
            if not mods:
                yield
                return
            w = torch.cat([m.emd.weight for m in mods], 0)
            b = torch.cat([m.emd.bias for m in mods], 0)
            e = torch.nn.functional.linear(style, w, b)
            parts = torch.split(e, [2 * m.n_channel for m in mods], dim=1)
            from .. import train_ops
            views = {id(m): train_ops.halves(p_) for m, p_ in zip(mods, parts)}   # chunk(2, 1) whose backward needs no cat
            prev, _ACTIVE = _ACTIVE, (style, views)
            try:
                yield
            finally:
                _ACTIVE = prev
            return
        w, b = self._weights()
        from .. import fused_ops
        e = fused_ops.linear_rows(style, w, b) if style.is_cuda and style.dtype == torch.float32 \
            else torch.nn.functional.linear(style, w, b)
        views, off = {}, 0
        for m in self.mods:
            c = m.n_channel
            views[id(m)] = (e[:, off:off + c], e[:, off + c:off + 2 * c])
            off += 2 * c
        prev, _ACTIVE = _ACTIVE, (style, views)
        try:
            yield
        finally:
            _ACTIVE = prev


class AdaGN(nn.Module):
    def __init__(self, ndim, cfg, n_channel):
        super().__init__()
        style_dim = cfg.latent_pts.style_dim
        self.ndim = ndim
        self.n_channel = n_channel
        self.style_dim = style_dim
        self.out_dim = n_channel * 2
        self.norm = nn.GroupNorm(8, n_channel)
        self.emd = dense(style_dim, n_channel * 2, init_scale=cfg.latent_pts.ada_mlp_init_scale)
        with torch.no_grad():  # through the parameter, not .data: the version counter must see it (_wcache.py)
            self.emd.bias[:n_channel] = 1
            self.emd.bias[n_channel:] = 0

    def __repr__(self):
        return f"AdaGN(GN(8, {self.n_channel}), Linear({self.style_dim}, {self.out_dim}))"

    def affine(self, style):
        """(factor, bias) each [B, C] -- the per-(batch, channel) scalars fused kernels consume."""
        assert style.dim() == 2, f"style must be [B, D], got {tuple(style.shape)}"
        if _ACTIVE is not None and _ACTIVE[0] is style:
            hit = _ACTIVE[1].get(id(self))
            if hit is not None:
                return hit
        if _RECORD is not None:
            _RECORD.add(id(self))
        return self.emd(style).chunk(2, 1)

    def forward(self, image, style):
        expect = {3: 5, 2: 4, 1: 3}.get(self.ndim)
        if expect is None:
            raise NotImplementedError
        assert image.dim() == expect, f"AdaGN(ndim={self.ndim}) expects a {expect}-D input"
        factor, bias = self.affine(style)
        shape = (image.shape[0], -1) + (1,) * (image.dim() - 2)
        return self.norm(image) * factor.reshape(shape) + bias.reshape(shape)
