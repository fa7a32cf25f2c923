"""Timestep embeddings of the global prior -- interface of the reference's models/utils.py:12-52
(`init_temb_fun`, `PositionalEmbedding`, `RandomFourierEmbedding`, `mask_inactive_variables`).

Both embeddings are [sin(a), cos(a)] of an angle table a = t (x) f; they differ in where the frequency row f
comes from (a fixed geometric ladder, or a frozen random parameter `w`, which is part of the state dict)."""
import math

import torch
import torch.nn as nn


def mask_inactive_variables(x, is_active):
    return x * is_active


def _sincos(angles):
    return torch.cat((angles.sin(), angles.cos()), dim=1)


class PositionalEmbedding(nn.Module):
    """f_i = 10000^(-i / (half - 1)), i < half = embedding_dim // 2; t is multiplied by `scale` first."""

    def __init__(self, embedding_dim, scale):
        super().__init__()
        self.embedding_dim, self.scale = embedding_dim, scale
        # per-device copy of the ladder: built on the host exactly like the reference, but only once -- a
        # host->device copy inside forward() cannot be captured into a hipGraph
        self._ladder = {}

    def _frequencies(self, device):
        f = self._ladder.get(device)
        if f is None:
            half = self.embedding_dim // 2
            f = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1))).to(device)
            self._ladder[device] = f
        return f

    def forward(self, timesteps):
        if timesteps.dim() != 1:
            raise AssertionError("timesteps must be 1-D")
        return _sincos((timesteps * self.scale).unsqueeze(1) * self._frequencies(timesteps.device).unsqueeze(0))


class RandomFourierEmbedding(nn.Module):
    """f = 2 pi w with w ~ N(0, scale^2) drawn once and frozen."""

    def __init__(self, embedding_dim, scale):
        super().__init__()
        self.w = nn.Parameter(torch.randn(size=(1, embedding_dim // 2)) * scale, requires_grad=False)

    def forward(self, timesteps):
        return _sincos(torch.mm(timesteps.unsqueeze(1), self.w * 2 * 3.14159265359))


_EMBEDDINGS = {"positional": PositionalEmbedding, "fourier": RandomFourierEmbedding}


def init_temb_fun(embedding_type, embedding_scale, embedding_dim):
    try:
        return _EMBEDDINGS[embedding_type](embedding_dim, embedding_scale)
    except KeyError:
        raise NotImplementedError(embedding_type) from None
