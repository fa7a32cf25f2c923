"""Timestep embeddings of the global prior (reference models/utils.py:12-52)."""
import math

import torch
import torch.nn as nn


def mask_inactive_variables(x, is_active):
    return x * is_active


class PositionalEmbedding(nn.Module):
    def __init__(self, embedding_dim, scale):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.scale = scale
        self._freq = {}  # device -> frequencies (built on the CPU exactly as the reference does, cached:
        #                  a host->device copy per call cannot be captured in a hipGraph)

    def forward(self, timesteps):
        assert timesteps.dim() == 1
        half = self.embedding_dim // 2
        freq = self._freq.get(timesteps.device)
        if freq is None:
            freq = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1))).to(timesteps.device)
            self._freq[timesteps.device] = freq
        ang = (timesteps * self.scale)[:, None] * freq[None, :]
        return torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)


class RandomFourierEmbedding(nn.Module):
    def __init__(self, embedding_dim, scale):
        super().__init__()
        self.w = nn.Parameter(torch.randn(size=(1, embedding_dim // 2)) * scale, requires_grad=False)

    def forward(self, timesteps):
        ang = torch.mm(timesteps[:, None], self.w * 2 * 3.14159265359)
        return torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)


def init_temb_fun(embedding_type, embedding_scale, embedding_dim):
    if embedding_type == "positional":
        return PositionalEmbedding(embedding_dim, embedding_scale)
    if embedding_type == "fourier":
        return RandomFourierEmbedding(embedding_dim, embedding_scale)
    raise NotImplementedError(embedding_type)
