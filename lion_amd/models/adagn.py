"""Adaptive group normalisation (reference models/adagn.py:19-65):
    y = GroupNorm8(x) * (W s + b)[:C] + (W s + b)[C:]
Parameters: ``norm.{weight,bias}`` (GroupNorm(8, C)), ``emd.{weight,bias}`` (Linear(style, 2C))."""
import torch.nn as nn

from .dense import dense


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
        self.emd.bias.data[:n_channel] = 1
        self.emd.bias.data[n_channel:] = 0

    def __repr__(self):
        return f"AdaGN(GN(8, {self.n_channel}), Linear({self.style_dim}, {self.out_dim}))"

    def affine(self, style):
        """(factor, bias) each [B, C] -- the per-(batch, channel) scalars fused kernels consume."""
        assert style.dim() == 2, f"style must be [B, D], got {tuple(style.shape)}"
        return self.emd(style).chunk(2, 1)

    def forward(self, image, style):
        expect = {3: 5, 2: 4, 1: 3}.get(self.ndim)
        if expect is None:
            raise NotImplementedError
        assert image.dim() == expect, f"AdaGN(ndim={self.ndim}) expects a {expect}-D input"
        factor, bias = self.affine(style)
        shape = (image.shape[0], -1) + (1,) * (image.dim() - 2)
        return self.norm(image) * factor.reshape(shape) + bias.reshape(shape)
