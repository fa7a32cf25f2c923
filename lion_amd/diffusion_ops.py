"""Fused denoiser-step updates (D1) over the C ABI: lion_ddim_update / lion_ddpm_update."""
import torch

from . import _lib

__all__ = ["ddim_update", "ddpm_update"]


def ddim_update(x, eps, z, s, c, sigma, out=None):
    """out = x*s + (c*eps + sigma*z)   (utils/diffusion_pvd.py:451-467); z may be None if sigma==0."""
    _lib.require_cuda(x, eps, z)
    out = torch.empty_like(x) if out is None else out
    _lib.check(_lib.load().lion_ddim_update(
        _lib.ptr(x), _lib.ptr(eps), _lib.ptr(z), x.numel(), float(s), float(c), float(sigma),
        _lib.ptr(out), _lib.stream_ptr(x.device)), "ddim_update")
    return out


def ddpm_update(x, eps, z, t_is_zero, k_outer, k_a, k_b, scale, temp, out=None):
    """DDPM ancestral step (utils/diffusion_pvd.py:283-296, :475-486)."""
    _lib.require_cuda(x, eps, z)
    out = torch.empty_like(x) if out is None else out
    _lib.check(_lib.load().lion_ddpm_update(
        _lib.ptr(x), _lib.ptr(eps), _lib.ptr(z), x.numel(), int(bool(t_is_zero)), float(k_outer),
        float(k_a), float(k_b), float(scale), float(temp), _lib.ptr(out),
        _lib.stream_ptr(x.device)), "ddpm_update")
    return out
