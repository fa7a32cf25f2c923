"""Approximate earth mover's distance -- replaces the reference's JIT-built ``emd_ext`` module
(third_party/PyTorchEMD/cuda/emd.cpp:23-27) and mirrors emd.py:6-51 / emd_nograd.py:6-44."""
import torch
from torch.amp import custom_fwd, custom_bwd

from . import _lib

__all__ = ["emd_ext", "emd_cuda", "earth_mover_distance", "earth_mover_distance_nograd",
           "EarthMoverDistanceFunction", "EarthMoverDistanceFunctionNoGrad"]


def _ws(b, n, m, dev):
    nbytes = _lib.load().lion_emd_workspace_bytes(b, n, m)
    return torch.empty((nbytes,), device=dev, dtype=torch.uint8), nbytes


class _EmdModule:
    """approxmatch_forward / matchcost_forward / matchcost_backward, xyz point-major [B,N,3]."""

    @staticmethod
    def approxmatch_forward(xyz1, xyz2):
        _lib.require_cuda(xyz1, xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        match = torch.empty((b, m, n), device=xyz1.device, dtype=torch.float32)
        ws, nbytes = _ws(b, n, m, xyz1.device)
        _lib.check(_lib.load().lion_emd_approxmatch(
            _lib.ptr(xyz1), _lib.ptr(xyz2), b, n, m, _lib.ptr(match), _lib.ptr(ws), nbytes,
            _lib.stream_ptr(xyz1.device)), "approxmatch_forward")
        return match

    @staticmethod
    def matchcost_forward(xyz1, xyz2, match):
        _lib.require_cuda(xyz1, xyz2, match)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        cost = torch.empty((b,), device=xyz1.device, dtype=torch.float32)
        ws, nbytes = _ws(b, n, m, xyz1.device)
        _lib.check(_lib.load().lion_emd_matchcost(
            _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(match), b, n, m, _lib.ptr(cost),
            _lib.ptr(ws), nbytes, _lib.stream_ptr(xyz1.device)), "matchcost_forward")
        return cost

    @staticmethod
    def matchcost_backward(grad_cost, xyz1, xyz2, match):
        _lib.require_cuda(grad_cost, xyz1, xyz2, match)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        g1 = torch.empty((b, n, 3), device=xyz1.device, dtype=torch.float32)
        g2 = torch.empty((b, m, 3), device=xyz1.device, dtype=torch.float32)
        _lib.check(_lib.load().lion_emd_matchcost_backward(
            _lib.ptr(grad_cost), _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(match), b, n, m,
            _lib.ptr(g1), _lib.ptr(g2), _lib.stream_ptr(xyz1.device)), "matchcost_backward")
        return [g1, g2]


emd_ext = emd_cuda = _EmdModule()


class EarthMoverDistanceFunction(torch.autograd.Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, xyz1, xyz2):
        xyz1 = xyz1.contiguous()
        xyz2 = xyz2.contiguous()
        assert xyz1.is_cuda and xyz2.is_cuda, "Only support cuda currently."
        match = emd_ext.approxmatch_forward(xyz1, xyz2)
        cost = emd_ext.matchcost_forward(xyz1, xyz2, match)
        ctx.save_for_backward(xyz1, xyz2, match)
        return cost

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_cost):
        xyz1, xyz2, match = ctx.saved_tensors
        g1, g2 = emd_ext.matchcost_backward(grad_cost.contiguous(), xyz1, xyz2, match)
        return g1, g2


class EarthMoverDistanceFunctionNoGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1 = xyz1.contiguous()
        xyz2 = xyz2.contiguous()
        assert xyz1.is_cuda and xyz2.is_cuda, "Only support cuda currently."
        # no gradient needed: the fused path never materialises the [B,N,M] match matrix
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        cost = torch.empty((b,), device=xyz1.device, dtype=torch.float32)
        ws, nbytes = _ws(b, n, m, xyz1.device)
        _lib.check(_lib.load().lion_emd_cost(_lib.ptr(xyz1), _lib.ptr(xyz2), b, n, m, _lib.ptr(cost), _lib.ptr(ws),
                                             nbytes, _lib.stream_ptr(xyz1.device)), "emd_cost")
        return cost


def _prep(xyz1, xyz2, transpose):
    if xyz1.dim() == 2:
        xyz1 = xyz1.unsqueeze(0)
    if xyz2.dim() == 2:
        xyz2 = xyz2.unsqueeze(0)
    if transpose:
        xyz1 = xyz1.transpose(1, 2)
        xyz2 = xyz2.transpose(1, 2)
    assert xyz1.shape[-1] == 3, f'require it to be B,N,3; get: {xyz1.shape}'
    return xyz1, xyz2


def earth_mover_distance(xyz1, xyz2, transpose=True):
    """(b,3,n)|(b,n,3) clouds -> per-pair cost / n  (emd.py:31-51)."""
    xyz1, xyz2 = _prep(xyz1, xyz2, transpose)
    return EarthMoverDistanceFunction.apply(xyz1, xyz2) / float(xyz1.shape[1])


def earth_mover_distance_nograd(xyz1, xyz2, transpose=True):
    """emd_nograd.py:19-44."""
    xyz1, xyz2 = _prep(xyz1, xyz2, transpose)
    return EarthMoverDistanceFunctionNoGrad.apply(xyz1, xyz2) / float(xyz1.shape[1])
