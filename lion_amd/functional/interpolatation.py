"""nearest_neighbor_interpolate -- mirrors third_party/pvcnn/functional/interpolatation.py:11-41
(the module keeps the reference's spelling of its file name)."""
import torch
from torch.autograd import Function
from torch.amp import custom_fwd, custom_bwd

from . import backend as _bk

__all__ = ["nearest_neighbor_interpolate"]


class NeighborInterpolation(Function):
    """points f32[B,3,N], centers f32[B,3,M], centers_features f32[B,C,M] -> f32[B,C,N]."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, points_coords, centers_coords, centers_features):
        centers_coords = centers_coords[:, :3].contiguous()
        points_coords = points_coords[:, :3].contiguous()
        centers_features = centers_features.contiguous()
        points_features, indices, weights = \
            _bk._backend.three_nearest_neighbors_interpolate_forward(
                points_coords, centers_coords, centers_features)
        ctx.save_for_backward(indices, weights)
        ctx.num_centers = centers_coords.size(-1)
        return points_features

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        indices, weights = ctx.saved_tensors
        grad_centers_features = _bk._backend.three_nearest_neighbors_interpolate_backward(
            grad_output.contiguous(), indices, weights, ctx.num_centers)
        return None, None, grad_centers_features


nearest_neighbor_interpolate = NeighborInterpolation.apply
