"""avg_voxelize -- mirrors third_party/pvcnn/functional/voxelization.py:10-46."""
import torch
from torch.autograd import Function
from torch.amp import custom_fwd, custom_bwd

from . import backend as _bk

__all__ = ["avg_voxelize", "voxelize_points"]


class AvgVoxelization(Function):
    """features f32[B,C,N], coords int[B,3,N] (voxel ids) -> f32[B,C,R,R,R] (mean per voxel)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, features, coords, resolution):
        features = features.contiguous()
        coords = coords.int()[:, :3].contiguous()
        b, c, _ = features.shape
        out, indices, counts = _bk._backend.avg_voxelize_forward(features, coords, resolution)
        ctx.save_for_backward(indices, counts)
        return out.view(b, c, resolution, resolution, resolution)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        b, c = grad_output.shape[:2]
        indices, counts = ctx.saved_tensors
        grad_features = _bk._backend.avg_voxelize_backward(
            grad_output.contiguous().view(b, c, -1), indices, counts)
        return grad_features, None, None


avg_voxelize = AvgVoxelization.apply


class _VoxelizePoints(Function):
    """Fused Voxelization.forward (models/pvcnn2_ada.py:173-188): raw float coords in,
    (voxel grid, norm_coords) out, two launches.  Gradient flows to `features` only
    (coords are detached in the reference, :176)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, features, coords, resolution, normalize, eps):
        features = features.contiguous()
        coords = coords[:, :3].contiguous()
        b, c, _ = features.shape
        out, norm, indices, counts = _bk._backend.voxelize_points_forward(
            features, coords, resolution, normalize, eps)
        ctx.save_for_backward(indices, counts)
        ctx.mark_non_differentiable(norm, counts)
        return out.view(b, c, resolution, resolution, resolution), norm, counts

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_output, _grad_norm, _grad_counts=None):
        b, c = grad_output.shape[:2]
        indices, counts = ctx.saved_tensors
        grad_features = _bk._backend.avg_voxelize_backward(
            grad_output.contiguous().view(b, c, -1), indices, counts)
        return grad_features, None, None, None, None


VOXEL_COUNTS_TAG = "_lion_voxel_counts"


def voxelize_points(features, coords, resolution, normalize=True, eps=0.0):
    grid, norm, counts = _VoxelizePoints.apply(features, coords.detach(), int(resolution), bool(normalize),
                                               float(eps))
    # the per-voxel point counts ride on the grid tensor: the convolution that reads it (conv_ops.conv3d_module, training)
    # derives its empty tiles from them, as the fused inference branch does from Voxelization(return_counts=True)
    setattr(grid, VOXEL_COUNTS_TAG, (counts, grid._version))
    return grid, norm
