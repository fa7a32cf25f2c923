"""trilinear_devoxelize(features f32[B,C,R,R,R], coords f32[B,3,N] in voxel units, resolution, is_training)
-> f32[B,C,N]; drop-in for third_party/pvcnn/functional/devoxelization.py:8-45.  The eval path keeps nothing for
backward (the kernel then skips the index / weight outputs)."""
from torch.autograd import Function

from . import backend as _bk

__all__ = ["trilinear_devoxelize"]


class _Devoxelize(Function):
    @staticmethod
    def forward(ctx, grid, coords, resolution, is_training=True):
        flat = grid.contiguous().flatten(2)
        out, corner_idx, corner_w = _bk._backend.trilinear_devoxelize_forward(
            resolution, is_training, coords[:, :3].contiguous(), flat)
        ctx.grid_shape = tuple(grid.shape)
        if is_training:
            ctx.save_for_backward(corner_idx, corner_w)
        return out

    @staticmethod
    def backward(ctx, grad):
        corner_idx, corner_w = ctx.saved_tensors
        r = ctx.grid_shape[-1]
        g = _bk._backend.trilinear_devoxelize_backward(grad.contiguous(), corner_idx, corner_w, r)
        return (g.reshape(ctx.grid_shape),) + (None,) * 3


def trilinear_devoxelize(features, coords, resolution, is_training=True):
    return _Devoxelize.apply(features, coords, resolution, is_training)
