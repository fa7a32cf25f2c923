"""trilinear_devoxelize -- mirrors third_party/pvcnn/functional/devoxelization.py:8-45."""
from torch.autograd import Function

from . import backend as _bk

__all__ = ["trilinear_devoxelize"]


class TrilinearDevoxelization(Function):
    """features f32[B,C,R,R,R], coords f32[B,3,N] in voxel units -> f32[B,C,N]."""

    @staticmethod
    def forward(ctx, features, coords, resolution, is_training=True):
        B, C = features.shape[:2]
        features = features.contiguous().view(B, C, -1)
        coords = coords[:, :3].contiguous()
        outs, inds, wgts = _bk._backend.trilinear_devoxelize_forward(
            resolution, is_training, coords, features)
        if is_training:
            ctx.save_for_backward(inds, wgts)
            ctx.r = resolution
        return outs

    @staticmethod
    def backward(ctx, grad_output):
        inds, wgts = ctx.saved_tensors
        grad_inputs = _bk._backend.trilinear_devoxelize_backward(
            grad_output.contiguous(), inds, wgts, ctx.r)
        return grad_inputs.view(grad_output.size(0), grad_output.size(1), ctx.r, ctx.r, ctx.r), \
            None, None, None


trilinear_devoxelize = TrilinearDevoxelization.apply
