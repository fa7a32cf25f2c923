"""grouping -- mirrors third_party/pvcnn/functional/grouping.py:9-33."""
from torch.autograd import Function

from . import backend as _bk

__all__ = ["grouping"]


class Grouping(Function):
    """features f32[B,C,N], indices int32[B,M,U] -> f32[B,C,M,U]."""

    @staticmethod
    def forward(ctx, features, indices):
        features = features.contiguous()
        indices = indices.contiguous()
        ctx.save_for_backward(indices)
        ctx.num_points = features.size(-1)
        return _bk._backend.grouping_forward(features, indices)

    @staticmethod
    def backward(ctx, grad_output):
        indices, = ctx.saved_tensors
        grad_features = _bk._backend.grouping_backward(grad_output.contiguous(), indices,
                                                       ctx.num_points)
        return grad_features, None


grouping = Grouping.apply
