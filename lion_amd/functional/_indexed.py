"""One autograd Function for the two "pick columns by index" operators of the PVCNN API -- grouping
(indices [B,M,U] -> [B,C,M,U]) and gather (indices [B,M] -> [B,C,M]).  Both are a gather forward whose
gradient is the matching scatter-add into [B,C,N]; only the backend entry points differ, so they are looked up
by name (reference: third_party/pvcnn/functional/grouping.py:9-33, sampling.py:11-36)."""
from torch.autograd import Function

from . import backend as _bk


class IndexedColumns(Function):
    @staticmethod
    def forward(ctx, op, features, indices):
        src = features.contiguous()
        sel = indices.int().contiguous()
        ctx.op, ctx.width = op, src.shape[-1]
        ctx.save_for_backward(sel)
        return getattr(_bk._backend, op + "_forward")(src, sel)

    @staticmethod
    def backward(ctx, grad):
        (sel,) = ctx.saved_tensors
        scatter = getattr(_bk._backend, ctx.op + "_backward")
        return None, scatter(grad.contiguous(), sel, ctx.width), None


def indexed_op(op):
    def run(features, indices):
        return IndexedColumns.apply(op, features, indices)
    run.__name__ = op
    return run
