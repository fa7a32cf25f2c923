"""grouping(features f32[B,C,N], indices int32[B,M,U]) -> f32[B,C,M,U]; drop-in for
third_party/pvcnn/functional/grouping.py.  Implementation: functional/_indexed.py."""
from ._indexed import indexed_op

__all__ = ["grouping"]

grouping = indexed_op("grouping")
