"""ball_query -- mirrors third_party/pvcnn/functional/ball_query.py:8-20."""
from . import backend as _bk

__all__ = ["ball_query"]


def ball_query(centers_coords, points_coords, radius, num_neighbors):
    """centers f32[B,3,M], points f32[B,3,N] -> neighbour indices int32[B,M,U]."""
    centers_coords = centers_coords[:, :3].contiguous()
    points_coords = points_coords[:, :3].contiguous()
    return _bk._backend.ball_query(centers_coords, points_coords, radius, num_neighbors)
