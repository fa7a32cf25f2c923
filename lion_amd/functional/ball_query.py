"""ball_query -- mirrors third_party/pvcnn/functional/ball_query.py:8-20."""
from . import backend as _bk

__all__ = ["ball_query"]


def ball_query(centers_coords, points_coords, radius, num_neighbors):
    """centers f32[B,3,M], points f32[B,3,N] -> neighbour indices int32[B,M,U]."""
    from .. import geometry
    hit = geometry.lookup_ball_query(centers_coords, points_coords, radius, num_neighbors)
    if hit is not None:  # prefetched on the side stream (inference)
        return hit
    key, hit = geometry.memo_get("bq", (centers_coords, points_coords), float(radius), int(num_neighbors))
    if hit is not None:  # a second network on the same cloud (geometry.shared)
        return hit
    src = (centers_coords, points_coords)
    centers_coords = centers_coords[:, :3].contiguous()
    points_coords = points_coords[:, :3].contiguous()
    return geometry.memo_put(key, src, _bk._backend.ball_query(centers_coords, points_coords, radius, num_neighbors))


def _ball_query_compute(centers_coords, points_coords, radius, num_neighbors):
    return _bk._backend.ball_query(centers_coords, points_coords, radius, num_neighbors)
