"""gather / furthest_point_sample / logits_mask -- mirrors
third_party/pvcnn/functional/sampling.py:11-100."""
import numpy as np
import torch
from . import backend as _bk
from ._indexed import indexed_op

__all__ = ["gather", "furthest_point_sample", "logits_mask"]


gather = indexed_op("gather_features")  # features f32[B,C,N], indices int[B,M] -> f32[B,C,M]


def furthest_point_sample(coords, num_samples, normals=None):
    """coords f32[B,3,N] -> coordinates of the M furthest-point samples f32[B,3,M]
    (sampling.py:39-54: only gathered coordinates leave this function, never the indices)."""
    assert len(coords.shape) == 3 and coords.shape[1] == 3, \
        f'expect input as B,3,N; get: {coords.shape}'
    if normals is None:
        from .. import geometry
        hit = geometry.lookup_fps(coords, num_samples)  # prefetched on the side stream (inference)
        if hit is not None:
            return hit
        key, hit = geometry.memo_get("fps", (coords,), int(num_samples))   # a second network on the same cloud (geometry.shared)
        if hit is not None:
            return hit
        src = coords
    coords = coords.contiguous()
    indices = _bk._backend.furthest_point_sampling(coords, num_samples)
    centers_coords = gather(coords, indices)
    if normals is not None:
        return centers_coords, gather(normals, indices)
    return geometry.memo_put(key, (src,), centers_coords)


def _fps_compute(coords, num_samples):
    indices = _bk._backend.furthest_point_sampling(coords, num_samples)
    return _bk._backend.gather_features_forward(coords, indices)


def logits_mask(coords, logits, num_points_per_object):
    """sampling.py:57-100: pick M points per object where logits[:,1] > logits[:,0]
    (host-side numpy choice, as in the reference)."""
    batch_size, _, num_points = coords.shape
    mask = torch.lt(logits[:, 0, :], logits[:, 1, :])
    num_candidates = torch.sum(mask, dim=-1, keepdim=True)
    masked_coords = coords * mask.view(batch_size, 1, num_points)
    masked_coords_mean = torch.sum(masked_coords, dim=-1) / torch.max(
        num_candidates, torch.ones_like(num_candidates)).float()
    selected_indices = torch.zeros((batch_size, num_points_per_object), device=coords.device,
                                   dtype=torch.int32)
    for i in range(batch_size):
        candidates = mask[i].nonzero().view(-1)
        k = candidates.numel()
        if k >= num_points_per_object:
            choices = np.random.choice(k, num_points_per_object, replace=False)
            selected_indices[i] = candidates[choices]
        elif k > 0:
            choices = np.concatenate([
                np.arange(k).repeat(num_points_per_object // k),
                np.random.choice(k, num_points_per_object % k, replace=False)])
            np.random.shuffle(choices)
            selected_indices[i] = candidates[choices]
    selected_coords = gather(masked_coords - masked_coords_mean.view(batch_size, -1, 1),
                             selected_indices)
    return selected_coords, masked_coords_mean, mask
