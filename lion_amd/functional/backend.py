"""Drop-in replacement for ``third_party.pvcnn.functional.backend`` (reference backend.py:8-27).

The reference JIT-builds a pybind11/ATen CUDA module ``_pvcnn_backend`` and exposes it as
``_backend``.  Here ``_backend`` is an object with exactly the same 12 callables
(bindings.cpp:10-37, signatures in the per-op ``*.hpp`` files) implemented on top of the C ABI of
``liblion_hip.so``.  This shim owns what the ATen side owned in the reference: dtype / device /
contiguity checks, output allocation, current-stream lookup, error -> exception.

Two extra callables (not in the reference) expose fused paths: ``voxelize_points_forward``
(Voxelization.forward, pvcnn2_ada.py:173-188, in two launches) -- used by ``lion_amd.models``.
"""
from __future__ import annotations

import torch

from .. import _lib

__all__ = ["_backend"]


def _f32(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float32 tensor")  # CHECK_IS_FLOAT


def _i32(t, name):
    if t.dtype != torch.int32:
        raise RuntimeError(f"{name} must be an int32 tensor")  # CHECK_IS_INT


class _HipBackend:
    """The 12 operator entry points of the reference's ``_pvcnn_backend`` module."""

    def __init__(self):
        self._l = None

    @property
    def lib(self):
        if self._l is None:
            self._l = _lib.load()
        return self._l

    # -- sampling/sampling.cpp:8-24 ------------------------------------------------------------
    def gather_features_forward(self, features, indices):
        _lib.require_cuda(features, indices); _f32(features, "features"); _i32(indices, "indices")
        b, c, n = features.shape
        m = indices.shape[1]
        out = torch.empty((b, c, m), device=features.device, dtype=torch.float32)
        _lib.check(self.lib.lion_gather_features_forward(
            _lib.ptr(features), _lib.ptr(indices), b, c, n, m, _lib.ptr(out),
            _lib.stream_ptr(features.device)), "gather_features_forward")
        return out

    # -- sampling/sampling.cpp:26-41 -----------------------------------------------------------
    def gather_features_backward(self, grad_y, indices, n):
        _lib.require_cuda(grad_y, indices); _f32(grad_y, "grad_y"); _i32(indices, "indices")
        b, c, m = grad_y.shape
        gx = torch.empty((b, c, n), device=grad_y.device, dtype=torch.float32)
        _lib.check(self.lib.lion_gather_features_backward(
            _lib.ptr(grad_y), _lib.ptr(indices), b, c, n, m, _lib.ptr(gx),
            _lib.stream_ptr(grad_y.device)), "gather_features_backward")
        return gx

    # -- sampling/sampling.cpp:43-58 -----------------------------------------------------------
    def furthest_point_sampling(self, coords, num_samples):
        _lib.require_cuda(coords); _f32(coords, "coords")
        b, _, n = coords.shape
        idx = torch.empty((b, num_samples), device=coords.device, dtype=torch.int32)
        _lib.check(self.lib.lion_furthest_point_sampling(
            _lib.ptr(coords), b, n, num_samples, _lib.ptr(idx), _lib.stream_ptr(coords.device)),
            "furthest_point_sampling")
        return idx

    # -- ball_query/ball_query.cpp:7-33 --------------------------------------------------------
    def ball_query(self, centers_coords, points_coords, radius, num_neighbors):
        _lib.require_cuda(centers_coords, points_coords)
        _f32(centers_coords, "centers_coords"); _f32(points_coords, "points_coords")
        b, _, m = centers_coords.shape
        n = points_coords.shape[2]
        idx = torch.empty((b, m, num_neighbors), device=centers_coords.device, dtype=torch.int32)
        _lib.check(self.lib.lion_ball_query(
            _lib.ptr(centers_coords), _lib.ptr(points_coords), b, m, n, float(radius),
            int(num_neighbors), _lib.ptr(idx), _lib.stream_ptr(idx.device)), "ball_query")
        return idx

    # -- grouping/grouping.cpp ----------------------------------------------------------------
    def grouping_forward(self, features, indices):
        _lib.require_cuda(features, indices); _f32(features, "features"); _i32(indices, "indices")
        b, c, n = features.shape
        _, m, u = indices.shape
        out = torch.empty((b, c, m, u), device=features.device, dtype=torch.float32)
        _lib.check(self.lib.lion_grouping_forward(
            _lib.ptr(features), _lib.ptr(indices), b, c, n, m, u, _lib.ptr(out),
            _lib.stream_ptr(out.device)), "grouping_forward")
        return out

    def _scatter_csr(self, grad_y, indices, weights, S, E, bins, what):
        """gx[b,c,bin] = sum_{e: idx[b,e] = bin} w[b,e] * gy[b,c, e mod S] on csrc/scatter_csr.hip (no float atomics,
        deterministic); None when the shape is outside its range (the callers fall back to the LDS-atomic kernels)."""
        b, c = grad_y.shape[:2]
        wsb = self.lib.lion_scatter_csr_workspace_bytes(b, E, bins)
        if wsb == 0 or ((bins + 7) // 8) * 4 > 60 * 1024 or S * 4 > 128 * 1024:
            return None
        gy, idx = grad_y.contiguous(), indices.contiguous()
        w = weights.contiguous() if weights is not None else None
        ws = torch.empty((wsb,), device=gy.device, dtype=torch.uint8)
        gx = torch.empty((b, c, bins), device=gy.device, dtype=torch.float32)
        _lib.check(self.lib.lion_scatter_csr(_lib.ptr(gy), _lib.ptr(idx), _lib.ptr(w), b, c, S, E, bins, _lib.ptr(ws), wsb,
                                             _lib.ptr(gx), _lib.stream_ptr(gx.device)), what)
        return gx

    def grouping_backward(self, grad_y, indices, n):
        _lib.require_cuda(grad_y, indices); _f32(grad_y, "grad_y"); _i32(indices, "indices")
        b, c, m, u = grad_y.shape
        gx = self._scatter_csr(grad_y, indices, None, m * u, m * u, int(n), "grouping_backward (csr)")
        if gx is not None:
            return gx
        gx = torch.empty((b, c, n), device=grad_y.device, dtype=torch.float32)
        _lib.check(self.lib.lion_grouping_backward(
            _lib.ptr(grad_y), _lib.ptr(indices), b, c, n, m, u, _lib.ptr(gx),
            _lib.stream_ptr(gx.device)), "grouping_backward")
        return gx

    # -- interpolate/neighbor_interpolate.cpp --------------------------------------------------
    def three_nearest_neighbors_interpolate_forward(self, points_coords, centers_coords,
                                                    centers_features):
        _lib.require_cuda(points_coords, centers_coords, centers_features)
        _f32(points_coords, "points_coords"); _f32(centers_coords, "centers_coords")
        _f32(centers_features, "centers_features")
        b, c, m = centers_features.shape
        n = points_coords.shape[2]
        dev = points_coords.device
        out = torch.empty((b, c, n), device=dev, dtype=torch.float32)
        idx = torch.empty((b, 3, n), device=dev, dtype=torch.int32)
        wgt = torch.empty((b, 3, n), device=dev, dtype=torch.float32)
        _lib.check(self.lib.lion_three_nn_interpolate_forward(
            _lib.ptr(points_coords), _lib.ptr(centers_coords), _lib.ptr(centers_features), b, c, n,
            m, _lib.ptr(out), _lib.ptr(idx), _lib.ptr(wgt), _lib.stream_ptr(dev)),
            "three_nearest_neighbors_interpolate_forward")
        return out, idx, wgt

    def three_nearest_neighbors_interpolate_cat_forward(self, points_coords, centers_coords, centers_features, temb_rows,
                                                        ld_t, skip):
        """[interpolate(cat(centers_features, temb)) ; skip] in one pass (lion_three_nn_interpolate_cat_forward): temb_rows
        f32[B-strided, C2] with row stride ld_t floats (0 = one row for the whole batch) or None, skip f32[B,C3,N] or None
        -> (out f32[B,C1+C2+C3,N], idx i32[B,3,N], wgt f32[B,3,N])."""
        _lib.require_cuda(points_coords, centers_coords, centers_features, skip)
        _f32(points_coords, "points_coords"); _f32(centers_coords, "centers_coords")
        _f32(centers_features, "centers_features")
        b, c1, m = centers_features.shape
        n = points_coords.shape[2]
        c2 = 0 if temb_rows is None else int(temb_rows.shape[1])
        c3 = 0 if skip is None else int(skip.shape[1])
        dev = points_coords.device
        out = torch.empty((b, c1 + c2 + c3, n), device=dev, dtype=torch.float32)
        idx = torch.empty((b, 3, n), device=dev, dtype=torch.int32)
        wgt = torch.empty((b, 3, n), device=dev, dtype=torch.float32)
        _lib.check(self.lib.lion_three_nn_interpolate_cat_forward(
            _lib.ptr(points_coords), _lib.ptr(centers_coords), _lib.ptr(centers_features), _lib.ptr(temb_rows), int(ld_t),
            _lib.ptr(skip), b, c1, c2, c3, n, m, _lib.ptr(out), _lib.ptr(idx), _lib.ptr(wgt), _lib.stream_ptr(dev)),
            "three_nearest_neighbors_interpolate_cat_forward")
        return out, idx, wgt

    def three_nearest_neighbors_interpolate_backward(self, grad_y, indices, weights, m):
        _lib.require_cuda(grad_y, indices, weights)
        _f32(grad_y, "grad_y"); _i32(indices, "indices"); _f32(weights, "weights")
        b, c, n = grad_y.shape
        gx = self._scatter_csr(grad_y, indices, weights, n, 3 * n, int(m), "three_nn_interpolate_backward (csr)")
        if gx is not None:
            return gx
        gx = torch.empty((b, c, m), device=grad_y.device, dtype=torch.float32)
        _lib.check(self.lib.lion_three_nn_interpolate_backward(
            _lib.ptr(grad_y), _lib.ptr(indices), _lib.ptr(weights), b, c, n, m, _lib.ptr(gx),
            _lib.stream_ptr(gx.device)), "three_nearest_neighbors_interpolate_backward")
        return gx

    # -- interpolate/trilinear_devox.cpp:18-55 -------------------------------------------------
    def trilinear_devoxelize_forward(self, r, is_training, coords, features):
        _lib.require_cuda(coords, features); _f32(coords, "coords"); _f32(features, "features")
        b, c, r3 = features.shape
        if r3 != r * r * r:
            raise RuntimeError("features.size(2) must be r**3")
        n = coords.shape[2]
        dev = features.device
        outs = torch.empty((b, c, n), device=dev, dtype=torch.float32)
        if is_training:
            inds = torch.empty((b, 8, n), device=dev, dtype=torch.int32)
            wgts = torch.empty((b, 8, n), device=dev, dtype=torch.float32)
        else:  # the reference returns [1]-shaped placeholders (trilinear_devox.cpp:45-54)
            inds = torch.zeros((1,), device=dev, dtype=torch.int32)
            wgts = torch.zeros((1,), device=dev, dtype=torch.float32)
        _lib.check(self.lib.lion_trilinear_devoxelize_forward(
            _lib.ptr(coords), _lib.ptr(features), b, c, n, int(r), int(bool(is_training)),
            _lib.ptr(outs), _lib.ptr(inds) if is_training else None,
            _lib.ptr(wgts) if is_training else None, _lib.stream_ptr(dev)),
            "trilinear_devoxelize_forward")
        return outs, inds, wgts

    # -- interpolate/trilinear_devox.cpp:67-95 -------------------------------------------------
    def trilinear_devoxelize_backward(self, grad_y, indices, weights, r):
        _lib.require_cuda(grad_y, indices, weights)
        _f32(grad_y, "grad_y"); _i32(indices, "indices"); _f32(weights, "weights")
        b, c, n = grad_y.shape
        r3 = r * r * r
        # (the atomics-free path of csrc/scatter_csr.hip is correct here too -- tests/test_scatter_csr_gpu.py -- but slower:
        # 32768 mostly empty bins per sample, 373 us against 198 at (64, 2048, 32), B = 32; K8 and K12-grad use it)
        gx = torch.empty((b, c, r3), device=grad_y.device, dtype=torch.float32)
        _lib.check(self.lib.lion_trilinear_devoxelize_backward(
            _lib.ptr(grad_y), _lib.ptr(indices), _lib.ptr(weights), b, c, n, r3, _lib.ptr(gx),
            _lib.stream_ptr(gx.device)), "trilinear_devoxelize_backward")
        return gx

    # -- voxelization/vox.cpp:17-43 -----------------------------------------------------------
    def avg_voxelize_forward(self, features, coords, resolution):
        _lib.require_cuda(features, coords); _f32(features, "features"); _i32(coords, "coords")
        b, c, n = features.shape
        r = int(resolution)
        r3 = r * r * r
        dev = features.device
        out = torch.empty((b, c, r3), device=dev, dtype=torch.float32)
        ind = torch.empty((b, n), device=dev, dtype=torch.int32)
        cnt = torch.empty((b, r3), device=dev, dtype=torch.int32)
        wsb = self.lib.lion_avg_voxelize_workspace_bytes(b, c, n, r)
        ws = torch.empty((wsb,), device=dev, dtype=torch.uint8)
        _lib.check(self.lib.lion_avg_voxelize_forward(
            _lib.ptr(features), _lib.ptr(coords), b, c, n, r, _lib.ptr(out), _lib.ptr(ind),
            _lib.ptr(cnt), _lib.ptr(ws), wsb, _lib.stream_ptr(dev)), "avg_voxelize_forward")
        return out, ind, cnt

    # -- voxelization/vox.cpp:54-79 -----------------------------------------------------------
    def avg_voxelize_backward(self, grad_y, indices, cnt):
        _lib.require_cuda(grad_y, indices, cnt)
        _f32(grad_y, "grad_y"); _i32(indices, "indices"); _i32(cnt, "cnt")
        b, c, s = grad_y.shape
        n = indices.shape[1]
        gx = torch.empty((b, c, n), device=grad_y.device, dtype=torch.float32)
        _lib.check(self.lib.lion_avg_voxelize_backward(
            _lib.ptr(grad_y), _lib.ptr(indices), _lib.ptr(cnt), b, c, n, s, _lib.ptr(gx),
            _lib.stream_ptr(gx.device)), "avg_voxelize_backward")
        return gx

    # -- fused: Voxelization.forward (pvcnn2_ada.py:173-188), not part of the reference module ---
    def voxelize_points_forward(self, features, coords, resolution, normalize=True, eps=0.0):
        """coords f32[B,3,N] raw -> (out f32[B,C,r^3] | None, norm_coords, ind, cnt)."""
        _lib.require_cuda(features, coords); _f32(coords, "coords")
        b, _, n = coords.shape
        r = int(resolution)
        r3 = r * r * r
        dev = coords.device
        c = 0
        out = None
        if features is not None:
            _f32(features, "features")
            c = features.shape[1]
            out = torch.empty((b, c, r3), device=dev, dtype=torch.float32)
        norm = torch.empty((b, 3, n), device=dev, dtype=torch.float32)
        ind = torch.empty((b, n), device=dev, dtype=torch.int32)
        cnt = torch.empty((b, r3), device=dev, dtype=torch.int32)
        wsb = self.lib.lion_avg_voxelize_workspace_bytes(b, max(c, 1), n, r)
        ws = torch.empty((wsb,), device=dev, dtype=torch.uint8)
        _lib.check(self.lib.lion_voxelize_points_forward(
            _lib.ptr(features), _lib.ptr(coords), b, c, n, r, int(bool(normalize)), float(eps),
            _lib.ptr(out), _lib.ptr(norm), _lib.ptr(ind), _lib.ptr(cnt), _lib.ptr(ws), wsb,
            _lib.stream_ptr(dev)), "voxelize_points_forward")
        return out, norm, ind, cnt

    # -- the same in two steps: index plan of (coords, r) once, mean-pool per feature tensor --------------------
    def voxel_index(self, coords, resolution, normalize=True, eps=0.0):
        """coords f32[B,3,N] raw -> plan dict {norm, ind, cnt, ws, key} for voxel_scatter, or None when the shape is
        outside the plan kernels' range (callers then use voxelize_points_forward)."""
        _lib.require_cuda(coords); _f32(coords, "coords")
        b, _, n = coords.shape
        r = int(resolution)
        nbytes = self.lib.lion_voxel_plan_bytes(b, n, r)
        if nbytes == 0:
            return None
        dev = coords.device
        norm = torch.empty((b, 3, n), device=dev, dtype=torch.float32)
        ind = torch.empty((b, n), device=dev, dtype=torch.int32)
        cnt = torch.empty((b, r * r * r), device=dev, dtype=torch.int32)
        ws = torch.empty((nbytes,), device=dev, dtype=torch.uint8)
        _lib.check(self.lib.lion_voxel_index(_lib.ptr(coords), b, n, r, int(bool(normalize)), float(eps), _lib.ptr(norm),
                                             _lib.ptr(ind), _lib.ptr(cnt), _lib.ptr(ws), nbytes, _lib.stream_ptr(dev)),
                   "voxel_index")
        return {"norm": norm, "ind": ind, "cnt": cnt, "ws": ws, "shape": (b, n, r)}

    def voxel_scatter(self, features, plan, occ_m1=None):
        """features f32[B,C,N] -> f32[B,C,r^3] mean-pooled with plan's voxel assignment (bit-identical to
        voxelize_points_forward on the plan's coordinates).  occ_m1 (the margin-1 buffer of fused_ops.conv3d_occupancy):
        the grid's only reader is the sparse convolution popping it -- z-rows outside every occupied tile's halo are
        left unwritten (lion_voxel_scatter_read)."""
        _lib.require_cuda(features); _f32(features, "features")
        b, n, r = plan["shape"]
        if features.shape[0] != b or features.shape[2] != n:
            raise RuntimeError("voxel_scatter: features do not match the plan's cloud")
        c = features.shape[1]
        out = torch.empty((b, c, r * r * r), device=features.device, dtype=torch.float32)
        ws = plan["ws"]
        if occ_m1 is not None:
            _lib.check(self.lib.lion_voxel_scatter_read(_lib.ptr(features), _lib.ptr(ws), ws.numel(), b, c, n, r,
                                                        _lib.ptr(occ_m1), _lib.ptr(out), _lib.stream_ptr(features.device)),
                       "voxel_scatter_read")
            return out
        _lib.check(self.lib.lion_voxel_scatter(_lib.ptr(features), _lib.ptr(ws), ws.numel(), b, c, n, r, _lib.ptr(out),
                                               _lib.stream_ptr(features.device)), "voxel_scatter")
        return out


_backend = _HipBackend()
