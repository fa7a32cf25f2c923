"""CPU oracle for the LION / PVCNN hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product (``lion_amd``) never does: it fails loudly when the HIP
extension is missing instead of falling back to anything here.

``oracle.lib`` wraps ``oracle/liboracle.so`` (built from ``lion_oracle.c`` by
``oracle/Makefile``); every wrapper takes / returns C-contiguous numpy arrays.
``oracle.TorchBackend`` exposes the same 12 callables as the reference's pybind module
(third_party/pvcnn/functional/src/bindings.cpp:10-37) over CPU torch tensors, so tests can
run reference-shaped nn.Modules on the CPU with the oracle as their ``_backend``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (seconds)."""
    src = os.path.join(_HERE, "lion_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


class _Lib:
    def __init__(self):
        build()
        self._l = C.CDLL(_SO)

    # ---- K1/K2/K3 -------------------------------------------------------------------
    def avg_voxelize_forward(self, feat, coords, r):
        feat, pf = _f(feat)
        coords, pc = _i(coords)
        b, c, n = feat.shape
        r3 = r ** 3
        ind = np.empty((b, n), np.int32)
        cnt = np.empty((b, r3), np.int32)
        out = np.empty((b, c, r3), np.float32)
        self._l.orc_avg_voxelize_forward(b, c, n, r, pc, pf, ind.ctypes.data_as(_i32p),
                                         cnt.ctypes.data_as(_i32p), out.ctypes.data_as(_f32p))
        return out, ind, cnt

    def avg_voxelize_backward(self, gy, ind, cnt):
        gy, pg = _f(gy)
        ind, pi = _i(ind)
        cnt, pc = _i(cnt)
        b, c, r3 = gy.shape
        n = ind.shape[1]
        gx = np.empty((b, c, n), np.float32)
        self._l.orc_avg_voxelize_backward(b, c, n, r3, pg, pi, pc, gx.ctypes.data_as(_f32p))
        return gx

    # ---- K4/K5 ----------------------------------------------------------------------
    def trilinear_devoxelize_forward(self, r, training, coords, feat):
        coords, pc = _f(coords)
        feat, pf = _f(feat)
        b, c, r3 = feat.shape
        assert r3 == r ** 3
        n = coords.shape[2]
        outs = np.empty((b, c, n), np.float32)
        if training:
            inds = np.empty((b, 8, n), np.int32)
            wgts = np.empty((b, 8, n), np.float32)
        else:
            inds = np.zeros((1,), np.int32)
            wgts = np.zeros((1,), np.float32)
        self._l.orc_trilinear_devoxelize_forward(b, c, n, r, int(bool(training)), pc, pf,
                                                 inds.ctypes.data_as(_i32p),
                                                 wgts.ctypes.data_as(_f32p),
                                                 outs.ctypes.data_as(_f32p))
        return outs, inds, wgts

    def trilinear_devoxelize_backward(self, gy, inds, wgts, r):
        gy, pg = _f(gy)
        inds, pi = _i(inds)
        wgts, pw = _f(wgts)
        b, c, n = gy.shape
        r3 = r ** 3
        gx = np.empty((b, c, r3), np.float32)
        self._l.orc_trilinear_devoxelize_backward(b, c, n, r3, pg, pi, pw,
                                                  gx.ctypes.data_as(_f32p))
        return gx

    # ---- K6/K7/K8 -------------------------------------------------------------------
    def ball_query(self, centers, points, radius, u):
        centers, pc = _f(centers)
        points, pp = _f(points)
        b, _, m = centers.shape
        n = points.shape[2]
        idx = np.empty((b, m, u), np.int32)
        self._l.orc_ball_query(b, n, m, C.c_float(radius), u, pc, pp, idx.ctypes.data_as(_i32p))
        return idx

    def grouping_forward(self, feat, idx):
        feat, pf = _f(feat)
        idx, pi = _i(idx)
        b, c, n = feat.shape
        _, m, u = idx.shape
        out = np.empty((b, c, m, u), np.float32)
        self._l.orc_grouping_forward(b, c, n, m, u, pf, pi, out.ctypes.data_as(_f32p))
        return out

    def grouping_backward(self, gy, idx, n):
        gy, pg = _f(gy)
        idx, pi = _i(idx)
        b, c, m, u = gy.shape
        gx = np.empty((b, c, n), np.float32)
        self._l.orc_grouping_backward(b, c, n, m, u, pg, pi, gx.ctypes.data_as(_f32p))
        return gx

    # ---- K9/K10 ---------------------------------------------------------------------
    def furthest_point_sampling(self, coords, m):
        coords, pc = _f(coords)
        b, _, n = coords.shape
        idx = np.empty((b, m), np.int32)
        self._l.orc_furthest_point_sampling(b, n, m, pc, idx.ctypes.data_as(_i32p))
        return idx

    def gather_features_forward(self, feat, idx):
        feat, pf = _f(feat)
        idx, pi = _i(idx)
        b, c, n = feat.shape
        m = idx.shape[1]
        out = np.empty((b, c, m), np.float32)
        self._l.orc_gather_features_forward(b, c, n, m, pf, pi, out.ctypes.data_as(_f32p))
        return out

    def gather_features_backward(self, gy, idx, n):
        gy, pg = _f(gy)
        idx, pi = _i(idx)
        b, c, m = gy.shape
        gx = np.empty((b, c, n), np.float32)
        self._l.orc_gather_features_backward(b, c, n, m, pg, pi, gx.ctypes.data_as(_f32p))
        return gx

    # ---- K11/K12 --------------------------------------------------------------------
    def three_nn(self, points, centers):
        points, pp = _f(points)
        centers, pc = _f(centers)
        b, _, n = points.shape
        m = centers.shape[2]
        idx = np.empty((b, 3, n), np.int32)
        w = np.empty((b, 3, n), np.float32)
        self._l.orc_three_nn(b, n, m, pp, pc, idx.ctypes.data_as(_i32p), w.ctypes.data_as(_f32p))
        return idx, w

    def three_nn_interpolate_forward(self, points, centers, cfeat):
        idx, w = self.three_nn(points, centers)
        cfeat, pf = _f(cfeat)
        b, c, m = cfeat.shape
        n = idx.shape[2]
        out = np.empty((b, c, n), np.float32)
        self._l.orc_three_nn_interpolate_forward(b, c, m, n, pf, idx.ctypes.data_as(_i32p),
                                                 w.ctypes.data_as(_f32p),
                                                 out.ctypes.data_as(_f32p))
        return out, idx, w

    def three_nn_interpolate_backward(self, gy, idx, w, m):
        gy, pg = _f(gy)
        idx, pi = _i(idx)
        w, pw = _f(w)
        b, c, n = gy.shape
        gx = np.empty((b, c, m), np.float32)
        self._l.orc_three_nn_interpolate_backward(b, c, n, m, pg, pi, pw,
                                                  gx.ctypes.data_as(_f32p))
        return gx

    # ---- P1 -------------------------------------------------------------------------
    def voxelize_coords(self, coords, r, normalize=True, eps=0.0):
        coords, pc = _f(coords)
        b, _, n = coords.shape
        nc = np.empty((b, 3, n), np.float32)
        vox = np.empty((b, 3, n), np.int32)
        self._l.orc_voxelize_coords(b, n, r, int(bool(normalize)), C.c_float(eps), pc,
                                    nc.ctypes.data_as(_f32p), vox.ctypes.data_as(_i32p))
        return nc, vox

    # ---- E1 -------------------------------------------------------------------------
    def chamfer_forward(self, xyz1, xyz2):
        xyz1, p1 = _f(xyz1)
        xyz2, p2 = _f(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        d1 = np.empty((b, n), np.float32)
        d2 = np.empty((b, m), np.float32)
        i1 = np.empty((b, n), np.int32)
        i2 = np.empty((b, m), np.int32)
        self._l.orc_chamfer_forward(b, n, m, p1, p2, d1.ctypes.data_as(_f32p),
                                    d2.ctypes.data_as(_f32p), i1.ctypes.data_as(_i32p),
                                    i2.ctypes.data_as(_i32p))
        return d1, d2, i1, i2

    def chamfer_backward(self, xyz1, xyz2, gd1, gd2, idx1, idx2):
        xyz1, p1 = _f(xyz1)
        xyz2, p2 = _f(xyz2)
        gd1, pg1 = _f(gd1)
        gd2, pg2 = _f(gd2)
        idx1, pi1 = _i(idx1)
        idx2, pi2 = _i(idx2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        g1 = np.empty((b, n, 3), np.float32)
        g2 = np.empty((b, m, 3), np.float32)
        self._l.orc_chamfer_backward(b, n, m, p1, p2, pg1, pg2, pi1, pi2,
                                     g1.ctypes.data_as(_f32p), g2.ctypes.data_as(_f32p))
        return g1, g2

    # ---- E2 -------------------------------------------------------------------------
    def approxmatch(self, xyz1, xyz2):
        xyz1, p1 = _f(xyz1)
        xyz2, p2 = _f(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        match = np.empty((b, m, n), np.float32)
        self._l.orc_approxmatch(b, n, m, p1, p2, match.ctypes.data_as(_f32p))
        return match

    def matchcost(self, xyz1, xyz2, match):
        xyz1, p1 = _f(xyz1)
        xyz2, p2 = _f(xyz2)
        match, pm = _f(match)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        cost = np.empty((b,), np.float32)
        self._l.orc_matchcost(b, n, m, p1, p2, pm, cost.ctypes.data_as(_f32p))
        return cost

    def matchcost_backward(self, grad_cost, xyz1, xyz2, match):
        grad_cost, pg = _f(grad_cost)
        xyz1, p1 = _f(xyz1)
        xyz2, p2 = _f(xyz2)
        match, pm = _f(match)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        g1 = np.empty((b, n, 3), np.float32)
        g2 = np.empty((b, m, 3), np.float32)
        self._l.orc_matchcost_backward(b, n, m, pg, p1, p2, pm, g1.ctypes.data_as(_f32p),
                                       g2.ctypes.data_as(_f32p))
        return g1, g2

    # ---- D1 -------------------------------------------------------------------------
    def ddim_update(self, x, eps, z, s, c, sigma):
        x, px = _f(x)
        eps, pe = _f(eps)
        z, pz = _f(z)
        out = np.empty_like(x)
        self._l.orc_ddim_update(C.c_size_t(x.size), px, pe, pz, C.c_float(s), C.c_float(c),
                                C.c_float(sigma), out.ctypes.data_as(_f32p))
        return out

    def ddpm_update(self, x, eps, z, t_is_zero, k_outer, k_a, k_b, scale, temp):
        x, px = _f(x)
        eps, pe = _f(eps)
        z, pz = _f(z)
        out = np.empty_like(x)
        self._l.orc_ddpm_update(C.c_size_t(x.size), px, pe, pz, int(bool(t_is_zero)),
                                C.c_float(k_outer), C.c_float(k_a), C.c_float(k_b),
                                C.c_float(scale), C.c_float(temp), out.ctypes.data_as(_f32p))
        return out


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


def _n(t):
    """CPU tensor -> numpy (autograd-tracked inputs are detached: the backends never differentiate)."""
    return t.detach().contiguous().numpy()


class TorchBackend:
    """The 12 ``_backend`` callables of bindings.cpp:10-37 over CPU torch tensors."""

    def __init__(self):
        import torch
        self.torch = torch
        self.l = lib()

    def _t(self, a):
        return self.torch.from_numpy(a)

    def gather_features_forward(self, features, indices):
        return self._t(self.l.gather_features_forward(_n(features), _n(indices)))

    def gather_features_backward(self, grad_y, indices, n):
        return self._t(self.l.gather_features_backward(_n(grad_y), _n(indices), n))

    def furthest_point_sampling(self, coords, m):
        return self._t(self.l.furthest_point_sampling(_n(coords), m))

    def ball_query(self, centers, points, radius, k):
        return self._t(self.l.ball_query(_n(centers), _n(points), radius, k))

    def grouping_forward(self, features, indices):
        return self._t(self.l.grouping_forward(_n(features), _n(indices)))

    def grouping_backward(self, grad_y, indices, n):
        return self._t(self.l.grouping_backward(_n(grad_y), _n(indices), n))

    def three_nearest_neighbors_interpolate_forward(self, points, centers, cfeat):
        o, i, w = self.l.three_nn_interpolate_forward(_n(points), _n(centers),
                                                      _n(cfeat))
        return self._t(o), self._t(i), self._t(w)

    def three_nearest_neighbors_interpolate_backward(self, grad_y, idx, w, m):
        return self._t(self.l.three_nn_interpolate_backward(_n(grad_y), _n(idx),
                                                            _n(w), m))

    def trilinear_devoxelize_forward(self, r, is_training, coords, feats):
        o, i, w = self.l.trilinear_devoxelize_forward(r, is_training, _n(coords),
                                                      _n(feats))
        return self._t(o), self._t(i), self._t(w)

    def trilinear_devoxelize_backward(self, grad_y, inds, wgts, r):
        return self._t(self.l.trilinear_devoxelize_backward(_n(grad_y), _n(inds),
                                                            _n(wgts), r))

    def avg_voxelize_forward(self, feats, coords, r):
        o, i, c = self.l.avg_voxelize_forward(_n(feats), _n(coords), r)
        return self._t(o), self._t(i), self._t(c)

    def avg_voxelize_backward(self, grad_y, ind, cnt):
        return self._t(self.l.avg_voxelize_backward(_n(grad_y), _n(ind), _n(cnt)))

    # fused Voxelization.forward (not in the reference module; lion_amd.models call it)
    def voxelize_points_forward(self, features, coords, resolution, normalize=True, eps=0.0):
        nc, vox = self.l.voxelize_coords(_n(coords), int(resolution), normalize, eps)
        if features is None:
            b, _, n = coords.shape
            f0 = np.zeros((b, 1, n), np.float32)
            _, i, c = self.l.avg_voxelize_forward(f0, vox, int(resolution))
            return None, self._t(nc), self._t(i), self._t(c)
        o, i, c = self.l.avg_voxelize_forward(_n(features), vox, int(resolution))
        return self._t(o), self._t(nc), self._t(i), self._t(c)
