"""Pins oracle/lion_oracle.c against the reference's OWN kernel bodies executed on the CPU
(oracle/_ref/libref.so, built by oracle/ref_build.py from /root/reference with a fiber-based CUDA
execution shim).  Index / integer outputs and every atomics-free float output must be identical;
outputs the reference accumulates with atomicAdd are order-dependent there, so they are compared
bit-exactly only where one thread owns one point (n <= 512) and with 1e-6 otherwise."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import gaussian_cloud, surface_cloud, voxel_coords

SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(SO), reason="oracle/_ref/libref.so not built (needs /root/reference)")

fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)


def F(a):
    a = np.ascontiguousarray(a, np.float32)
    return a, a.ctypes.data_as(fp)


def I(a):
    a = np.ascontiguousarray(a, np.int32)
    return a, a.ctypes.data_as(ip)


@pytest.fixture(scope="module")
def ref():
    return C.CDLL(SO)


@pytest.mark.parametrize("C_,N,r", [(5, 300, 8), (3, 512, 16), (6, 256, 8), (4, 2048, 32), (2, 1300, 16)])
def test_voxelize(ref, orc, C_, N, r):
    rng = np.random.default_rng(N)
    B = 2
    vc = np.rint(voxel_coords(rng, B, N, r, "surface")).astype(np.int32)
    feat = rng.standard_normal((B, C_, N)).astype(np.float32)
    o_out, o_ind, o_cnt = orc.avg_voxelize_forward(feat, vc, r)
    out = np.empty_like(o_out); ind = np.empty_like(o_ind); cnt = np.empty_like(o_cnt)
    _, pv = I(vc); _, pf = F(feat)
    ref.ref_avg_voxelize_forward(B, C_, N, r, pv, pf, ind.ctypes.data_as(ip), cnt.ctypes.data_as(ip), out.ctypes.data_as(fp))
    assert np.array_equal(ind, o_ind) and np.array_equal(cnt, o_cnt)
    if N <= 512 and (N & (N - 1)) == 0:  # one point per thread: accumulation order == point order
        assert np.array_equal(out, o_out)
    else:
        np.testing.assert_allclose(out, o_out, rtol=1e-6, atol=1e-6)
    gy = rng.standard_normal((B, C_, r ** 3)).astype(np.float32)
    gx = np.empty((B, C_, N), np.float32)
    _, pg = F(gy)
    ref.ref_avg_voxelize_backward(B, C_, N, r ** 3, pg, ind.ctypes.data_as(ip), cnt.ctypes.data_as(ip), gx.ctypes.data_as(fp))
    assert np.array_equal(gx, orc.avg_voxelize_backward(gy, o_ind, o_cnt))


@pytest.mark.parametrize("C_,N,r", [(6, 400, 8), (4, 512, 8), (3, 1500, 16)])
def test_devoxelize(ref, orc, C_, N, r):
    rng = np.random.default_rng(N + 1)
    B = 2
    co = voxel_coords(rng, B, N, r)
    co[:, :, :3] = np.array([[0.0, r - 1.0, 2.0]] * 3, np.float32)
    feat = rng.standard_normal((B, C_, r ** 3)).astype(np.float32)
    o_out, o_i, o_w = orc.trilinear_devoxelize_forward(r, True, co, feat)
    out = np.empty_like(o_out); inds = np.empty_like(o_i); wgts = np.empty_like(o_w)
    _, pc = F(co); _, pf = F(feat)
    ref.ref_trilinear_devoxelize_forward(B, C_, N, r, 1, pc, pf, inds.ctypes.data_as(ip), wgts.ctypes.data_as(fp), out.ctypes.data_as(fp))
    assert np.array_equal(inds, o_i) and np.array_equal(wgts, o_w) and np.array_equal(out, o_out)
    gy = rng.standard_normal((B, C_, N)).astype(np.float32)
    gx = np.empty((B, C_, r ** 3), np.float32)
    _, pg = F(gy)
    ref.ref_trilinear_devoxelize_backward(B, C_, N, r ** 3, pg, inds.ctypes.data_as(ip), wgts.ctypes.data_as(fp), gx.ctypes.data_as(fp))
    o_gx = orc.trilinear_devoxelize_backward(gy, o_i, o_w, r)
    if N <= 512 and (N & (N - 1)) == 0:
        assert np.array_equal(gx, o_gx)
    else:
        np.testing.assert_allclose(gx, o_gx, rtol=1e-5, atol=1e-5)


def test_ball_query_grouping_gather(ref, orc):
    rng = np.random.default_rng(3)
    B, N, M, U, C_ = 2, 700, 90, 16, 5
    pts = (gaussian_cloud(rng, B, N) * 0.4).astype(np.float32)
    ctr = pts[:, :, :M].copy()
    for radius in (0.15, 0.4, 5.0, 1e-6):
        idx = np.empty((B, M, U), np.int32)
        _, pc = F(ctr); _, pp = F(pts)
        ref.ref_ball_query(B, N, M, C.c_float(radius), U, pc, pp, idx.ctypes.data_as(ip))
        assert np.array_equal(idx, orc.ball_query(ctr, pts, radius, U)), radius
    feat = rng.standard_normal((B, C_, N)).astype(np.float32)
    out = np.empty((B, C_, M, U), np.float32)
    _, pf = F(feat)
    ref.ref_grouping_forward(B, C_, N, M, U, pf, idx.ctypes.data_as(ip), out.ctypes.data_as(fp))
    assert np.array_equal(out, orc.grouping_forward(feat, idx))
    gy = rng.standard_normal((B, C_, M, U)).astype(np.float32)
    gx = np.empty((B, C_, N), np.float32)
    _, pg = F(gy)
    ref.ref_grouping_backward(B, C_, N, M, U, pg, idx.ctypes.data_as(ip), gx.ctypes.data_as(fp))
    np.testing.assert_allclose(gx, orc.grouping_backward(gy, idx, N), rtol=1e-5, atol=1e-5)
    i1 = np.ascontiguousarray(idx[:, :, 0])
    o = np.empty((B, C_, M), np.float32)
    ref.ref_gather_features_forward(B, C_, N, M, pf, i1.ctypes.data_as(ip), o.ctypes.data_as(fp))
    assert np.array_equal(o, orc.gather_features_forward(feat, i1))


@pytest.mark.parametrize("N,M", [(300, 64), (1300, 100), (2048, 256)])
def test_fps_including_tie_break(ref, orc, N, M):
    rng = np.random.default_rng(N)
    B = 2
    co = gaussian_cloud(rng, B, N)
    co[1, :, N // 2:] = co[1, :, : N - N // 2]  # exact duplicates -> exact distance ties
    idx = np.empty((B, M), np.int32)
    _, pc = F(co)
    ref.ref_furthest_point_sampling(B, N, M, pc, idx.ctypes.data_as(ip))
    assert np.array_equal(idx, orc.furthest_point_sampling(co, M))


def test_fps_lattice_ties(ref, orc):
    g3 = np.stack(np.meshgrid(np.arange(6), np.arange(6), np.arange(6), indexing="ij"), 0).reshape(3, -1)
    co = np.concatenate([g3, g3, g3, g3], 1)[None].astype(np.float32)  # 864 points, heavy ties, N > 512
    rng = np.random.default_rng(0)
    co = np.ascontiguousarray(co[:, :, rng.permutation(co.shape[2])])
    idx = np.empty((1, 150), np.int32)
    _, pc = F(co)
    ref.ref_furthest_point_sampling(1, co.shape[2], 150, pc, idx.ctypes.data_as(ip))
    assert np.array_equal(idx, orc.furthest_point_sampling(co, 150))


@pytest.mark.parametrize("N,M", [(500, 40), (200, 2), (64, 16)])
def test_three_nn(ref, orc, N, M):
    rng = np.random.default_rng(M)
    B, C_ = 2, 6
    pts, ctr = gaussian_cloud(rng, B, N), gaussian_cloud(rng, B, M)
    cf = rng.standard_normal((B, C_, M)).astype(np.float32)
    o_out, o_idx, o_w = orc.three_nn_interpolate_forward(pts, ctr, cf)
    idx = np.empty_like(o_idx); w = np.empty_like(o_w); out = np.empty_like(o_out)
    _, pp = F(pts); _, pc = F(ctr); _, pf = F(cf)
    ref.ref_three_nn_interpolate_forward(B, C_, M, N, pp, pc, pf, idx.ctypes.data_as(ip), w.ctypes.data_as(fp), out.ctypes.data_as(fp))
    assert np.array_equal(idx, o_idx) and np.array_equal(w, o_w) and np.array_equal(out, o_out)
    gy = rng.standard_normal((B, C_, N)).astype(np.float32)
    gx = np.empty((B, C_, M), np.float32)
    _, pg = F(gy)
    ref.ref_three_nn_interpolate_backward(B, C_, N, M, pg, idx.ctypes.data_as(ip), w.ctypes.data_as(fp), gx.ctypes.data_as(fp))
    np.testing.assert_allclose(gx, orc.three_nn_interpolate_backward(gy, o_idx, o_w, M), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,N,M", [(4, 100, 200), (2, 700, 1300)])
def test_chamfer(ref, orc, B, N, M):
    rng = np.random.default_rng(N)
    x1, x2 = rng.random((B, N, 3)).astype(np.float32), rng.random((B, M, 3)).astype(np.float32)
    x2[0, 5] = x2[0, 600 % M]  # duplicate targets across 512-tiles: tie -> lowest index
    o = orc.chamfer_forward(x1, x2)
    d1 = np.empty((B, N), np.float32); d2 = np.empty((B, M), np.float32)
    i1 = np.empty((B, N), np.int32); i2 = np.empty((B, M), np.int32)
    _, p1 = F(x1); _, p2 = F(x2)
    ref.ref_chamfer_forward(B, N, M, p1, p2, d1.ctypes.data_as(fp), d2.ctypes.data_as(fp), i1.ctypes.data_as(ip), i2.ctypes.data_as(ip))
    assert np.array_equal(i1, o[2]) and np.array_equal(i2, o[3])
    assert np.array_equal(d1, o[0]) and np.array_equal(d2, o[1])
    g1, g2 = rng.standard_normal((B, N)).astype(np.float32), rng.standard_normal((B, M)).astype(np.float32)
    gx1 = np.empty((B, N, 3), np.float32); gx2 = np.empty((B, M, 3), np.float32)
    _, pg1 = F(g1); _, pg2 = F(g2)
    ref.ref_chamfer_backward(B, N, M, p1, p2, pg1, pg2, i1.ctypes.data_as(ip), i2.ctypes.data_as(ip), gx1.ctypes.data_as(fp), gx2.ctypes.data_as(fp))
    og1, og2 = orc.chamfer_backward(x1, x2, g1, g2, i1, i2)
    np.testing.assert_allclose(gx1, og1, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gx2, og2, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,N,M", [(2, 96, 96), (1, 200, 120), (33, 16, 16)])
def test_emd(ref, orc, B, N, M):
    """approxmatch (10 levels x 3 passes) and matchcost (512-thread partial sums + tree) bit for bit;
    B=33 exercises the reference's `for i = blockIdx.x; i < b; i += gridDim.x` loop over 32 blocks."""
    rng = np.random.default_rng(N + M)
    x1, x2 = rng.random((B, N, 3)).astype(np.float32), rng.random((B, M, 3)).astype(np.float32)
    o_match = orc.approxmatch(x1, x2)
    match = np.empty_like(o_match)
    _, p1 = F(x1); _, p2 = F(x2)
    ref.ref_approxmatch(B, N, M, p1, p2, match.ctypes.data_as(fp))
    assert np.array_equal(match, o_match)
    cost = np.empty((B,), np.float32)
    ref.ref_matchcost(B, N, M, p1, p2, match.ctypes.data_as(fp), cost.ctypes.data_as(fp))
    assert np.array_equal(cost, orc.matchcost(x1, x2, o_match))
    gc = rng.standard_normal((B,)).astype(np.float32)
    g1 = np.empty((B, N, 3), np.float32); g2 = np.empty((B, M, 3), np.float32)
    _, pg = F(gc)
    ref.ref_matchcost_backward(B, N, M, pg, p1, p2, match.ctypes.data_as(fp), g1.ctypes.data_as(fp), g2.ctypes.data_as(fp))
    og1, og2 = orc.matchcost_backward(gc, x1, x2, o_match)
    assert np.array_equal(g1, og1)
    np.testing.assert_allclose(g2, og2, rtol=1e-5, atol=1e-6)
