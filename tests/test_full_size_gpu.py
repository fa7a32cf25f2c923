"""-m gpu: BASELINE.json configs[1] at its REAL size (B = 32 clouds of 2048 points) against the CPU oracle.

The small-batch parity tests (test_hip_parity_gpu.py) never reach the batch-dependent code paths: the
voxelize kernel's workgroup -> (cloud, slab, XCD) mapping, the devoxelize slab schedule, the grids of
the geometry kernels.  Here every (C, N, r) tuple the local prior evaluates at r = 32 / 16 / 8 is run at
B = 32 and compared with the oracle bit for bit (integer outputs AND floats: the kernels keep the
oracle's summation order), including the `affine` devoxelize variant the fused PVConv uses.
"""
import numpy as np
import pytest
import torch

from conftest import gaussian_cloud, surface_cloud

pytestmark = pytest.mark.gpu

B = 32


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def bk():
    from lion_amd.functional.backend import _backend
    _backend.lib
    return _backend


def _cloud(rng, n, kind):
    co = gaussian_cloud(rng, B, n) if kind == "gauss" else surface_cloud(rng, B, n)
    return (co * 0.7 + 0.1).astype(np.float32)


# (C, N, r): the three PVConv stages of PVCNN2Prior that dominate the voxel traffic
FULL_CASES = [(64, 2048, 32), (128, 1024, 16), (128, 256, 8)]


@pytest.mark.parametrize("C,N,r", FULL_CASES)
@pytest.mark.parametrize("kind", ["gauss", "surface"])
def test_voxelize_points_full_batch_bit_exact(bk, orc, C, N, r, kind):
    """P1 + K1 + K2 (vox.cu:18-72, pvcnn2_ada.py:173-188) at B = 32: norm coords, voxel ids, counts and the
    float means, all bit-exact."""
    rng = np.random.default_rng(C + N + r)
    co = _cloud(rng, N, kind)
    feat = rng.standard_normal((B, C, N)).astype(np.float32)
    o_nc, o_vc = orc.voxelize_coords(co, r, True, 0.0)
    o_out, o_ind, o_cnt = orc.avg_voxelize_forward(feat, o_vc, r)
    out, nc, ind, cnt = bk.voxelize_points_forward(dev(feat), dev(co), r, True, 0.0)
    assert np.array_equal(host(nc), o_nc)
    assert np.array_equal(host(ind), o_ind)
    assert np.array_equal(host(cnt), o_cnt)
    assert np.array_equal(host(out), o_out)
    # the reference's own entry point (integer coords in, vox.cpp:17-43) on the same batch
    out2, ind2, cnt2 = bk.avg_voxelize_forward(dev(feat), dev(o_vc), r)
    assert np.array_equal(host(ind2), o_ind) and np.array_equal(host(cnt2), o_cnt)
    assert np.array_equal(host(out2), o_out)


@pytest.mark.parametrize("C,N,r", FULL_CASES)
def test_devoxelize_full_batch_bit_exact(bk, orc, C, N, r):
    """K4 (trilinear_devox.cu:21-105) at B = 32, eval and training outputs; then the `affine` variant
    (scale * interp + shift * sum(w), the AdaGN x SE fold of the fused PVConv) against the same expression
    evaluated in float32 on the oracle's interpolation and weights."""
    rng = np.random.default_rng(C * 3 + N + r)
    co = _cloud(rng, N, "gauss")
    o_nc, _ = orc.voxelize_coords(co, r, True, 0.0)
    grid = rng.standard_normal((B, C, r ** 3)).astype(np.float32)
    o_out, o_inds, o_wgts = orc.trilinear_devoxelize_forward(r, True, o_nc, grid)
    d_nc, d_grid = dev(o_nc), dev(grid)
    out, inds, wgts = bk.trilinear_devoxelize_forward(r, True, d_nc, d_grid)
    assert np.array_equal(host(inds), o_inds)
    assert np.array_equal(host(wgts), o_wgts)
    assert np.array_equal(host(out), o_out)
    out_e, _, _ = bk.trilinear_devoxelize_forward(r, False, d_nc, d_grid)
    assert np.array_equal(host(out_e), o_out)
    # affine variant
    from lion_amd.fused_ops import devoxelize_affine
    scale = rng.standard_normal((B, C)).astype(np.float32)
    shift = rng.standard_normal((B, C)).astype(np.float32)
    wsum = o_wgts[:, 0].copy()
    for k in range(1, 8):                       # float32, ascending corner order: the kernel's order
        wsum = (wsum + o_wgts[:, k]).astype(np.float32)
    expect = (o_out * scale[:, :, None]).astype(np.float32) + (shift[:, :, None] * wsum[:, None, :]).astype(np.float32)
    got = devoxelize_affine(d_grid, d_nc, r, dev(scale), dev(shift))
    assert np.array_equal(host(got), expect.astype(np.float32))


def test_geometry_full_batch_bit_exact(bk, orc):
    """K9 / K10 / K6 / K7 / K11 / K12 at B = 32 on the first set-abstraction / last feature-propagation stage."""
    rng = np.random.default_rng(2048)
    N, M, U = 2048, 1024, 32
    co = _cloud(rng, N, "gauss")
    o_idx = orc.furthest_point_sampling(co, M)
    idx = bk.furthest_point_sampling(dev(co), M)
    assert np.array_equal(host(idx), o_idx)
    o_ctr = orc.gather_features_forward(co, o_idx)
    ctr = bk.gather_features_forward(dev(co), idx)
    assert np.array_equal(host(ctr), o_ctr)
    o_nb = orc.ball_query(o_ctr, co, 0.1, U)
    nb = bk.ball_query(ctr, dev(co), 0.1, U)
    assert np.array_equal(host(nb), o_nb)
    feat = rng.standard_normal((B, 35, N)).astype(np.float32)
    assert np.array_equal(host(bk.grouping_forward(dev(feat), nb)), orc.grouping_forward(feat, o_nb))
    cf = rng.standard_normal((B, 192, M)).astype(np.float32)
    o_it, o_ii, o_iw = orc.three_nn_interpolate_forward(co, o_ctr, cf)
    it, ii, iw = bk.three_nearest_neighbors_interpolate_forward(dev(co), ctr, dev(cf))
    assert np.array_equal(host(ii), o_ii) and np.array_equal(host(iw), o_iw)
    assert np.array_equal(host(it), o_it)


def test_backward_scatters_full_batch_vs_fp64(bk, orc):
    """K3 / K5 / K8 / K12-grad at B = 32.  The scatters accumulate in LDS with float atomics (order free, as in the
    reference), so they are compared at north_star's 1e-5 with the sum carried out in float64."""
    rng = np.random.default_rng(77)
    C, N, r = 64, 2048, 32
    co = _cloud(rng, N, "surface")
    o_nc, o_vc = orc.voxelize_coords(co, r, True, 0.0)
    feat = rng.standard_normal((B, C, N)).astype(np.float32)
    _, o_ind, o_cnt = orc.avg_voxelize_forward(feat, o_vc, r)
    gy = rng.standard_normal((B, C, r ** 3)).astype(np.float32)
    # K3: single writer per point -> exact
    assert np.array_equal(host(bk.avg_voxelize_backward(dev(gy), dev(o_ind), dev(o_cnt))),
                          orc.avg_voxelize_backward(gy, o_ind, o_cnt))
    # K5: 8-corner scatter-add
    Cs = 16
    grid = rng.standard_normal((B, Cs, r ** 3)).astype(np.float32)
    _, inds, wgts = orc.trilinear_devoxelize_forward(r, True, o_nc, grid)
    gyp = rng.standard_normal((B, Cs, N)).astype(np.float32)
    ref = np.zeros((B, Cs, r ** 3), np.float64)
    for b in range(B):
        for k in range(8):
            np.add.at(ref[b], (slice(None), inds[b, k]), gyp[b].astype(np.float64) * wgts[b, k].astype(np.float64))
    got = host(bk.trilinear_devoxelize_backward(dev(gyp), dev(inds), dev(wgts), r))
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
