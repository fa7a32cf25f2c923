"""-m gpu: the HIP kernels (called through the C ABI via lion_amd.functional.backend) against the
CPU oracle on identical seeded inputs.  Integer / index outputs must be bit-exact; float outputs
are bit-exact where the kernel reproduces the oracle's summation order (forward paths), and
within 1e-5 (north_star tolerance) of the float64 sum where atomics make the order free (backward
scatters; the reference leaves that order to atomicAdd as well).  B = 32 versions: test_full_size_gpu.py."""
import numpy as np
import pytest
import torch

from conftest import gaussian_cloud, surface_cloud, voxel_coords

pytestmark = pytest.mark.gpu

TOL = 1e-5  # north_star: fp32 within 1e-5


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def bk():
    from lion_amd.functional.backend import _backend
    _backend.lib  # loads the .so, raises if missing
    return _backend


@pytest.fixture(params=["fp32-mfma", "fp16x2-split"])
def conv_kernel(request, monkeypatch):
    """runs a test once per voxel-convolution kernel: the exact-fp32 MFMA kernel (csrc/conv3d.hip) and the
    split-operand kernel on the 16-bit pipe (csrc/conv3d_split.hip; shapes with Cin % 16 != 0 stay on fp32 there)."""
    from lion_amd import conv_ops
    monkeypatch.setattr(conv_ops, "SPLIT", request.param == "fp16x2-split")
    return request.param


# (C, N, r) tuples of one PVCNN2Prior forward, plus edge cases
VOX_CASES = [(4, 2048, 32), (32, 2048, 32), (128, 1024, 16), (192, 256, 8), (128, 64, 8),
             (64, 2048, 32), (3, 1, 8), (5, 777, 16), (7, 4096, 16), (2, 100, 6)]


@pytest.mark.parametrize("C,N,r", VOX_CASES)
@pytest.mark.parametrize("kind", ["gauss", "surface"])
def test_avg_voxelize_forward_bit_exact(bk, orc, C, N, r, kind):
    rng = np.random.default_rng(C * 131 + N + r)
    B = 3
    vc = np.rint(voxel_coords(rng, B, N, r, kind)).astype(np.int32)
    feat = rng.standard_normal((B, C, N)).astype(np.float32)
    o_out, o_ind, o_cnt = orc.avg_voxelize_forward(feat, vc, r)
    out, ind, cnt = bk.avg_voxelize_forward(dev(feat), dev(vc), r)
    assert np.array_equal(host(ind), o_ind)
    assert np.array_equal(host(cnt), o_cnt)
    assert np.array_equal(host(out), o_out), np.abs(host(out) - o_out).max()


def test_avg_voxelize_all_points_one_voxel(bk, orc):
    B, C, N, r = 2, 8, 512, 8
    rng = np.random.default_rng(1)
    vc = np.full((B, 3, N), 3, np.int32)
    feat = rng.standard_normal((B, C, N)).astype(np.float32)
    o_out, o_ind, o_cnt = orc.avg_voxelize_forward(feat, vc, r)
    out, ind, cnt = bk.avg_voxelize_forward(dev(feat), dev(vc), r)
    assert np.array_equal(host(cnt), o_cnt) and np.array_equal(host(ind), o_ind)
    assert np.array_equal(host(out), o_out)


def test_avg_voxelize_fallback_large_n(bk, orc):
    # N > 8192 takes the atomic fallback: indices exact, floats within tolerance
    B, C, N, r = 2, 6, 9000, 16
    rng = np.random.default_rng(2)
    vc = np.rint(voxel_coords(rng, B, N, r)).astype(np.int32)
    feat = rng.standard_normal((B, C, N)).astype(np.float32)
    o_out, o_ind, o_cnt = orc.avg_voxelize_forward(feat, vc, r)
    out, ind, cnt = bk.avg_voxelize_forward(dev(feat), dev(vc), r)
    assert np.array_equal(host(cnt), o_cnt) and np.array_equal(host(ind), o_ind)
    np.testing.assert_allclose(host(out), o_out, rtol=TOL, atol=TOL)


@pytest.mark.parametrize("C,N,r", [(32, 2048, 32), (128, 1024, 16), (5, 333, 8)])
def test_avg_voxelize_backward(bk, orc, C, N, r):
    rng = np.random.default_rng(7)
    B = 3
    vc = np.rint(voxel_coords(rng, B, N, r, "surface")).astype(np.int32)
    feat = rng.standard_normal((B, C, N)).astype(np.float32)
    _, o_ind, o_cnt = orc.avg_voxelize_forward(feat, vc, r)
    gy = rng.standard_normal((B, C, r ** 3)).astype(np.float32)
    o_gx = orc.avg_voxelize_backward(gy, o_ind, o_cnt)
    gx = bk.avg_voxelize_backward(dev(gy), dev(o_ind), dev(o_cnt))
    assert np.array_equal(host(gx), o_gx)


@pytest.mark.parametrize("N,r", [(2048, 32), (1024, 16), (256, 8), (64, 8), (1000, 16), (1, 4)])
@pytest.mark.parametrize("kind", ["gauss", "surface"])
def test_voxelize_points_fused_p1(bk, orc, N, r, kind):
    """P1 fused into phase 1: norm_coords and voxel ids bit-exact vs the oracle's fixed tree."""
    rng = np.random.default_rng(N + r)
    B, C = 4, 6
    co = (gaussian_cloud(rng, B, N) if kind == "gauss" else surface_cloud(rng, B, N))
    co = (co * 0.7 + 0.1).astype(np.float32)
    if N == 1:
        co[:] = np.array([[0.3], [0.2], [0.9]], np.float32)  # max-norm 0 -> NaN path skipped below
    feat = rng.standard_normal((B, C, N)).astype(np.float32)
    o_nc, o_vc = orc.voxelize_coords(co, r, True, 0.0)
    out, nc, ind, cnt = bk.voxelize_points_forward(dev(feat), dev(co), r, True, 0.0)
    if N == 1:
        return  # 0/0: undefined in the reference too
    assert np.array_equal(host(nc), o_nc)
    o_out, o_ind, o_cnt = orc.avg_voxelize_forward(feat, o_vc, r)
    assert np.array_equal(host(ind), o_ind)
    assert np.array_equal(host(cnt), o_cnt)
    assert np.array_equal(host(out), o_out)


@pytest.mark.parametrize("C,N,r", [(6, 2048, 32), (70, 2048, 32), (64, 2048, 32), (128, 1024, 16), (9, 1024, 16), (130, 256, 8), (5, 64, 8),
                                   (3, 1000, 16), (20, 4096, 32), (12, 8192, 32)])
@pytest.mark.parametrize("kind", ["gauss", "surface", "clump", "one-voxel"])
def test_voxel_index_then_scatter_equals_fused_and_oracle(bk, orc, C, N, r, kind):
    """The two-step form the models use (lion_voxel_index once per (cloud, r), lion_voxel_scatter per feature tensor):
    norm_coords / ids / counts and the float means bit-identical to the fused entry point and to the oracle -- also on
    clouds that put hundreds of points into one voxel (2 % outliers set the normalisation; the latents of a sampling
    chain look like this) and on a cloud that collapses into a single voxel (round 5: on those, the scatter's workgroups
    of empty slabs take over channel chunks of the crowded slab; the long ordered sums run one lane per channel).
    vox.cu:18-72, pvcnn2_ada.py:173-188."""
    rng = np.random.default_rng(C + N + r)
    B = 5
    co = surface_cloud(rng, B, N) if kind == "surface" else gaussian_cloud(rng, B, N)
    if kind == "clump":
        co[:, :, : int(N * 0.98)] *= 0.08
    if kind == "one-voxel":
        co[:, :, 1:] = co[:, :, 1:] * 1e-4 + 0.3
    feat = rng.standard_normal((B, C, N)).astype(np.float32)
    o_nc, o_vc = orc.voxelize_coords(co, r, True, 0.0)
    o_out, o_ind, o_cnt = orc.avg_voxelize_forward(feat, o_vc, r)
    plan = bk.voxel_index(dev(co), r, True, 0.0)
    assert plan is not None
    assert np.array_equal(host(plan["norm"]), o_nc)
    assert np.array_equal(host(plan["ind"]), o_ind)
    assert np.array_equal(host(plan["cnt"]), o_cnt)
    out = bk.voxel_scatter(dev(feat), plan)
    assert np.array_equal(host(out), o_out)
    out2 = bk.voxel_scatter(dev(feat[:, : max(C // 2, 1)].copy()), plan)        # the plan serves any channel count
    assert np.array_equal(host(out2), o_out[:, : max(C // 2, 1)])
    f_out, f_nc, f_ind, f_cnt = bk.voxelize_points_forward(dev(feat), dev(co), r, True, 0.0)
    assert torch.equal(f_out, out) and torch.equal(f_nc, plan["norm"]) and torch.equal(f_ind, plan["ind"])
    assert torch.equal(f_cnt, plan["cnt"])


def test_voxelize_points_no_normalize_and_eps(bk, orc):
    rng = np.random.default_rng(11)
    B, C, N, r = 2, 3, 500, 16
    co = (rng.uniform(-1, 1, (B, 3, N))).astype(np.float32)
    feat = rng.standard_normal((B, C, N)).astype(np.float32)
    for normalize, eps in [(False, 0.0), (True, 1e-3)]:
        o_nc, o_vc = orc.voxelize_coords(co, r, normalize, eps)
        out, nc, ind, cnt = bk.voxelize_points_forward(dev(feat), dev(co), r, normalize, eps)
        assert np.array_equal(host(nc), o_nc)
        o_out, o_ind, _ = orc.avg_voxelize_forward(feat, o_vc, r)
        assert np.array_equal(host(ind), o_ind) and np.array_equal(host(out), o_out)


DEVOX_CASES = [(32, 2048, 32), (64, 2048, 32), (64, 1024, 16), (128, 256, 8), (128, 64, 8), (3, 1, 4), (7, 999, 16)]


@pytest.mark.parametrize("C,N,r", DEVOX_CASES)
@pytest.mark.parametrize("training", [False, True])
def test_trilinear_devoxelize_forward_bit_exact(bk, orc, C, N, r, training):
    rng = np.random.default_rng(C + N + r)
    B = 3
    co = voxel_coords(rng, B, N, r)
    co[:, :, : min(N, 4)] = np.array([[0.0, r - 1, 1.0, 2.5][: min(N, 4)]] * 3, np.float32)  # exact/edge coords
    feat = rng.standard_normal((B, C, r ** 3)).astype(np.float32)
    o_out, o_inds, o_wgts = orc.trilinear_devoxelize_forward(r, training, co, feat)
    out, inds, wgts = bk.trilinear_devoxelize_forward(r, training, dev(co), dev(feat))
    assert np.array_equal(host(out), o_out), np.abs(host(out) - o_out).max()
    if training:
        assert np.array_equal(host(inds), o_inds)
        assert np.array_equal(host(wgts), o_wgts)
    else:
        assert tuple(inds.shape) == (1,) and tuple(wgts.shape) == (1,)


@pytest.mark.parametrize("training", [False, True])
def test_trilinear_devoxelize_r32_row_paths(bk, orc, training):
    """devox_ring_kernel (r = 32) moves only the 32-byte pieces of z-rows some point reads, compacted in LDS.  Its three other paths:
    (a) a cloud spread over every (x, y) column needs more pieces than an LDS buffer holds -> in-kernel global gather;
    (b) z in (r-1, r): the reference's flat index arithmetic makes the "+1" corner the first element of the NEXT row
        (Voxelization clamps to r-1, so only a foreign caller gets there) -> per-point global path, same flat indices;
    (c) coordinates whose corners leave the grid's memory -> 0 instead of an out-of-bounds read (documented deviation:
        the reference reads past the buffer)."""
    rng = np.random.default_rng(32)
    B, C, N, r = 3, 10, 2048, 32
    feat = rng.standard_normal((B, C, r ** 3)).astype(np.float32)
    co = rng.uniform(0, r - 1, (B, 3, N)).astype(np.float32)           # (a) in cloud 0 and 1
    co[2] = voxel_coords(rng, 1, N, r)[0]                                # sparse cloud next to them
    co[2, 2, :50] = (r - 1) + rng.uniform(0.1, 0.9, 50).astype(np.float32)   # (b) z beyond the clamp range
    co[2, 0, :50] = np.minimum(co[2, 0, :50], r - 3)                     # ... with every flat index still inside the grid
    co[1, 2, 100:120] = (r - 1) + 0.5                                    # (b) inside a dense cloud as well
    co[1, 0, 100:120] = np.minimum(co[1, 0, 100:120], r - 3)
    o_out, o_inds, o_wgts = orc.trilinear_devoxelize_forward(r, training, co, feat)
    out, inds, wgts = bk.trilinear_devoxelize_forward(r, training, dev(co), dev(feat))
    assert np.array_equal(host(out), o_out), np.abs(host(out) - o_out).max()
    if training:
        assert np.array_equal(host(inds), o_inds) and np.array_equal(host(wgts), o_wgts)
    # (c)
    co2 = co[2:3].copy()
    co2[0, :, 7] = (r - 1) + 0.5                                         # all three corners +1 out of the grid
    co2[0, 0, 9] = -1.5
    out2, _, _ = bk.trilinear_devoxelize_forward(r, False, dev(co2), dev(feat[2:3]))
    o2 = host(out2)
    assert np.all(o2[0, :, 7] == 0) and np.all(o2[0, :, 9] == 0)
    keep = np.ones(N, bool); keep[[7, 9]] = False
    assert np.array_equal(o2[0][:, keep], o_out[2][:, keep])


@pytest.mark.parametrize("N", [2048, 999])
def test_trilinear_devoxelize_planned_equals_one_step(bk, orc, N):
    """K4 in two steps (lion_trilinear_devoxelize_plan + ..._planned_forward, what a PVConv at r = 32 runs): the plan of a
    cloud serves every feature tensor devoxelised at its coordinates; plain and affine forms are bit-identical to the
    one-step entry points and to the oracle -- regular clouds, a cloud dense enough for the global-gather path, points
    with z beyond the clamp range and out-of-contract coordinates."""
    import torch
    from lion_amd import fused_ops as fo, _lib
    rng = np.random.default_rng(N)
    B, r = 4, 32
    co = voxel_coords(rng, B, N, r)
    co[0] = rng.uniform(0, r - 1, (3, N)).astype(np.float32)            # dense: more pieces than a ring buffer holds
    co[1, 2, :40] = (r - 1) + rng.uniform(0.1, 0.9, 40).astype(np.float32)  # z beyond the clamp range (flat indexing)
    co[1, 0, :40] = np.minimum(co[1, 0, :40], r - 3)
    co[2, :, 5] = (r - 1) + 0.5                                          # corners outside the grid's memory -> 0
    co[2, :, :4] = np.array([[0.0, r - 1, 1.0, 2.5]] * 3, np.float32)     # exact / edge coordinates
    d_co = dev(co)
    plan = fo.devoxelize_plan(d_co, r)
    assert plan is not None
    lib = _lib.load()
    for C in (10, 64):
        feat = rng.standard_normal((B, C, r ** 3)).astype(np.float32)
        d_feat = dev(feat)
        o_out, _, _ = orc.trilinear_devoxelize_forward(r, False, co, feat)
        one, _, _ = bk.trilinear_devoxelize_forward(r, False, d_co, d_feat)
        two = torch.empty_like(one)
        _lib.check(lib.lion_trilinear_devoxelize_planned_forward(
            _lib.ptr(plan["buf"]), plan["buf"].numel(), _lib.ptr(d_co), _lib.ptr(d_feat), None, None, B, C, N, r,
            _lib.ptr(two), _lib.stream_ptr(d_co.device)), "planned")
        assert torch.equal(one, two)
        keep = np.ones(N, bool); keep[5] = False
        assert np.array_equal(host(two)[:, :, keep][[0, 1, 3]], o_out[:, :, keep][[0, 1, 3]])
        assert np.array_equal(host(two)[2][:, keep], o_out[2][:, keep]) and np.all(host(two)[2][:, 5] == 0)
        scale = dev(rng.uniform(0.5, 1.5, (B, C)).astype(np.float32)); shift = dev(rng.standard_normal((B, C)).astype(np.float32))
        g5 = d_feat.view(B, C, r, r, r)
        a1 = fo.devoxelize_affine(g5, d_co, r, scale, shift)
        a2 = fo.devoxelize_affine(g5, d_co, r, scale, shift, plan=plan)
        assert torch.equal(a1, a2)
    assert fo.devoxelize_plan(dev(co[:, :, :64].copy()), 16) is None     # outside the planned kernel's range


@pytest.mark.parametrize("C,N,r", [(32, 2048, 32), (64, 1024, 16), (16, 300, 8)])
def test_trilinear_devoxelize_backward(bk, orc, C, N, r):
    rng = np.random.default_rng(5)
    B = 2
    co = voxel_coords(rng, B, N, r, "surface")
    feat = rng.standard_normal((B, C, r ** 3)).astype(np.float32)
    _, inds, wgts = orc.trilinear_devoxelize_forward(r, True, co, feat)
    gy = rng.standard_normal((B, C, N)).astype(np.float32)
    o_gx = orc.trilinear_devoxelize_backward(gy, inds, wgts, r)
    gx = bk.trilinear_devoxelize_backward(dev(gy), dev(inds), dev(wgts), r)
    np.testing.assert_allclose(host(gx), o_gx, rtol=TOL, atol=TOL)


BQ_CASES = [(1024, 2048, 0.1, 32), (256, 1024, 0.2, 32), (64, 256, 0.4, 32), (16, 64, 0.8, 32),
            (10, 3000, 0.3, 16), (5, 7, 10.0, 4), (33, 100, 1e-6, 8)]


@pytest.mark.parametrize("M,N,radius,U", BQ_CASES)
def test_ball_query_bit_exact(bk, orc, M, N, radius, U):
    rng = np.random.default_rng(M + N)
    B = 3
    pts = (gaussian_cloud(rng, B, N) * 0.35).astype(np.float32)
    ctr = pts[:, :, :M].copy() if M <= N else (gaussian_cloud(rng, B, M) * 0.35)
    o = orc.ball_query(ctr, pts, radius, U)
    got = bk.ball_query(dev(ctr), dev(pts), radius, U)
    assert np.array_equal(host(got), o)


@pytest.mark.parametrize("C,N,M,U", [(35, 2048, 1024, 32), (67, 1024, 256, 32), (3, 64, 16, 32),
                                     (5, 100, 7, 3)])
def test_grouping_forward_backward(bk, orc, C, N, M, U):
    rng = np.random.default_rng(C)
    B = 2
    feat = rng.standard_normal((B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, (B, M, U)).astype(np.int32)
    o = orc.grouping_forward(feat, idx)
    got = bk.grouping_forward(dev(feat), dev(idx))
    assert np.array_equal(host(got), o)
    gy = rng.standard_normal((B, C, M, U)).astype(np.float32)
    gx = host(bk.grouping_backward(dev(gy), dev(idx), N))
    # K8 (grouping.cu:58-77): scatter-add whose order the reference leaves to atomicAdd; pinned at north_star's
    # 1e-5 against the sum carried out in float64, and checked against the oracle's ascending-order float sum
    ref = np.zeros((B, C, N), np.float64)
    for b in range(B):
        np.add.at(ref[b], (slice(None), idx[b].reshape(-1)), gy[b].reshape(C, -1).astype(np.float64))
    np.testing.assert_allclose(gx, ref, rtol=TOL, atol=TOL)
    np.testing.assert_allclose(orc.grouping_backward(gy, idx, N), ref, rtol=TOL, atol=TOL)


@pytest.mark.parametrize("C,N,M,U", [(32, 2048, 1024, 32), (0, 256, 64, 32), (5, 100, 7, 3)])
def test_group_points_equals_grouping_sub_cat(bk, orc, C, N, M, U):
    """lion_group_points_forward == cat([grouping(coords) - centers, grouping(feat)]) of BallQuery.forward
    (pvcnn2_ada.py:98-114), bit for bit (the subtraction is one IEEE operation either way)."""
    from lion_amd import fused_ops as fo
    rng = np.random.default_rng(C + N)
    B = 2
    co = gaussian_cloud(rng, B, N)
    ctr = np.ascontiguousarray(co[:, :, :M])
    idx = rng.integers(0, N, (B, M, U)).astype(np.int32)
    feat = rng.standard_normal((B, C, N)).astype(np.float32) if C else None
    want = orc.grouping_forward(co, idx) - ctr[:, :, :, None]
    if C:
        want = np.concatenate([want, orc.grouping_forward(feat, idx)], 1)
    got = fo.group_points(dev(co), dev(ctr), dev(feat) if C else None, dev(idx))
    assert np.array_equal(host(got), want.astype(np.float32))


@pytest.mark.parametrize("N,M", [(2048, 1024), (1024, 256), (256, 64), (64, 16), (700, 33),
                                 (4096, 50), (5000, 20), (3, 3), (10, 1)])
def test_furthest_point_sampling_bit_exact(bk, orc, N, M):
    rng = np.random.default_rng(N * 3 + M)
    B = 3
    co = gaussian_cloud(rng, B, N)
    o = orc.furthest_point_sampling(co, M)
    got = bk.furthest_point_sampling(dev(co), M)
    assert np.array_equal(host(got), o)


def test_furthest_point_sampling_ties(bk, orc):
    """Duplicated points and a lattice: exact ties must resolve like the reference's
    512-thread reduction (lowest (k mod 512, k))."""
    rng = np.random.default_rng(0)
    B, N, M = 2, 1536, 200
    g = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij"), 0).reshape(3, -1)
    co = np.concatenate([g, g, g], 1)[None].repeat(B, 0).astype(np.float32)  # 3 copies of a lattice
    perm = rng.permutation(N)
    co = np.ascontiguousarray(co[:, :, perm])
    o = orc.furthest_point_sampling(co, M)
    got = bk.furthest_point_sampling(dev(co), M)
    assert np.array_equal(host(got), o)


def test_gather_forward_backward(bk, orc):
    rng = np.random.default_rng(3)
    B, C, N, M = 3, 3, 2048, 1024
    feat = rng.standard_normal((B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, (B, M)).astype(np.int32)
    assert np.array_equal(host(bk.gather_features_forward(dev(feat), dev(idx))),
                          orc.gather_features_forward(feat, idx))
    gy = rng.standard_normal((B, C, M)).astype(np.float32)
    np.testing.assert_allclose(host(bk.gather_features_backward(dev(gy), dev(idx), N)),
                               orc.gather_features_backward(gy, idx, N), rtol=TOL, atol=TOL)


@pytest.mark.parametrize("N,M", [(1000, 343), (257, 27), (64, 8), (33, 5)])
def test_three_nn_exact_ties_follow_the_scan_order(bk, orc, N, M):
    """Round 5: four lanes share a point's scan and their lists are merged by (distance, index).  Points and centres on an
    integer lattice (duplicated centres included) make most of the three nearest distances exact ties: indices and
    weights must still be the reference's -- the lowest index wins (neighbor_interpolate.cu:45-59, strict '<')."""
    rng = np.random.default_rng(N + M)
    B = 3
    side = max(2, int(round(M ** (1 / 3))))
    ctr = rng.integers(0, side, (B, 3, M)).astype(np.float32)           # many coincident centres
    pts = rng.integers(-1, side + 1, (B, 3, N)).astype(np.float32)
    pts[:, :, ::3] += 0.5                                                # a third of the points sit between lattice sites
    cf = rng.standard_normal((B, 5, M)).astype(np.float32)
    o_out, o_idx, o_w = orc.three_nn_interpolate_forward(pts, ctr, cf)
    out, idx, w = bk.three_nearest_neighbors_interpolate_forward(dev(pts), dev(ctr), dev(cf))
    assert np.array_equal(host(idx), o_idx)
    assert np.array_equal(host(w), o_w)
    assert np.array_equal(host(out), o_out)


@pytest.mark.parametrize("C,N,M", [(192, 2048, 1024), (192, 1024, 256), (128, 256, 64),
                                   (128, 64, 16), (4, 50, 2), (4, 50, 1), (9, 3000, 2500)])
def test_three_nn_interpolate(bk, orc, C, N, M):
    rng = np.random.default_rng(C + M)
    B = 2
    pts = gaussian_cloud(rng, B, N)
    ctr = pts[:, :, :M].copy() if M <= N else gaussian_cloud(rng, B, M)
    cf = rng.standard_normal((B, C, M)).astype(np.float32)
    o_out, o_idx, o_w = orc.three_nn_interpolate_forward(pts, ctr, cf)
    out, idx, w = bk.three_nearest_neighbors_interpolate_forward(dev(pts), dev(ctr), dev(cf))
    assert np.array_equal(host(idx), o_idx)
    assert np.array_equal(host(w), o_w)
    assert np.array_equal(host(out), o_out)
    gy = rng.standard_normal((B, C, N)).astype(np.float32)
    gx = host(bk.three_nearest_neighbors_interpolate_backward(dev(gy), dev(o_idx), dev(o_w), M))
    # K12 grad (neighbor_interpolate.cu:145-170): 3-tap scatter-add, float64 reference sum, 1e-5
    ref = np.zeros((B, C, M), np.float64)
    for b in range(B):
        for k in range(3):
            np.add.at(ref[b], (slice(None), o_idx[b, k]), gy[b].astype(np.float64) * o_w[b, k].astype(np.float64))
    np.testing.assert_allclose(gx, ref, rtol=TOL, atol=TOL)
    np.testing.assert_allclose(orc.three_nn_interpolate_backward(gy, o_idx, o_w, M), ref, rtol=TOL, atol=TOL)


@pytest.mark.parametrize("B,N,M", [(4, 100, 200), (2, 2048, 2048), (3, 513, 1025), (1, 1, 5)])
def test_chamfer_forward_backward(orc, B, N, M):
    """Reference acceptance test shape (unit_test.py:14-35: [4,100,3] vs [4,200,3]) + LION's 2048."""
    from lion_amd.chamfer3d import chamfer_3D
    rng = np.random.default_rng(N + M)
    x1 = rng.random((B, N, 3)).astype(np.float32)
    x2 = rng.random((B, M, 3)).astype(np.float32)
    o_d1, o_d2, o_i1, o_i2 = orc.chamfer_forward(x1, x2)
    d1 = torch.empty(B, N).cuda(); d2 = torch.empty(B, M).cuda()
    i1 = torch.empty(B, N, dtype=torch.int32).cuda(); i2 = torch.empty(B, M, dtype=torch.int32).cuda()
    chamfer_3D.forward(dev(x1), dev(x2), d1, d2, i1, i2)
    assert np.array_equal(host(i1), o_i1) and np.array_equal(host(i2), o_i2)
    assert np.array_equal(host(d1), o_d1) and np.array_equal(host(d2), o_d2)
    g1 = rng.standard_normal((B, N)).astype(np.float32)
    g2 = rng.standard_normal((B, M)).astype(np.float32)
    o_g1, o_g2 = orc.chamfer_backward(x1, x2, g1, g2, o_i1, o_i2)
    gx1 = torch.empty(B, N, 3).cuda(); gx2 = torch.empty(B, M, 3).cuda()
    chamfer_3D.backward(dev(x1), dev(x2), gx1, gx2, dev(g1), dev(g2), i1, i2)
    # gradient (chamfer3D.cu:155-185): own term + atomically scattered terms; float64 reference sum, 1e-5
    def ref_grad(xa, xb, ga, gb, ia, ib):
        out = np.zeros(xa.shape, np.float64)
        for b in range(B):
            da = xa[b].astype(np.float64) - xb[b][ia[b]].astype(np.float64)
            out[b] += 2.0 * ga[b][:, None].astype(np.float64) * da
            db = xb[b].astype(np.float64) - xa[b][ib[b]].astype(np.float64)
            np.add.at(out[b], ib[b], -2.0 * gb[b][:, None].astype(np.float64) * db)
        return out
    r1, r2 = ref_grad(x1, x2, g1, g2, o_i1, o_i2), ref_grad(x2, x1, g2, g1, o_i2, o_i1)
    np.testing.assert_allclose(host(gx1), r1, rtol=TOL, atol=TOL)
    np.testing.assert_allclose(host(gx2), r2, rtol=TOL, atol=TOL)
    np.testing.assert_allclose(o_g1, r1, rtol=TOL, atol=TOL)
    np.testing.assert_allclose(o_g2, r2, rtol=TOL, atol=TOL)


@pytest.mark.parametrize("B,N,M", [(3, 256, 256), (2, 300, 200), (2, 128, 512)])
def test_emd_match_and_cost(orc, B, N, M):
    from lion_amd.emd import emd_ext
    rng = np.random.default_rng(N)
    x1 = rng.random((B, N, 3)).astype(np.float32)
    x2 = rng.random((B, M, 3)).astype(np.float32)
    o_match = orc.approxmatch(x1, x2)
    o_cost = orc.matchcost(x1, x2, o_match)
    match = emd_ext.approxmatch_forward(dev(x1), dev(x2))
    cost = emd_ext.matchcost_forward(dev(x1), dev(x2), match)
    # SURVEY.md 7 risk 7: fast exp -> tolerance on the cost (1e-4 rel), looser on match entries
    np.testing.assert_allclose(host(cost), o_cost, rtol=1e-4)
    np.testing.assert_allclose(host(match), o_match, rtol=2e-3, atol=1e-5)
    # matchcost on the SAME match must agree tightly (only the sum order differs)
    cost2 = emd_ext.matchcost_forward(dev(x1), dev(x2), dev(o_match))
    np.testing.assert_allclose(host(cost2), o_cost, rtol=1e-5)
    gc = rng.standard_normal((B,)).astype(np.float32)
    o_g1, o_g2 = orc.matchcost_backward(gc, x1, x2, o_match)
    g1, g2 = emd_ext.matchcost_backward(dev(gc), dev(x1), dev(x2), dev(o_match))
    np.testing.assert_allclose(host(g1), o_g1, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(host(g2), o_g2, rtol=1e-4, atol=1e-5)


def test_emd_two_point_known_answer():
    """third_party/PyTorchEMD/test_emd_loss.py:7-19: crossed matching of 2 points."""
    from lion_amd.emd import earth_mover_distance
    p1 = torch.tensor([[[1.7, -0.1, 0.1], [0.1, 1.2, 0.3]]]).repeat(3, 1, 1).cuda()
    p2 = torch.tensor([[[0.3, 1.8, 0.2], [1.2, -0.2, 0.3]]]).repeat(3, 1, 1).cuda()
    p1.requires_grad_(True); p2.requires_grad_(True)
    d = earth_mover_distance(p1, p2, transpose=False)
    gt = (((p1[0, 0] - p2[0, 1]) ** 2).sum() + ((p1[0, 1] - p2[0, 0]) ** 2).sum()) / 2
    np.testing.assert_allclose(host(d), np.full(3, gt.item(), np.float32), rtol=1e-4)
    loss = d[0] / 2 + d[1] * 2 + d[2] / 3
    loss.backward()
    q1 = p1.detach().clone().requires_grad_(True); q2 = p2.detach().clone().requires_grad_(True)
    gt_loss = sum(w * ((((q1[i, 0] - q2[i, 1]) ** 2).sum() + ((q1[i, 1] - q2[i, 0]) ** 2).sum()) / 2)
                  for i, w in enumerate([0.5, 2.0, 1 / 3]))
    gt_loss.backward()
    np.testing.assert_allclose(host(p1.grad), host(q1.grad), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(host(p2.grad), host(q2.grad), rtol=1e-3, atol=1e-4)


def test_ddim_ddpm_update_bit_exact(orc):
    from lion_amd.diffusion_ops import ddim_update, ddpm_update
    rng = np.random.default_rng(9)
    for numel in [32 * 8192, 32 * 128, 1001]:
        x, e, z = (rng.standard_normal(numel).astype(np.float32) for _ in range(3))
        s, c, sg = np.float32(1.0001), np.float32(-0.0123), np.float32(0.0456)
        assert np.array_equal(host(ddim_update(dev(x), dev(e), dev(z), s, c, sg)),
                              orc.ddim_update(x, e, z, s, c, sg))
        assert np.array_equal(host(ddim_update(dev(x), dev(e), None, s, c, 0.0)),
                              orc.ddim_update(x, e, np.zeros_like(x), s, c, 0.0))
        args = (np.float32(1.00005), np.float32(1e-4), np.float32(0.83), np.float32(0.01), 1.0)
        assert np.array_equal(host(ddpm_update(dev(x), dev(e), dev(z), False, *args)),
                              orc.ddpm_update(x, e, z, False, *args))
        assert np.array_equal(host(ddpm_update(dev(x), dev(e), None, True, *args)),
                              orc.ddpm_update(x, e, z, True, *args))


def test_errors_are_exceptions(bk):
    with pytest.raises(RuntimeError):
        bk.avg_voxelize_forward(torch.zeros(1, 2, 3), torch.zeros(1, 3, 3, dtype=torch.int32), 4)  # CPU tensor
    with pytest.raises(RuntimeError):
        bk.ball_query(torch.zeros(1, 3, 4, dtype=torch.float64).cuda(), torch.zeros(1, 3, 4).cuda(), 0.1, 4)


def test_full_size_properties(bk):
    """BASELINE config 2 sizes (B=32, N=2048): size-independent properties, no oracle."""
    B, C, N, r = 32, 64, 2048, 32
    g = torch.Generator(device="cuda").manual_seed(0)
    co = torch.randn(B, 3, N, device="cuda", generator=g)
    feat = torch.randn(B, C, N, device="cuda", generator=g)
    out, nc, ind, cnt = bk.voxelize_points_forward(feat, co, r, True, 0.0)
    # counts sum to N per cloud; ind consistent with norm coords; mean-pool conserves the sum of means
    assert torch.equal(cnt.sum(1), torch.full((B,), N, device="cuda", dtype=cnt.dtype))
    vc = torch.round(nc).int()
    assert torch.equal(ind, vc[:, 0] * r * r + vc[:, 1] * r + vc[:, 2])
    dense = torch.zeros(B, C, r ** 3, device="cuda")
    dense.scatter_add_(2, ind.long().unsqueeze(1).expand(-1, C, -1), feat)
    dense = dense / cnt.clamp(min=1).unsqueeze(1)
    assert torch.allclose(out, dense, atol=1e-5, rtol=1e-5)
    # devoxelize(voxelize(const)) == const wherever all 8 corners are occupied is too strict;
    # use linearity instead: devox(a*f + g) == a*devox(f) + devox(g)
    f1 = torch.randn(B, 8, r ** 3, device="cuda", generator=g)
    f2 = torch.randn(B, 8, r ** 3, device="cuda", generator=g)
    d1, _, _ = bk.trilinear_devoxelize_forward(r, False, nc, f1)
    d2, _, _ = bk.trilinear_devoxelize_forward(r, False, nc, f2)
    d3, _, _ = bk.trilinear_devoxelize_forward(r, False, nc, (2.0 * f1 + f2).contiguous())
    assert torch.allclose(d3, 2.0 * d1 + d2, atol=1e-4, rtol=1e-4)
    # a constant grid devoxelizes to the constant (weights sum to 1)
    ones = torch.full((B, 2, r ** 3), 3.0, device="cuda")
    dc, _, _ = bk.trilinear_devoxelize_forward(r, False, nc, ones)
    assert torch.allclose(dc, torch.full_like(dc, 3.0), atol=1e-5)
    # FPS: indices unique per cloud, first is 0; ball query: every listed neighbour is in range
    idx = bk.furthest_point_sampling(co.contiguous(), 1024)
    assert (idx[:, 0] == 0).all()
    assert all(len(torch.unique(idx[b])) == 1024 for b in range(0, B, 8))
    ctr = torch.gather(co, 2, idx.long().unsqueeze(1).expand(-1, 3, -1)).contiguous()
    nb = bk.ball_query(ctr, co.contiguous(), 0.5, 32)
    d = (torch.gather(co, 2, nb.view(B, 1, -1).long().expand(-1, 3, -1)).view(B, 3, 1024, 32)
         - ctr.unsqueeze(-1)).pow(2).sum(1)
    assert (d < 0.25 + 1e-6).all()  # centre itself is always a hit, so no empty rows


@pytest.mark.parametrize("cin,cout,r", [(64, 64, 32), (4, 32, 32), (128, 64, 16), (192, 128, 8), (33, 32, 8)])
def test_conv3d_mfma_matches_fp64_reference(cin, cout, r, conv_kernel):
    """C3: fp32-MFMA implicit-GEMM Conv3d vs an fp64 convolution of the same fp32 operands: the
    error must be of fp32-roundoff class (each MFMA is an fmaf chain), forward and backward."""
    from lion_amd.conv_ops import conv3d_k3, conv3d_module
    torch.manual_seed(cin + cout + r)
    B = 2
    conv = torch.nn.Conv3d(cin, cout, 3, padding=1).cuda()
    x = torch.randn(B, cin, r, r, r, device="cuda")
    with torch.no_grad():
        ref = torch.nn.functional.conv3d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
        got = conv3d_k3(x, conv.weight, conv.bias)
        lib = conv(x)
    scale = ref.abs().max().item()
    err = (got.double() - ref).abs().max().item() / scale
    err_lib = (lib.double() - ref).abs().max().item() / scale
    assert err < 5e-6, (err, err_lib)       # measured ~1e-6; MIOpen's own fp32 kernel is in the same class
    # boundary voxels (zero padding) are the classic place to be wrong: check faces explicitly
    assert torch.allclose(got[..., 0, :, :].double(), ref[..., 0, :, :], rtol=1e-4, atol=1e-5 * scale)
    assert torch.allclose(got[..., :, :, -1].double(), ref[..., :, :, -1], rtol=1e-4, atol=1e-5 * scale)
    # autograd path
    x2 = x.clone().requires_grad_(True)
    y = conv3d_module(conv, x2)
    y.square().sum().backward()
    gw, gx = conv.weight.grad.clone(), x2.grad.clone()
    conv.weight.grad = None
    x3 = x.clone().requires_grad_(True)
    conv(x3).square().sum().backward()
    assert torch.allclose(gx, x3.grad, rtol=1e-3, atol=1e-3 * x3.grad.abs().max().item())
    assert torch.allclose(gw, conv.weight.grad, rtol=1e-3, atol=1e-3 * gw.abs().max().item())


def _swish64(t):
    return t * torch.sigmoid(t)


@pytest.mark.parametrize("cin,cout,L,pro", [(35, 32, 4096, False), (32, 64, 4096, True), (67, 128, 1000, True),
                                            (131, 128, 333, False), (64, 256, 2048, True),
                                            # > 4096 columns: the LDS-weight kernel for the large activations
                                            (35, 32, 9000, False), (32, 64, 20000, True), (67, 128, 8192, True),
                                            (64, 256, 5000, True)])
def test_pwconv_matches_fp64_reference(cin, cout, L, pro):
    """G1: 1x1 conv on the fp32-MFMA kernel (lion_pwconv_forward) with the AdaGN+Swish prologue and the
    GroupNorm tile sums vs an fp64 evaluation of the same expression (odd Cin, ragged L, masked tail)."""
    from lion_amd import _lib
    from lion_amd import fused_ops as fo
    torch.manual_seed(cin + L)
    B = 3
    conv = torch.nn.Conv1d(cin, cout, 1).cuda()
    x = torch.randn(B, cin, L, device="cuda")
    A = torch.randn(B, cin, device="cuda") * 0.5 + 1.0
    Bs = torch.randn(B, cin, device="cuda") * 0.3
    xin = x.double()
    if pro:
        xin = _swish64(xin * A.double()[:, :, None] + Bs.double()[:, :, None])
    ref = torch.einsum("oc,bcl->bol", conv.weight.double()[:, :, 0], xin) + conv.bias.double()[None, :, None]
    y, st = fo.pwconv_fused(x, conv, (A, Bs) if pro else None, split=False)   # the split kernel: test_pwconv_split_gpu.py
    scale = ref.abs().max().item()
    assert (y.double() - ref).abs().max().item() / scale < 1e-5
    sums = st.double().sum(2)  # [B, Cout, 2] over the column tiles
    assert torch.allclose(sums[..., 0], ref.sum(-1), rtol=1e-4, atol=1e-4 * scale * L ** 0.5)
    assert torch.allclose(sums[..., 1], ref.square().sum(-1), rtol=1e-4)
    assert _lib.load().lion_pwconv_stat_tiles(cout, cin, L) == st.shape[2]


@pytest.mark.parametrize("cin,cout,L", [(128, 4, 2048), (64, 384, 1024), (131, 48, 16), (320, 256, 300), (64, 256, 1),
                                        (259, 96, 77)])
def test_pwconv_any_cout_and_short_rows(cin, cout, L):
    """G1 for the short activations / classifier / attention projections: any Cout (rows padded to 32 inside the
    kernel, never stored), rows as short as one column, weight tiles chosen to fit LDS (320 -> 256: 64-row tiles)."""
    from lion_amd import fused_ops as fo
    torch.manual_seed(cin + cout + L)
    B = 3
    conv = torch.nn.Conv1d(cin, cout, 1).cuda()
    x = torch.randn(B, cin, L, device="cuda")
    assert fo.pw_supported(conv, x)
    ref = torch.einsum("oc,bcl->bol", conv.weight.double()[:, :, 0], x.double()) + conv.bias.double()[None, :, None]
    with torch.no_grad():
        y, st = fo.pwconv_fused(x, conv, None, split=False)
        y2, st2 = fo.pwconv_fused(x, conv, None, want_stats=False, split=False)
    assert st2 is None and torch.equal(y, y2) and tuple(y.shape) == (B, cout, L)
    scale = ref.abs().max().item()
    assert (y.double() - ref).abs().max().item() / scale < 1e-5
    sums = st.double().sum(2)
    assert torch.allclose(sums[..., 0], ref.sum(-1), rtol=1e-4, atol=1e-4 * scale * max(L, 1) ** 0.5)
    assert torch.allclose(sums[..., 1], ref.square().sum(-1), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("B,K,O,act", [(32, 128, 6144, 0), (32, 64, 64, 2), (5, 64, 64, 0), (40, 129, 37, 1), (1, 512, 128, 0)])
def test_linear_rows_matches_fp64_reference(B, K, O, act):
    """nn.Linear on own MFMA kernel (time-embedding MLP, batched AdaGN projections): odd K / O, batch tails, slabs."""
    from lion_amd import fused_ops as fo
    torch.manual_seed(B + K + O)
    lin = torch.nn.Linear(K, O).cuda()
    x = torch.randn(B, K, device="cuda")
    ref = x.double() @ lin.weight.double().t() + lin.bias.double()
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = torch.nn.functional.leaky_relu(ref, 0.1)
    with torch.no_grad():
        got = fo.linear_rows(x, lin.weight, lin.bias, act, 0.1)
    assert (got.double() - ref).abs().max().item() / ref.abs().max().item() < 1e-5


@pytest.mark.parametrize("B,C,N", [(3, 64, 1024), (2, 128, 16), (2, 64, 100), (1, 32, 2048)])
def test_linear_attention_matches_fp64_reference(B, C, N):
    """P5 (models/pvcnn2_ada.py:43-71): the module's inference path -- to_qkv on the MFMA 1x1 conv, the attention
    core of csrc/attention.hip, to_out -- against the reference's expressions evaluated in float64, and against the
    module's own torch path (grad mode)."""
    from lion_amd.models.pvcnn2_ada import LinearAttention
    torch.manual_seed(B + C + N)
    att = LinearAttention(C).cuda().eval()
    x = torch.randn(B, C, N, device="cuda")
    wq, wo, bo = att.to_qkv.weight.double()[:, :, 0, 0], att.to_out.weight.double()[:, :, 0, 0], att.to_out.bias.double()
    qkv = torch.einsum("oc,bcn->bon", wq, x.double()).view(B, 3, att.heads, 32, N)
    q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(B, att.heads * 32, N)
    ref = torch.einsum("oc,bcn->bon", wo, out) + bo[None, :, None]
    with torch.no_grad():
        got = att(x)
    assert tuple(got.shape) == (B, C, N)
    assert (got.double() - ref).abs().max().item() / ref.abs().max().item() < 1e-5
    with torch.enable_grad():
        lib_path = att(x.clone().requires_grad_(True))     # grad mode: the torch formulation
    assert (lib_path.double() - ref).abs().max().item() / ref.abs().max().item() < 1e-4


@pytest.mark.parametrize("cin,cmid,cout,B", [(2048, 2048, 256, 32), (128, 2048, 2048, 5), (256, 128, 64, 40)])
def test_skinny_gemm_chain_matches_fp64_reference(cin, cmid, cout, B):
    """D2: split-K 32-row GEMMs on channel-major activations with the deferred epilogue (lion_skinny_gemm /
    lion_skinny_finish): conv(x + t), then conv(relu(. + b1)) consuming the raw partials, the plain finish and the
    squeeze-excite tail; batch padding (B = 5) and two slabs (B = 40)."""
    from lion_amd import fused_ops as fo
    torch.manual_seed(cin + B)
    c1 = torch.nn.Conv2d(cin, cmid, 1).cuda()
    c2 = torch.nn.Conv2d(cmid, cout, 1).cuda()
    x = torch.randn(B, cin, 1, 1, device="cuda")
    t = torch.randn(B, cin, 1, 1, device="cuda")
    w1, b1 = c1.weight.double()[:, :, 0, 0], c1.bias.double()
    w2, b2 = c2.weight.double()[:, :, 0, 0], c2.bias.double()
    h1 = (x + t).double()[:, :, 0, 0] @ w1.t() + b1
    ref = torch.relu(h1) @ w2.t() + b2
    p1 = fo.skinny_conv(fo.to_channel_major(x), c1, add=fo.to_channel_major(t))
    got1 = fo.from_channel_major(fo.skinny_finish(p1, c1.bias.detach()), B)
    assert (got1.double()[:, :, 0, 0] - h1).abs().max().item() / h1.abs().max().item() < 1e-5
    p2 = fo.skinny_conv(p1, c2, bias_in=c1.bias.detach(), act_in=1)
    got = fo.from_channel_major(fo.skinny_finish(p2, c2.bias.detach()), B)
    assert (got.double()[:, :, 0, 0] - ref).abs().max().item() / ref.abs().max().item() < 1e-5
    res = torch.randn(B, cout, 1, 1, device="cuda")
    gate = torch.randn(1, *p2.shape[1:], device="cuda")  # a one-split "partial" as the gate pre-activation
    gate_b = fo.from_channel_major(gate[0], B).double()[:, :, 0, 0]
    ref2 = res.double()[:, :, 0, 0] + torch.relu(ref) * torch.sigmoid(gate_b)
    got2 = fo.from_channel_major(fo.skinny_finish(p2, c2.bias.detach(), gate, fo.to_channel_major(res)), B)
    assert (got2.double()[:, :, 0, 0] - ref2).abs().max().item() / ref2.abs().max().item() < 1e-5


def test_se_gate_and_groupnorm_fold_match_torch():
    """lion_groupnorm_fold (strided fac/gbias, tile sums) + lion_se_gate vs GroupNorm / SE3d of torch."""
    from lion_amd import fused_ops as fo
    from lion_amd.models.pvcnn2_ada import SE3d
    torch.manual_seed(3)
    B, C, T, V = 4, 64, 16, 4096
    y = torch.randn(B, C, V, device="cuda") * 2 + 0.5
    tiles = y.view(B, C, T, V // T)
    stats = torch.stack([tiles.sum(-1), tiles.square().sum(-1)], -1).contiguous()
    gn = torch.nn.GroupNorm(8, C).cuda()
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.uniform_(-0.5, 0.5)
    e = torch.randn(B, 2 * C, device="cuda")
    fac, gb = e.chunk(2, 1)  # strided views: consumed in place
    with torch.no_grad():
        A, Bs, cm = fo.groupnorm_fold(stats, gn, fac, gb, V)
        ref = gn(y) * fac[:, :, None] + gb[:, :, None]
        got = y * A[:, :, None] + Bs[:, :, None]
        assert torch.allclose(got, ref, rtol=1e-4, atol=1e-4)
        assert torch.allclose(cm, y.mean(-1), rtol=1e-5, atol=1e-6)
        se = SE3d(C).cuda()
        ref_gate = se.fc(ref.mean(-1))
        A2, B2 = fo.se_gate_(A.clone(), Bs.clone(), cm, se)
        assert torch.allclose(A2, A * ref_gate, rtol=1e-4, atol=1e-5)
        assert torch.allclose(B2, Bs * ref_gate, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("cin,cout,r,n", [(32, 32, 32, 2048), (64, 64, 32, 2048), (128, 128, 16, 512)])
def test_conv3d_empty_tile_skip_is_bit_identical(cin, cout, r, n, conv_kernel):
    """conv1 of a PVConv on the voxelised (sparse) grid: skipping the K loop of tiles whose halo holds no
    point (lion_conv3d_tile_occupancy) gives bit-identical outputs and GroupNorm sums; a flat cloud
    leaves most tiles empty."""
    from lion_amd import fused_ops as fo
    torch.manual_seed(r + cin)
    B = 3
    coords = torch.randn(B, 3, n, device="cuda") * torch.tensor([1.0, 0.2, 0.6], device="cuda").view(1, 3, 1)
    feat = torch.randn(B, cin, n, device="cuda")
    out, _, _, cnt = bk_().voxelize_points_forward(feat, coords, r, True, 0.0)
    grid = out.view(B, cin, r, r, r)
    conv = torch.nn.Conv3d(cin, cout, 3, padding=1).cuda()
    with torch.no_grad():
        y0, s0 = fo.conv3d_fused(grid, conv, None, True, None)
        occ1, _ = fo.conv3d_occupancy(cnt, r, cout, B)
        y1, s1 = fo.conv3d_fused(grid, conv, None, True, occ1)
        nt = occ1.numel() // 2 // B
        assert ((occ1[:B * nt] & 0xf) != 0).float().mean().item() < 0.9  # the flat cloud leaves tiles empty
    assert torch.equal(y0, y1)  # same K order per voxel whatever the tiling
    t0, t1 = s0.sum(2), s1.sum(2)  # the sparse launch tiles differently: compare the totals
    assert torch.allclose(t0, t1, rtol=1e-4, atol=1e-5 * t0.abs().max().item())


def bk_():
    from lion_amd.functional.backend import _backend
    return _backend


@pytest.mark.parametrize("c,r,n,flat", [(32, 32, 2048, True), (64, 32, 2048, False), (64, 16, 700, True)])
def test_conv3d_constant_plus_delta_matches_dense(c, r, n, flat, conv_kernel):
    """second conv of a PVConv: swish(AdaGN(conv1)) = per-channel constant + sparse delta.  The delta-mode kernel
    (constant response per border configuration in the epilogue, tiles with no point within 2 voxels skipped) must
    agree with the dense evaluation of the same convolution to fp32 rounding, including faces / edges / corners."""
    from lion_amd import fused_ops as fo
    torch.manual_seed(r + c + n)
    B = 3
    sc = torch.tensor([1.0, 0.2, 0.6] if flat else [1.0, 1.0, 1.0], device="cuda").view(1, 3, 1)
    coords = torch.randn(B, 3, n, device="cuda") * sc
    feat = torch.randn(B, c, n, device="cuda")
    out, _, _, cnt = bk_().voxelize_points_forward(feat, coords, r, True, 0.0)
    grid = out.view(B, c, r, r, r)
    conv1 = torch.nn.Conv3d(c, c, 3, padding=1).cuda()
    conv2 = torch.nn.Conv3d(c, c, 3, padding=1).cuda()
    A = torch.rand(B, c, device="cuda") + 0.5
    Bs = torch.randn(B, c, device="cuda") * 0.5
    with torch.no_grad():
        occ1, occ2 = fo.conv3d_occupancy(cnt, r, c, B)
        y1, _ = fo.conv3d_fused(grid, conv1, None, True, occ1)
        dense, sd = fo.conv3d_fused(y1, conv2, (A, Bs), True, None)
        sparse, ss = fo.conv3d_fused(y1, conv2, (A, Bs), True, occ2, prev_conv=conv1)
        ref = torch.nn.functional.conv3d(
            torch.nn.functional.silu(y1.double() * A.double().view(B, c, 1, 1, 1) + Bs.double().view(B, c, 1, 1, 1)),
            conv2.weight.double(), conv2.bias.double(), padding=1)
    scale = ref.abs().max().item()
    assert (dense.double() - ref).abs().max().item() / scale < 1e-5
    assert (sparse.double() - ref).abs().max().item() / scale < 1e-5
    for sl in ((..., 0, 0, 0), (..., r - 1, r - 1, r - 1), (..., 0, r // 2, r - 1)):  # corners / edge
        assert torch.allclose(sparse[sl].double(), ref[sl], rtol=1e-4, atol=1e-5 * scale)
    tot_s, tot_d = ss.sum(2), sd.sum(2)
    assert torch.allclose(tot_s, tot_d, rtol=1e-3, atol=1e-4 * tot_d.abs().max().item())


def _cloud(kind, B, n, gen):
    """clouds that drive the voxel compaction of the sparse split convolution through its cases: `clumped` = what the
    latents of the chain look like after ~20 steps (95 % of the points in a tenth of the extent: a few active voxels per
    occupied tile, most waves without a block); `one-point` (a centre voxel + two corners); `two-corners` (active voxels on the grid's faces / edges /
    corners only); `full` = every voxel occupied (every tile has its 8 blocks); `gauss`"""
    if kind == "one-point":   # one point in the middle of the grid (+ the two corner points that pin the normalisation)
        c = torch.zeros(B, 3, 3, device="cuda")
        c[:, :, 1], c[:, :, 2] = 1.0, -1.0
        return c
    if kind == "two-corners":
        c = torch.ones(B, 3, 2, device="cuda")
        c[:, :, 1] = -1.0
        return c
    c = torch.randn(B, 3, n, device="cuda", generator=gen)
    if kind == "clumped":
        c[:, :, : int(0.95 * n)] *= 0.1
    if kind == "full":
        c = torch.rand(B, 3, n, device="cuda", generator=gen) * 2 - 1
        c[:, :, :8] = torch.tensor([[x, y, z] for x in (-1., 1.) for y in (-1., 1.) for z in (-1., 1.)], device="cuda").t()
    return c


@pytest.mark.parametrize("kind,c,r,n", [("clumped", 64, 32, 2048), ("clumped", 128, 16, 1024), ("one-point", 32, 32, 1),
                                        ("two-corners", 64, 16, 2), ("full", 32, 16, 60000), ("gauss", 64, 32, 2048),
                                        ("gauss", 128, 16, 1024), ("clumped", 32, 32, 2048)])
def test_conv3d_voxel_compaction_matches_dense(kind, c, r, n):
    """The sparse split convolution (work queue over occupied tiles, per-wave masks, bias / constant response everywhere else)
    on clouds from one point to a full grid.  conv1 (margin 1): outputs bit-identical to the dense launch of the same kernel;
    conv2 (constant + delta, margin 2): within 1e-5 of float64, borders included; GroupNorm tile sums equal to fp32 summation
    order in both.  (Named after round 5's voxel-compaction experiment, whose bit maps left the occupancy buffer in round 6.)"""
    from lion_amd import conv_ops, fused_ops as fo
    if not conv_ops.SPLIT:
        pytest.skip("the split kernel is switched off")
    gen = torch.Generator(device="cuda").manual_seed(r * 7 + c + n)
    B = 3
    coords = _cloud(kind, B, n, gen)
    n = coords.shape[2]
    feat = torch.randn(B, c, n, device="cuda", generator=gen)
    # `full`: coordinates in [-1, 1] taken as they are (normalize = False maps them onto the whole grid); otherwise the
    # reference's normalisation by the cloud's own extent
    out, _, _, cnt = bk_().voxelize_points_forward(feat, coords, r, kind != "full", 0.0)
    grid = out.view(B, c, r, r, r)
    conv1 = torch.nn.Conv3d(c, c, 3, padding=1).cuda()
    conv2 = torch.nn.Conv3d(c, c, 3, padding=1).cuda()
    A = torch.rand(B, c, device="cuda", generator=gen) + 0.5
    Bs = torch.randn(B, c, device="cuda", generator=gen) * 0.5
    with torch.no_grad():
        occ1, occ2 = fo.conv3d_occupancy(cnt, r, c, B)
        nt = (occ1.numel() - 4) // 2 // B
        occupied = int(((occ1[:B * nt] & 0xf) != 0).sum())          # tiles whose halo holds a point (margin 1)
        if kind == "full":
            assert occupied == B * nt
        elif kind in ("one-point", "two-corners"):
            assert 0 < occupied <= B * 3 * 4                         # each of <= 3 points touches <= 4 tiles of the (d, h) tiling
        y_d, s_d = fo.conv3d_fused(grid, conv1, None, True, None)
        y_s, s_s = fo.conv3d_fused(grid, conv1, None, True, occ1)
        assert torch.equal(y_d, y_s)
        t_d, t_s = s_d.sum(2), s_s.sum(2)
        assert torch.allclose(t_d, t_s, rtol=1e-4, atol=1e-5 * t_d.abs().max().item())
        dense, sd = fo.conv3d_fused(y_s, conv2, (A, Bs), True, None)
        sparse, ss = fo.conv3d_fused(y_s, conv2, (A, Bs), True, occ2, prev_conv=conv1)
        ref = torch.nn.functional.conv3d(
            torch.nn.functional.silu(y_s.double() * A.double().view(B, c, 1, 1, 1) + Bs.double().view(B, c, 1, 1, 1)),
            conv2.weight.double(), conv2.bias.double(), padding=1)
    scale = ref.abs().max().item()
    assert (sparse.double() - ref).abs().max().item() / scale < 1e-5
    assert (dense.double() - ref).abs().max().item() / scale < 1e-5
    tot_s, tot_d = ss.sum(2), sd.sum(2)
    assert torch.allclose(tot_s, tot_d, rtol=1e-3, atol=1e-4 * tot_d.abs().max().item())
    # run twice: the packing, the fill and the closed-form sums are deterministic
    with torch.no_grad():
        occ1b, _ = fo.conv3d_occupancy(cnt, r, c, B)
        y_s2, s_s2 = fo.conv3d_fused(grid, conv1, None, True, occ1b)
    assert torch.equal(y_s, y_s2) and torch.equal(s_s, s_s2)


@pytest.mark.parametrize("r,cout,n", [(32, 64, 2048), (32, 32, 300), (16, 128, 1024)])
def test_conv3d_tile_occupancy_matches_dilation_reference(r, cout, n):
    """lion_conv3d_tile_occupancy flags (margin 1 and 2) == max-pool dilation of the count grid reduced over
    the tile footprint; each sample's work list is a permutation of its tiles with the occupied ones first."""
    from lion_amd import _lib, fused_ops as fo
    torch.manual_seed(r + n)
    B = 5
    coords = torch.randn(B, 3, n, device="cuda") * torch.tensor([1.0, 0.3, 0.7], device="cuda").view(1, 3, 1)
    _, _, _, cnt = bk_().voxelize_points_forward(torch.randn(B, 4, n, device="cuda"), coords, r, True, 0.0)
    occ1, occ2 = fo.conv3d_occupancy(cnt, r, cout, B)
    nt = _lib.load().lion_conv3d_stat_tiles(r, cout, B, 1)
    vb = 2  # sparse launches use the 2x2 MFMA tiles: (2,4,32) at r=32, (4,4,16) at r=16
    td, th = (2, 4) if r == 32 else (4, 4)
    g = (cnt.view(B, 1, r, r, r) > 0).float()
    for occ, m in ((occ1, 1), (occ2, 2)):
        d = torch.nn.functional.max_pool3d(g, 2 * m + 1, 1, m)[:, 0]
        ref = d.view(B, r // td, td, r // th, th, r).amax(dim=(2, 4, 5)).reshape(B, nt).int()
        word = occ[:B * nt].view(B, nt)
        flags = word & 0xf                # 4-bit masks: bit w = wave w's 64-voxel block sees a point; 0 = empty tile
        assert torch.equal((flags != 0).int(), ref), m
        assert int((word >> 10).max()) == 0
        # bit 9 (margin-2 words only): the tile is occupied at margin 1 -- the split delta convolution loads the first
        # convolution's output only inside such tiles (aware level 2)
        d1 = torch.nn.functional.max_pool3d(g, 3, 1, 1)[:, 0]
        occ1_ref = d1.view(B, r // td, td, r // th, th, r).amax(dim=(2, 4, 5)).reshape(B, nt).int()
        assert torch.equal((word >> 9) & 1, occ1_ref if m == 2 else torch.zeros_like(occ1_ref)), m
        # round 5, bit 8 = the tile's output has a reader: margin 2 -- the tiles around the points themselves (the
        # devoxelisation); margin 1 -- the margin-2 tiles and their neighbours in the (d, h) tile grid (the delta conv's halos)
        d2 = torch.nn.functional.max_pool3d(g, 5, 1, 2)[:, 0]
        occ2_ref = d2.view(B, r // td, td, r // th, th, r).amax(dim=(2, 4, 5))           # [B, r/td, r/th]
        need_ref = occ2_ref if m == 2 else torch.nn.functional.max_pool2d(occ2_ref[:, None], 3, 1, 1)[:, 0]
        assert torch.equal((word >> 8) & 1, need_ref.reshape(B, nt).int()), m
        assert int(occ[2 * B * nt + 2]) == 0      # the plain entry point: every voxel is written
        lst = occ[B * nt:2 * B * nt].view(B, nt)
        for b in range(B):
            assert sorted(lst[b].tolist()) == list(range(nt))
            k = int(ref[b].sum())
            assert ref[b][lst[b][:k].long()].all() and not ref[b][lst[b][k:].long()].any()
        assert int(occ[2 * B * nt]) == 0
        # (round 6: the per-tile active-voxel bit maps and per-sample counts of the rejected compaction experiment are gone)
        assert occ.numel() == 2 * B * nt + 4


@pytest.mark.parametrize("B,N,M", [(3, 512, 512), (2, 300, 1024), (4, 2048, 2048)])
def test_emd_fused_cost_matches_materialised_path(B, N, M):
    """lion_emd_cost (no [B,N,M] match matrix: row sums of w * d^2 accumulated over the levels) == matchcost of the
    materialised approxmatch, to fp32 summation-order noise."""
    from lion_amd.emd import emd_ext, earth_mover_distance_nograd
    torch.manual_seed(N + M)
    x1 = torch.rand(B, N, 3, device="cuda")
    x2 = torch.rand(B, M, 3, device="cuda")
    ref = emd_ext.matchcost_forward(x1, x2, emd_ext.approxmatch_forward(x1, x2)) / float(N)
    got = earth_mover_distance_nograd(x1, x2, transpose=False)
    assert torch.allclose(got, ref, rtol=2e-5, atol=1e-7), (got, ref)


@pytest.mark.parametrize("cin,cout,r", [(64, 64, 32), (4, 32, 32), (128, 64, 16), (192, 128, 8)])
def test_conv3d_wgrad_matches_fp64_reference(cin, cout, r):
    """weight gradient on the MFMA kernel (voxels on the k axis, fixed-order partial sums) vs an fp64 evaluation of
    sum_{b,v} gy * shifted x; also the bias gradient and the dgrad-as-forward identity used by the autograd path."""
    from lion_amd.conv_ops import conv3d_k3, conv3d_k3_wgrad, dgrad_weight
    torch.manual_seed(cin + r)
    B = 2
    x = torch.randn(B, cin, r, r, r, device="cuda")
    gy = torch.randn(B, cout, r, r, r, device="cuda")
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.1
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    torch.nn.functional.conv3d(xd, wd, None, padding=1).backward(gy.double())
    gw = conv3d_k3_wgrad(x, gy, w.shape, split=False)
    e32 = (gw.double() - wd.grad).abs().max().item() / wd.grad.abs().max().item()
    assert e32 < 2e-5
    if cin % 8 == 0:
        # round 4: the same gradient on the 16-bit pipe (fp16 pairs cut in registers, per-tensor scales): held to the fp32
        # kernel's own error (both accumulate ~10^5 products in fp32), also with operands 1e-6 / 1e4 away from unit scale
        gs = conv3d_k3_wgrad(x, gy, w.shape, split=True)
        es = (gs.double() - wd.grad).abs().max().item() / wd.grad.abs().max().item()
        assert es < max(2 * e32, 4e-6), (es, e32)
        gs2 = conv3d_k3_wgrad(x * 1e4, gy * 1e-6, w.shape, split=True)
        assert (gs2.double() - wd.grad * 1e-2).abs().max().item() / (wd.grad.abs().max().item() * 1e-2) < max(2 * e32, 4e-6)
        assert torch.equal(conv3d_k3_wgrad(x, gy, w.shape, split=True), gs)      # deterministic
    if cin % 32 == 0:  # the data gradient is a forward conv with Cin output channels
        gx = conv3d_k3(gy, dgrad_weight(w), None)
        assert (gx.double() - xd.grad).abs().max().item() / xd.grad.abs().max().item() < 1e-5


def test_conv3d_work_queue_rearms_itself(conv_kernel):
    """Round 5: a sparse convolution re-arms the queue of its occupancy buffer when its last workgroup leaves, so ONE buffer
    serves every convolution that shares its (cloud, resolution) pair (a PVCNN2 forward used to clone it 7 times).  Three
    launches from the same buffer == three launches from fresh buffers, bit for bit; queue and exit counter are zero
    after each."""
    from lion_amd import fused_ops as fo
    torch.manual_seed(9)
    B, c, r, n = 3, 32, 32, 2048
    coords = torch.randn(B, 3, n, device="cuda") * torch.tensor([1.0, 0.2, 0.6], device="cuda").view(1, 3, 1)
    feat = torch.randn(B, c, n, device="cuda")
    out, _, _, cnt = bk_().voxelize_points_forward(feat, coords, r, True, 0.0)
    grid = out.view(B, c, r, r, r)
    conv1 = torch.nn.Conv3d(c, c, 3, padding=1).cuda()
    conv2 = torch.nn.Conv3d(c, c, 3, padding=1).cuda()
    A = torch.rand(B, c, device="cuda") + 0.5
    Bs = torch.randn(B, c, device="cuda") * 0.5
    with torch.no_grad():
        occ1, occ2 = fo.conv3d_occupancy(cnt, r, c, B)
        nt = int(_lib_().lion_conv3d_stat_tiles(r, c, B, 1))
        fresh = lambda: fo.conv3d_occupancy(cnt, r, c, B)
        for _ in range(3):
            y_a, s_a = fo.conv3d_fused(grid, conv1, None, True, occ1)
            y_b, s_b = fo.conv3d_fused(grid, conv1, None, True, fresh()[0])
            assert torch.equal(y_a, y_b) and torch.equal(s_a, s_b)
            assert occ1[2 * B * nt:2 * B * nt + 2].tolist() == [0, 0]
            z_a, t_a = fo.conv3d_fused(y_a, conv2, (A, Bs), True, occ2, prev_conv=conv1)
            z_b, t_b = fo.conv3d_fused(y_a, conv2, (A, Bs), True, fresh()[1], prev_conv=conv1)
            assert torch.equal(z_a, z_b) and torch.equal(t_a, t_b)
            assert occ2[2 * B * nt:2 * B * nt + 2].tolist() == [0, 0]


def _lib_():
    from lion_amd import _lib
    return _lib.load()


@pytest.mark.parametrize("C,N,r,kind", [(64, 2048, 32, "flat"), (32, 2048, 32, "gauss"), (128, 1024, 16, "flat"), (4, 2048, 32, "clumped")])
def test_voxel_scatter_for_the_sparse_reader(C, N, r, kind):
    """Round 5, lion_voxel_scatter_read: a grid whose only reader is the sparse convolution is written where that convolution
    stages -- the z-rows within one voxel (in d and h) of a tile with a point within one voxel -- bit-identical to the full
    scatter there, and NOT written elsewhere (the buffer handed in keeps its sentinel; the full scatter holds zeros there)."""
    from lion_amd import _lib, fused_ops as fo
    gen = torch.Generator(device="cuda").manual_seed(C + N + r)
    B = 3
    co = _cloud(kind, B, N, gen)
    if kind == "flat":
        co = torch.randn(B, 3, N, device="cuda", generator=gen) * torch.tensor([1.0, 0.15, 0.6], device="cuda").view(1, 3, 1)
    feat = torch.randn(B, C, N, device="cuda", generator=gen)
    plan = bk_().voxel_index(co, r, True, 0.0)
    full = bk_().voxel_scatter(feat, plan)
    occ1, _ = fo.conv3d_occupancy(plan["cnt"], r, 64, B)
    lib = _lib.load()
    out = torch.full((B, C, r ** 3), -7.5, device="cuda")
    ws = plan["ws"]
    _lib.check(lib.lion_voxel_scatter_read(_lib.ptr(feat), _lib.ptr(ws), ws.numel(), B, C, N, r, _lib.ptr(occ1), _lib.ptr(out),
                                           _lib.stream_ptr(feat.device)), "voxel_scatter_read")
    td, th = (2, 4) if r == 32 else (4, 4)
    nt = (r // td) * (r // th)
    occ_t = ((occ1[:B * nt] & 0xf) != 0).view(B, 1, r // td, r // th).float()
    rows = occ_t.repeat_interleave(td, 2).repeat_interleave(th, 3)                      # [B, 1, r, r]: rows of occupied tiles
    need = torch.nn.functional.max_pool2d(rows, 3, 1, 1)[:, 0].bool()                   # + one row around them
    need_v = need.view(B, 1, r, r, 1).expand(B, C, r, r, r).reshape(B, C, r ** 3)
    assert torch.equal(out[need_v], full[need_v])
    assert bool((out[~need_v] == -7.5).all()) and bool((full[~need_v] == 0).all())
    assert 0.0 < need.float().mean().item() < (0.95 if kind != "gauss" else 1.01)


# ---- round 6: the layout / concatenation passes that replaced ATen kernels inside a captured step -------------------------
@pytest.mark.parametrize("B,N,D", [(3, 2048, 4), (2, 300, 4), (2, 257, 6), (1, 64, 3)])
def test_latent_unpack_equals_view_permute_slices(B, N, D):
    from lion_amd import fused_ops as fo
    torch.manual_seed(N + D)
    x = torch.randn(B, N * D, 1, 1, device="cuda")
    ref = x.view(B, N, D).permute(0, 2, 1).contiguous()
    al, co, re = fo.latent_unpack(x, N, D, True, True, D > 3)
    assert torch.equal(al, ref) and torch.equal(co, ref[:, :3].contiguous())
    if D > 3:
        assert torch.equal(re, ref[:, 3:].contiguous())
    only = fo.latent_unpack(x, N, D, False, True, False)
    assert only[0] is None and only[2] is None and torch.equal(only[1], ref[:, :3].contiguous())


@pytest.mark.parametrize("B,Ca,Ct,N,expand_batch", [(3, 64, 64, 1024, False), (2, 128, 64, 256, True), (4, 192, 64, 64, True)])
def test_concat_broadcast_equals_torch_cat(B, Ca, Ct, N, expand_batch):
    from lion_amd import fused_ops as fo
    torch.manual_seed(Ca + N)
    a = torch.randn(B, Ca, N, device="cuda")
    emb = torch.randn(1 if expand_batch else B, Ct, device="cuda")
    if expand_batch:
        emb = emb.expand(B, -1)
    temb = emb[:, :, None].expand(-1, -1, N)
    got = fo.concat_broadcast(a, temb)
    assert got is not None and torch.equal(got, torch.cat([a, temb], dim=1))
    # operands that do not qualify are refused, not mangled: a materialised (non-broadcast) embedding, N % 4 != 0
    assert fo.concat_broadcast(a, temb.contiguous()) is None
    assert fo.concat_broadcast(a[:, :, :N - 1].contiguous(), temb[:, :, :N - 1]) is None


@pytest.mark.parametrize("B,C1,C2,C3,N,M", [(3, 128, 64, 192, 256, 64), (2, 128, 64, 1, 2048, 1024), (2, 64, 0, 32, 300, 100),
                                            (2, 128, 64, 0, 64, 16)])
def test_three_nn_interpolate_cat_equals_the_composition(B, C1, C2, C3, N, M):
    """[interpolate(cat(cfeat, temb)) ; skip] in one pass == the reference's composition (pvcnn2_ada.py:403-411) through the
    plain entry point and two torch.cat, bit for bit"""
    from lion_amd import fused_ops as fo
    torch.manual_seed(C1 + N)
    pts = torch.randn(B, 3, N, device="cuda")
    ctr = pts[:, :, torch.randperm(N, device="cuda")[:M]].contiguous() if M <= N else torch.randn(B, 3, M, device="cuda")
    cf = torch.randn(B, C1, M, device="cuda")
    temb = None if C2 == 0 else torch.randn(1, C2, device="cuda").expand(B, -1)[:, :, None].expand(-1, -1, M)
    skip = None if C3 == 0 else torch.randn(B, C3, N, device="cuda")
    got = fo.three_nn_interpolate_cat(pts, ctr, cf, temb, skip)
    full = cf if temb is None else torch.cat([cf, temb], dim=1).contiguous()
    ref = bk_().three_nearest_neighbors_interpolate_forward(pts, ctr, full)[0]
    if skip is not None:
        ref = torch.cat([ref, skip], dim=1)
    assert got is not None and torch.equal(got, ref)


def test_chain_update_channel_major_eps_equals_the_transposed_update():
    """lion_chain_update_noise_cm on the denoiser's [B, 4, N] output == lion_chain_update_noise on its transposed copy: same
    Philox counters, same arithmetic, bit for bit (DDIM and DDPM rows)"""
    from lion_amd import _lib
    lib = _lib.load()
    B, N = 3, 2048
    torch.manual_seed(5)
    x = torch.randn(B, N * 4, 1, 1, device="cuda")
    eps_cm = torch.randn(B, 4, N, device="cuda")
    eps_pm = eps_cm.permute(0, 2, 1).contiguous()
    seed = torch.tensor([123, 456], dtype=torch.int32, device="cuda")
    for mode, cur in ((0, [1.0, 0.99, -0.01, 0.02, 0.0, 0.0, 0.0, 0.0]), (1, [1.0, 1.01, 0.02, 0.5, 0.1, 1.0, 0.0, 0.0])):
        c = torch.tensor(cur, device="cuda")
        c[7] = torch.tensor([7], dtype=torch.int32).view(torch.float32)[0]
        o1, z1 = torch.empty_like(x), torch.empty_like(x)
        o2, z2 = torch.empty_like(x), torch.empty_like(x)
        st = _lib.stream_ptr(x.device)
        _lib.check(lib.lion_chain_update_noise(mode, _lib.ptr(x), _lib.ptr(eps_pm), x.numel(), _lib.ptr(c), _lib.ptr(seed), 0,
                                               _lib.ptr(o1), _lib.ptr(z1), st), "update")
        _lib.check(lib.lion_chain_update_noise_cm(mode, _lib.ptr(x), _lib.ptr(eps_cm), B, N, _lib.ptr(c), _lib.ptr(seed), 0,
                                                  _lib.ptr(o2), _lib.ptr(z2), st), "update_cm")
        assert torch.equal(o1, o2) and torch.equal(z1, z2)


@pytest.mark.parametrize("B,C,Ct,bcast", [(32, 128, 2048, True), (5, 128, 64, False), (40, 16, 8, True), (1, 128, 2048, True)])
def test_channel_major_transposes_equal_the_torch_formulation(B, C, Ct, bcast):
    """lion_to_channel_major / lion_from_channel_major (round 6: the global prior's step without ATen copies) == the reshape /
    pad / transpose formulation, incl. a single time-embedding row broadcast over the batch and batches that are not multiples of 32"""
    from lion_amd import fused_ops as fo
    torch.manual_seed(B + C)
    x = torch.randn(B, C, 1, 1, device="cuda")
    t = torch.randn(1 if bcast else B, Ct, 1, 1, device="cuda")
    with torch.no_grad():
        ox, ot = fo.to_channel_major_pair(x, t)
        te = t.expand(B, -1, -1, -1) if bcast else t
        assert torch.equal(ox, fo.to_channel_major(x)) and torch.equal(ot, fo.to_channel_major(te))
        back = fo.from_channel_major(ox, B)
        assert back.shape == x.shape and torch.equal(back, x)
    with torch.enable_grad():     # the torch formulation under autograd: same values
        ox2, ot2 = fo.to_channel_major_pair(x, t)
    assert torch.equal(ox2, ox) and torch.equal(ot2, ot)


@pytest.mark.parametrize("nb_b,C", [(32, 2048), (5, 256)])
def test_skinny_se_finish_equals_gemm_then_finish(nb_b, C):
    """lion_skinny_gemm_se_finish (round 6: the block's second squeeze-excite GEMM with the block's tail in its epilogue) ==
    lion_skinny_gemm + lion_skinny_finish(mode 1), bit for bit"""
    from lion_amd import fused_ops as fo
    torch.manual_seed(C + nb_b)
    H = C // 8
    fc2 = torch.nn.Conv2d(H, C, 1, bias=False).cuda()
    nb = (nb_b + 31) // 32
    p3 = torch.randn(2, nb, H, 32, device="cuda")            # fc1's raw partials (two k-splits)
    p2 = torch.randn(4, nb, C, 32, device="cuda")            # conv2's raw partials
    b2 = torch.randn(C, device="cuda")
    h = torch.randn(nb, C, 32, device="cuda")
    with torch.no_grad():
        ref = fo.skinny_finish(p2, b2, fo.skinny_conv(p3, fc2, act_in=1), h)
        got = fo.skinny_conv_se_finish(p3, fc2, p2, b2, h)
    assert got is not None and torch.equal(got, ref)
