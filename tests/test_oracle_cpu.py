"""CPU suite: the oracle (oracle/lion_oracle.c) against INDEPENDENT pure-torch / numpy
formulations of the same operators (nothing in the reference pins K1-K12, so the
restatement is cross-checked against differently-written implementations), plus the reference's
own known-answer tests that exist (Chamfer vs a brute-force distance matrix, EMD 2-point KAT)."""
import numpy as np
import pytest
import torch

from conftest import gaussian_cloud, surface_cloud, voxel_coords


def test_voxelize_vs_index_add(orc):
    rng = np.random.default_rng(0)
    B, C, N, r = 3, 5, 700, 8
    vc = np.rint(voxel_coords(rng, B, N, r, "surface")).astype(np.int32)
    feat = rng.standard_normal((B, C, N)).astype(np.float32)
    out, ind, cnt = orc.avg_voxelize_forward(feat, vc, r)
    flat = vc[:, 0] * r * r + vc[:, 1] * r + vc[:, 2]
    assert np.array_equal(ind, flat)
    for b in range(B):
        assert np.array_equal(cnt[b], np.bincount(flat[b], minlength=r ** 3))
        ref = torch.zeros(C, r ** 3, dtype=torch.float64).index_add_(
            1, torch.from_numpy(flat[b]).long(), torch.from_numpy(feat[b]).double())
        ref = ref / torch.from_numpy(cnt[b]).clamp(min=1)
        np.testing.assert_allclose(out[b], ref.numpy(), rtol=1e-5, atol=1e-6)
    # backward == autograd of the same formulation
    gy = rng.standard_normal((B, C, r ** 3)).astype(np.float32)
    gx = orc.avg_voxelize_backward(gy, ind, cnt)
    f = torch.from_numpy(feat).double().requires_grad_(True)
    tot = 0
    for b in range(B):
        o = torch.zeros(C, r ** 3, dtype=torch.float64).index_add(1, torch.from_numpy(flat[b]).long(), f[b])
        o = o / torch.from_numpy(cnt[b]).clamp(min=1)
        tot = tot + (o * torch.from_numpy(gy[b]).double()).sum()
    tot.backward()
    np.testing.assert_allclose(gx, f.grad.numpy(), rtol=1e-5, atol=1e-6)


def _devox_torch(co, feat, r):
    """explicit 8-corner gather, written independently (floor/ceil form)."""
    B, C, _ = feat.shape
    lo = torch.floor(co)
    fr = co - lo
    lo = lo.long()
    hi = torch.where(fr > 0, lo + 1, lo)
    out = torch.zeros(B, C, co.shape[2], dtype=torch.float64)
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                ix = (hi[:, 0] if dx else lo[:, 0]) * r * r + (hi[:, 1] if dy else lo[:, 1]) * r \
                    + (hi[:, 2] if dz else lo[:, 2])
                w = (fr[:, 0] if dx else 1 - fr[:, 0]) * (fr[:, 1] if dy else 1 - fr[:, 1]) \
                    * (fr[:, 2] if dz else 1 - fr[:, 2])
                out = out + w.unsqueeze(1) * torch.gather(feat, 2, ix.unsqueeze(1).expand(-1, C, -1))
    return out


def test_devoxelize_vs_explicit_gather(orc):
    rng = np.random.default_rng(1)
    B, C, N, r = 2, 6, 500, 8
    co = voxel_coords(rng, B, N, r)
    co[:, :, :3] = np.array([[0.0, r - 1.0, 2.0]] * 3, np.float32)
    feat = rng.standard_normal((B, C, r ** 3)).astype(np.float32)
    out, inds, wgts = orc.trilinear_devoxelize_forward(r, True, co, feat)
    ref = _devox_torch(torch.from_numpy(co).double(), torch.from_numpy(feat).double(), r)
    np.testing.assert_allclose(out, ref.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(wgts.sum(1), 1.0, atol=1e-6)
    assert inds.min() >= 0 and inds.max() < r ** 3
    out2, i2, w2 = orc.trilinear_devoxelize_forward(r, False, co, feat)
    assert np.array_equal(out, out2) and i2.shape == (1,) and w2.shape == (1,)
    # backward == autograd
    gy = rng.standard_normal((B, C, N)).astype(np.float32)
    gx = orc.trilinear_devoxelize_backward(gy, inds, wgts, r)
    f = torch.from_numpy(feat).double().requires_grad_(True)
    (_devox_torch(torch.from_numpy(co).double(), f, r) * torch.from_numpy(gy).double()).sum().backward()
    np.testing.assert_allclose(gx, f.grad.numpy(), rtol=1e-5, atol=1e-5)


def test_ball_query_vs_bruteforce(orc):
    rng = np.random.default_rng(2)
    B, N, M, U, radius = 2, 300, 40, 8, 0.6
    pts = gaussian_cloud(rng, B, N)
    ctr = pts[:, :, :M].copy()
    got = orc.ball_query(ctr, pts, radius, U)
    d2 = ((ctr[:, :, :, None] - pts[:, :, None, :]) ** 2)
    d2 = (d2[:, 0] + d2[:, 1]) + d2[:, 2]  # same association as the kernel
    for b in range(B):
        for j in range(M):
            hits = np.nonzero(d2[b, j] < np.float32(radius) * np.float32(radius))[0][:U]
            exp = np.zeros(U, np.int32)
            if len(hits):
                exp[:] = hits[0]
                exp[: len(hits)] = hits
            assert np.array_equal(got[b, j], exp)
    # no hit at all -> zeros
    far = orc.ball_query(ctr + 100.0, pts, radius, U)
    assert (far == 0).all()


def test_grouping_and_gather_vs_torch(orc):
    rng = np.random.default_rng(3)
    B, C, N, M, U = 2, 4, 50, 9, 5
    feat = rng.standard_normal((B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, (B, M, U)).astype(np.int32)
    out = orc.grouping_forward(feat, idx)
    ref = np.stack([feat[b][:, idx[b]] for b in range(B)])
    assert np.array_equal(out, ref)
    gy = rng.standard_normal((B, C, M, U)).astype(np.float32)
    gx = orc.grouping_backward(gy, idx, N)
    f = torch.from_numpy(feat).double().requires_grad_(True)
    tot = sum((f[b][:, torch.from_numpy(idx[b]).long()] * torch.from_numpy(gy[b]).double()).sum()
              for b in range(B))
    tot.backward()
    np.testing.assert_allclose(gx, f.grad.numpy(), rtol=1e-5, atol=1e-5)
    i1 = idx[:, :, 0].copy()
    assert np.array_equal(orc.gather_features_forward(feat, i1),
                          np.stack([feat[b][:, i1[b]] for b in range(B)]))
    g1 = rng.standard_normal((B, C, M)).astype(np.float32)
    ref = np.zeros((B, C, N), np.float64)
    for b in range(B):
        np.add.at(ref[b], (slice(None), i1[b]), g1[b])
    np.testing.assert_allclose(orc.gather_features_backward(g1, i1, N), ref, rtol=1e-5, atol=1e-6)


def test_fps_vs_python(orc):
    rng = np.random.default_rng(4)
    B, N, M = 2, 400, 60
    co = gaussian_cloud(rng, B, N)
    got = orc.furthest_point_sampling(co, M)
    for b in range(B):
        p = co[b].T.astype(np.float32)
        dist = np.full(N, 1e38, np.float32)
        sel = [0]
        for _ in range(1, M):
            d = ((p - p[sel[-1]]) ** 2)
            d = (d[:, 0] + d[:, 1]) + d[:, 2]
            dist = np.minimum(dist, d)
            sel.append(int(np.argmax(dist)))  # random data: no exact ties; N<512 -> lowest index
        assert np.array_equal(got[b], np.array(sel, np.int32))
    assert (got[:, 0] == 0).all()


def test_fps_tie_break_matches_512_thread_tree(orc):
    """N > 512 with exact duplicates: the winner is the lowest (k mod 512, k)."""
    N = 1100
    co = np.zeros((1, 3, N), np.float32)
    co[0, 0, 600] = 5.0   # k mod 512 = 88
    co[0, 0, 40] = 5.0    # k mod 512 = 40  -> wins over 600 and 1000
    co[0, 0, 1000] = 5.0  # k mod 512 = 488
    got = orc.furthest_point_sampling(co, 2)
    assert got[0, 1] == 40
    co2 = np.zeros((1, 3, N), np.float32)
    co2[0, 0, 600] = 5.0  # mod 88
    co2[0, 0, 100] = 5.0  # mod 100 -> 600 wins although 100 < 600
    assert orc.furthest_point_sampling(co2, 2)[0, 1] == 600


def test_three_nn_vs_topk(orc):
    rng = np.random.default_rng(5)
    B, C, N, M = 2, 7, 200, 33
    pts = gaussian_cloud(rng, B, N)
    ctr = gaussian_cloud(rng, B, M)
    cf = rng.standard_normal((B, C, M)).astype(np.float32)
    out, idx, w = orc.three_nn_interpolate_forward(pts, ctr, cf)
    d = torch.cdist(torch.from_numpy(pts).double().transpose(1, 2),
                    torch.from_numpy(ctr).double().transpose(1, 2)) ** 2
    dk, ik = torch.topk(d, 3, dim=2, largest=False)
    assert np.array_equal(idx, ik.permute(0, 2, 1).int().numpy())
    dk = dk.clamp(1e-10, 1e10)
    inv = 1.0 / dk
    wref = (inv / inv.sum(2, keepdim=True)).permute(0, 2, 1)
    np.testing.assert_allclose(w, wref.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(w.sum(1), 1.0, atol=1e-6)
    ref = sum(np.stack([cf[b][:, idx[b, q]] for b in range(B)]) * w[:, q:q + 1] for q in range(3))
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-6)
    gy = rng.standard_normal((B, C, N)).astype(np.float32)
    gx = orc.three_nn_interpolate_backward(gy, idx, w, M)
    refg = np.zeros((B, C, M), np.float64)
    for b in range(B):
        for q in range(3):
            np.add.at(refg[b], (slice(None), idx[b, q]), gy[b] * w[b, q])
    np.testing.assert_allclose(gx, refg, rtol=1e-4, atol=1e-5)
    # fewer than 3 centres: the missing slots keep index 0 and the clamped 1e10 distance
    out1, idx1, w1 = orc.three_nn_interpolate_forward(pts, ctr[:, :, :1].copy(), cf[:, :, :1].copy())
    assert (idx1 == 0).all() and np.isfinite(w1).all()


def test_chamfer_vs_distance_matrix(orc):
    """Same acceptance shape as the reference's unit_test.py:14-35 ([4,100,3] vs [4,200,3])."""
    rng = np.random.default_rng(6)
    x1 = rng.random((4, 100, 3)).astype(np.float32)
    x2 = rng.random((4, 200, 3)).astype(np.float32)
    d1, d2, i1, i2 = orc.chamfer_forward(x1, x2)
    P = ((torch.from_numpy(x1).double()[:, :, None] - torch.from_numpy(x2).double()[:, None]) ** 2).sum(-1)
    assert np.array_equal(i1, P.argmin(2).int().numpy())
    assert np.array_equal(i2, P.argmin(1).int().numpy())
    assert ((d1 - P.min(2)[0].numpy()) ** 2).mean() < 1e-8
    assert ((d2 - P.min(1)[0].numpy()) ** 2).mean() < 1e-8
    # gradient == autograd of min-distance sums
    g1 = rng.standard_normal((4, 100)).astype(np.float32)
    g2 = rng.standard_normal((4, 200)).astype(np.float32)
    gx1, gx2 = orc.chamfer_backward(x1, x2, g1, g2, i1, i2)
    a = torch.from_numpy(x1).double().requires_grad_(True)
    b = torch.from_numpy(x2).double().requires_grad_(True)
    P = ((a[:, :, None] - b[:, None]) ** 2).sum(-1)
    ((P.min(2)[0] * torch.from_numpy(g1).double()).sum() + (P.min(1)[0] * torch.from_numpy(g2).double()).sum()).backward()
    np.testing.assert_allclose(gx1, a.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gx2, b.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_chamfer_tie_lowest_index(orc):
    x1 = np.zeros((1, 1, 3), np.float32)
    x2 = np.ones((1, 700, 3), np.float32)  # all targets equidistant, spans two 512-tiles
    _, _, i1, _ = orc.chamfer_forward(x1, x2)
    assert i1[0, 0] == 0


def test_emd_two_point_known_answer(orc):
    """third_party/PyTorchEMD/test_emd_loss.py:7-19 -- the crossed matching is optimal."""
    p1 = np.array([[[1.7, -0.1, 0.1], [0.1, 1.2, 0.3]]], np.float32).repeat(3, 0)
    p2 = np.array([[[0.3, 1.8, 0.2], [1.2, -0.2, 0.3]]], np.float32).repeat(3, 0)
    match = orc.approxmatch(p1, p2)
    cost = orc.matchcost(p1, p2, match) / 2.0
    gt = (((p1[0, 0] - p2[0, 1]) ** 2).sum() + ((p1[0, 1] - p2[0, 0]) ** 2).sum()) / 2
    np.testing.assert_allclose(cost, gt, rtol=1e-4)
    np.testing.assert_allclose(match[0], np.array([[0, 1], [1, 0]], np.float32), atol=1e-4)
    g1, g2 = orc.matchcost_backward(np.ones(3, np.float32), p1, p2, match)
    np.testing.assert_allclose(g1[0, 0], 2 * (p1[0, 0] - p2[0, 1]), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(g2[0, 0], 2 * (p2[0, 0] - p1[0, 1]), rtol=1e-3, atol=1e-4)


def test_emd_is_a_transport_plan(orc):
    rng = np.random.default_rng(8)
    x1 = rng.random((2, 64, 3)).astype(np.float32)
    x2 = rng.random((2, 64, 3)).astype(np.float32)
    match = orc.approxmatch(x1, x2)  # [b, m, n]
    assert (match >= 0).all()
    np.testing.assert_allclose(match.sum(1), 1.0, atol=2e-3)  # every xyz1 point fully shipped
    np.testing.assert_allclose(match.sum(2), 1.0, atol=2e-2)
    same = orc.matchcost(x1, x1, orc.approxmatch(x1, x1))
    assert (same < 1e-3).all()


def test_p1_matches_torch_voxelization_formula(orc):
    """Oracle P1 vs the formula of models/pvcnn2_ada.py:173-188 evaluated with torch ops.  torch's
    mean() uses a different (unspecified) summation order, so a coordinate that lands within an
    ulp of k+0.5 may round differently: allow <= 1e-4 of the voxel ids to differ and require each
    differing coordinate to sit on a rounding boundary."""
    rng = np.random.default_rng(9)
    for N, r in [(2048, 32), (1024, 16), (256, 8)]:
        co = surface_cloud(rng, 8, N) * 0.5 + 0.2
        nc, vox = orc.voxelize_coords(co, r, True, 0.0)
        t = torch.from_numpy(co)
        n = t - t.mean(2, keepdim=True)
        n = n / (n.norm(dim=1, keepdim=True).max(dim=2, keepdim=True).values * 2.0 + 0.0) + 0.5
        n = torch.clamp(n * r, 0, r - 1)
        v = torch.round(n).to(torch.int32).numpy()
        np.testing.assert_allclose(nc, n.numpy(), rtol=0, atol=1e-4)
        diff = vox != v
        assert diff.mean() <= 1e-4
        frac = np.abs(nc[diff] - np.floor(nc[diff]) - 0.5)
        assert (frac < 1e-3).all()
        assert vox.min() >= 0 and vox.max() <= r - 1


def test_ddim_ddpm_update_vs_torch(orc):
    rng = np.random.default_rng(10)
    x, e, z = (rng.standard_normal(5000).astype(np.float32) for _ in range(3))
    s, c, sg = np.float32(0.99991), np.float32(0.0123), np.float32(0.02)
    tx, te, tz = map(torch.from_numpy, (x, e, z))
    ref = tx * torch.tensor(s)
    ref = ref + (torch.tensor(c) * te + torch.tensor(sg) * tz)   # diffusion_pvd.py:451,465
    assert np.array_equal(orc.ddim_update(x, e, z, s, c, sg), ref.numpy())
    ko, ka, kb, sc = np.float32(1.00005), np.float32(1e-4), np.float32(0.83), np.float32(0.01)
    mean = torch.tensor(ko) * (tx - torch.tensor(ka) * te / torch.tensor(kb))
    ref = mean + torch.tensor(sc) * tz * 1.0
    assert np.array_equal(orc.ddpm_update(x, e, z, False, ko, ka, kb, sc, 1.0), ref.numpy())
    ref0 = torch.tensor(ko) * (tx - torch.tensor(ka) * te)
    assert np.array_equal(orc.ddpm_update(x, e, z, True, ko, ka, kb, sc, 1.0), ref0.numpy())
