"""-m gpu: lion_groupnorm_fold_se (GroupNorm fold + SE3d gate in one launch) against the two-launch path it replaces
(lion_groupnorm_fold, lion_se_gate) and against the module arithmetic in float64 (models/adagn.py:45-65,
models/pvcnn2_ada.py:27-41)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,C,T", [(32, 64, 128), (4, 128, 16), (3, 256, 2), (2, 32, 77), (5, 8, 1)])
def test_fold_se_equals_fold_then_gate(B, C, T):
    from lion_amd import fused_ops
    from lion_amd.models.pvcnn2_ada import SE3d
    torch.manual_seed(B * 1000 + C + T)
    voxels = 4096
    gn = torch.nn.GroupNorm(8, C).cuda()
    se = SE3d(C).cuda()
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.uniform_(-0.5, 0.5)
    # per-tile sums of a plausible conv output: sum and sum of squares over voxels / T elements each
    n_t = voxels // T if voxels % T == 0 else voxels / T
    x = torch.randn(B, C, T, 64, device="cuda") * 0.7 + 0.2
    s1 = x.sum(-1) * (n_t / 64.0)
    s2 = (x * x).sum(-1) * (n_t / 64.0)
    stats = torch.stack([s1, s2], -1).contiguous()
    proj = torch.randn(B, 2 * C, device="cuda") * 0.3 + torch.cat([torch.ones(C), torch.zeros(C)]).cuda()
    fac, gb = proj.chunk(2, 1)
    merged = fused_ops.groupnorm_fold_se(stats, gn, fac, gb, voxels, se)
    assert merged is not None
    A, Bs, cm = fused_ops.groupnorm_fold(stats, gn, fac, gb, voxels)
    A2, B2 = fused_ops.se_gate_(A.clone(), Bs.clone(), cm, se)
    for got, want in ((merged[0], A2), (merged[1], B2)):
        assert (got - want).abs().max().item() <= 2e-6 * max(want.abs().max().item(), 1e-3)
    # float64 module arithmetic
    d = stats.double()
    mean_c = d[..., 0].sum(-1) / voxels
    g = C // 8
    gm = d[..., 0].sum(-1).view(B, 8, g).sum(-1) / (voxels * g)
    gv = d[..., 1].sum(-1).view(B, 8, g).sum(-1) / (voxels * g) - gm * gm
    rstd = (gv.clamp_min(0) + gn.eps).rsqrt().repeat_interleave(g, 1)
    gmc = gm.repeat_interleave(g, 1)
    a = rstd * gn.weight.double() * fac.double()
    b = (gn.bias.double() - gmc * rstd * gn.weight.double()) * fac.double() + gb.double()
    m = a * mean_c + b
    gate = torch.sigmoid(torch.relu(m @ se.fc[0].weight.double().t()) @ se.fc[2].weight.double().t())
    for got, want in ((merged[0].double(), a * gate), (merged[1].double(), b * gate)):
        assert (got - want).abs().max().item() <= 1e-5 * max(want.abs().max().item(), 1e-3)


@pytest.mark.parametrize("cin,mid,cout,M", [(35, 32, 64, 1024), (19, 48, 96, 512)])
def test_sa_mlp_last_layer_max_without_its_output(cin, mid, cout, M):
    """fused_ops.shared_mlp(reduce_max=True) on a grouped activation [B, Cin, M, 32]: the last layer evaluated twice
    (GroupNorm sums, then AdaGN + Swish + max over the 32 neighbours in the GEMM's epilogue: lion_pwconv_forward_max) ==
    the same layer stored and reduced by lion_affine_swish_max, bit for bit.  pvcnn2_ada.py:120-164, :375-377."""
    from types import SimpleNamespace
    from lion_amd import fused_ops
    from lion_amd.models.adagn import AdaGN
    torch.manual_seed(cin + M)
    B = 32
    cfg = SimpleNamespace(latent_pts=SimpleNamespace(style_dim=128, ada_mlp_init_scale=1.0))
    convs = [torch.nn.Conv2d(cin, mid, 1).cuda(), torch.nn.Conv2d(mid, cout, 1).cuda()]
    gns = [AdaGN(2, cfg, mid).cuda(), AdaGN(2, cfg, cout).cuda()]
    x = torch.randn(B, cin, M, 32, device="cuda")
    style = torch.randn(B, 128, device="cuda")
    saved = fused_ops.MAX_RECOMPUTE
    try:
        with torch.no_grad():
            fused_ops.MAX_RECOMPUTE = False
            ref = fused_ops.shared_mlp(x, convs, gns, style, reduce_max=True)
            fused_ops.MAX_RECOMPUTE = True
            got = fused_ops.pwconv_max_recompute  # the path must actually be taken for this shape
            calls = []
            fused_ops.pwconv_max_recompute = lambda *a, **k: calls.append(1) or got(*a, **k)
            try:
                out = fused_ops.shared_mlp(x, convs, gns, style, reduce_max=True)
            finally:
                fused_ops.pwconv_max_recompute = got
    finally:
        fused_ops.MAX_RECOMPUTE = saved
    assert calls and tuple(out.shape) == (B, cout, M)
    assert torch.equal(out, ref)


def _fold_case(B, cin, cout, r, with_se, sparse):
    from lion_amd import fused_ops
    from lion_amd.functional.backend import _backend as bk
    from lion_amd.models.pvcnn2_ada import SE3d
    torch.manual_seed(B + cin + cout + r)
    conv1 = torch.nn.Conv3d(cin, cout, 3, padding=1).cuda()
    conv2 = torch.nn.Conv3d(cout, cout, 3, padding=1).cuda()
    gn1, gn2 = torch.nn.GroupNorm(8, cout).cuda(), torch.nn.GroupNorm(8, cout).cuda()
    se = SE3d(cout).cuda() if with_se else None
    with torch.no_grad():
        for g in (gn1, gn2):
            g.weight.uniform_(0.5, 1.5)
            g.bias.uniform_(-0.5, 0.5)
    proj = torch.randn(B, 4 * cout, device="cuda") * 0.3 + 1.0
    f1, g1, f2, g2 = proj.chunk(4, 1)
    feats = torch.randn(B, cin, 2048 if r == 32 else 512, device="cuda")
    coords = torch.randn(B, 3, feats.shape[2], device="cuda")
    grid, _, _, cnt = bk.voxelize_points_forward(feats, coords, r, True, 0.0)
    grid = grid.view(B, cin, r, r, r)
    return fused_ops, conv1, conv2, gn1, gn2, se, (f1, g1, f2, g2), grid, cnt


@pytest.mark.parametrize("B,cin,cout,r,with_se,sparse", [(6, 128, 128, 8, True, 0), (32, 128, 128, 8, False, 0), (1, 192, 128, 8, True, 0),
                                                         (3, 128, 256, 8, True, 0), (4, 64, 64, 32, True, 2), (5, 64, 128, 16, True, 2)])
def test_fold_in_the_producers_tail_equals_the_separate_fold_launch(B, cin, cout, r, with_se, sparse, monkeypatch):
    """csrc/fold.h: the workgroup that finishes a sample's last tile folds that sample's GroupNorm sums (+ SE gate) itself.
    (A, Bs) must equal the separate lion_groupnorm_fold_se launch on the same tile sums BIT FOR BIT (same arithmetic, same
    order, whichever workgroup arrives last), for dense / sparse / consumer-aware launches, on every repetition (the arrival
    counters re-arm themselves), with the sums buffer poisoned beforehand (a stale or early read would show)."""
    fo, conv1, conv2, gn1, gn2, se, (f1, g1, f2, g2), grid, cnt = _fold_case(B, cin, cout, r, with_se, sparse)
    n = r ** 3

    monkeypatch.setattr(fo, "FOLD_MAX_TILES", 1 << 20)      # every resolution through the tail (the product: few-tile layers)

    def run(in_tail):
        monkeypatch.setattr(fo, "FOLD_IN_PRODUCER", in_tail)
        occ1 = occ2 = None
        if sparse and r >= 16:
            occ1, occ2 = fo.conv3d_occupancy(cnt, r, cout, B, consumer_aware=sparse)
        y1, ab1 = fo.conv3d_fused(grid, conv1, None, True, occ1, fold=fo.FoldSpec(gn1, f1, g1, n))
        y2, ab2 = fo.conv3d_fused(y1, conv2, ab1, True, occ2, prev_conv=conv1, fold=fo.FoldSpec(gn2, f2, g2, n, se))
        return ab1, ab2

    with torch.no_grad():
        ref1, ref2 = run(False)
        for rep in range(6):
            # poison what the caching allocator will hand out next as the sums buffers
            junk = [torch.full((B, cout, 128, 2), float("nan"), device="cuda") for _ in range(4)]
            del junk
            got1, got2 = run(True)
            for got, ref in ((got1, ref1), (got2, ref2)):
                assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), rep
    # the counters of both layers are back at zero (r = 8: the layers that fold in their tail)
    for owner in (conv1, conv2):
        hit = fo._FOLD_COUNTERS.get((id(owner), B, str(grid.device)))
        assert (hit is not None) == (r == 8)
        if hit is not None:
            assert int(hit[1].abs().sum()) == 0


def test_fold_in_tail_inside_a_replayed_graph():
    """the same through hipGraph replay (how the sampling step runs it): 20 replays, identical results"""
    fo, conv1, conv2, gn1, gn2, se, (f1, g1, f2, g2), grid, cnt = _fold_case(8, 128, 128, 8, True, 0)
    n = 8 ** 3
    saved = fo.FOLD_MAX_TILES

    def run():
        occ1 = occ2 = None
        y1, ab1 = fo.conv3d_fused(grid, conv1, None, True, occ1, fold=fo.FoldSpec(gn1, f1, g1, n))
        y2, ab2 = fo.conv3d_fused(y1, conv2, ab1, True, occ2, prev_conv=conv1, fold=fo.FoldSpec(gn2, f2, g2, n, se))
        return ab2
    with torch.no_grad():
        ref = [t.clone() for t in run()]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            run()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = run()
        for _ in range(20):
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
    fo.FOLD_MAX_TILES = saved
