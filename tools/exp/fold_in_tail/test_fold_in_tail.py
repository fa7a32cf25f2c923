"""the GPU tests of the tail fold (ran green while it was in the product; not collected from here)"""
import pytest
import torch


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
