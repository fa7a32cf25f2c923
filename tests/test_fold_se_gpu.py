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

