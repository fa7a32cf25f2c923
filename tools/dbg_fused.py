import sys, torch
sys.path.insert(0, "/root/repo")
from lion_amd import fused_ops
from lion_amd.functional.backend import _backend as bk
torch.manual_seed(0)
B, cin, cout, r, n = 3, 16, 32, 8, 256
conv1 = torch.nn.Conv3d(cin, cout, 3, padding=1).cuda(); conv2 = torch.nn.Conv3d(cout, cout, 3, padding=1).cuda()
gn = torch.nn.GroupNorm(8, cout).cuda(); gn.weight.data.normal_(); gn.bias.data.normal_()
x = torch.randn(B, cin, r, r, r, device="cuda")
fac, gb = torch.randn(B, cout, device="cuda"), torch.randn(B, cout, device="cuda")
with torch.no_grad():
    y1, st = fused_ops.conv3d_fused(x, conv1, None, True)
    ref1 = conv1(x)
    print("conv err", (y1 - ref1).abs().max().item())
    s1 = st[..., 0].sum(-1); s2 = st[..., 1].sum(-1)
    print("sum err", (s1 - ref1.flatten(2).sum(-1)).abs().max().item(), "sq err", (s2 - ref1.flatten(2).square().sum(-1)).abs().max().item())
    A, Bs, cm = fused_ops.groupnorm_fold(st, gn, fac, gb, r ** 3)
    ref_ada = gn(ref1) * fac.view(B, -1, 1, 1, 1) + gb.view(B, -1, 1, 1, 1)
    got_ada = y1 * A.view(B, -1, 1, 1, 1) + Bs.view(B, -1, 1, 1, 1)
    print("adagn err", (got_ada - ref_ada).abs().max().item(), "chmean err", (cm - ref1.flatten(2).mean(-1)).abs().max().item())
    y2, _ = fused_ops.conv3d_fused(y1, conv2, (A, Bs), False)
    act = ref_ada * torch.sigmoid(ref_ada)
    ref2 = conv2(act)
    print("conv2 prologue err", (y2 - ref2).abs().max().item(), ref2.abs().max().item())
    co = torch.rand(B, 3, n, device="cuda") * (r - 1)
    sc, sh = torch.randn(B, cout, device="cuda"), torch.randn(B, cout, device="cuda")
    d1 = fused_ops.devoxelize_affine(ref2, co, r, sc, sh)
    d2, _, _ = bk.trilinear_devoxelize_forward(r, False, co, (ref2 * sc.view(B, -1, 1, 1, 1) + sh.view(B, -1, 1, 1, 1)).flatten(2).contiguous())
    print("devox affine err", (d1 - d2).abs().max().item())
