"""Adds tests/golden/block_grads.npz: for the three block types every VAE / prior U-Net is built from (PVConv with
SE + LinearAttention, PointNetSAModule, PointNetFPModule) the gradient of EVERY parameter, produced by the
reference's own modules under PyTorch-CPU on the inputs already stored in blocks.npz (same name-derived weights).
Pins each backward kernel of the training path (Conv3d dgrad / wgrad, GroupNorm / AdaGN, SE3d, LinearAttention,
SharedMLP 1x1 convs, voxelize / devoxelize / grouping / interpolation backward) at block level.
Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_block_grads.py   (needs /root/reference)"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (registers the stand-ins, puts the reference on sys.path)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    cfg = mg.ref_cfg()
    from models import pvcnn2_ada as ref_ada
    z = np.load(os.path.join(HERE, "blocks.npz"))
    pv = ref_ada.PVConv(16, 32, 3, 8, with_se=True, attention=True, dropout=0.0, cfg=cfg)
    sa = ref_ada.PointNetSAModule(32, 0.5, 16, 16, [32, 48], cfg=cfg)
    fp = ref_ada.PointNetFPModule(48 + 16, [32, 24], cfg=cfg)
    for m in (pv, sa, fp):
        mg.fill_(m)
        m.train()
    feat = torch.from_numpy(z["pv_feat"]).requires_grad_(True)
    coords, sty = torch.from_numpy(z["pv_coords"]), torch.from_numpy(z["pv_style"])
    out = {}
    o = pv((feat, coords, None, sty))[0]
    o.square().sum().backward()
    assert np.array_equal(o.detach().numpy(), z["pv_out"])      # same run as blocks.npz
    for n, p in pv.named_parameters():
        out["pv/" + n] = p.grad.numpy().copy()
    feat.grad = None
    o_sa, c_sa, _, _ = sa((feat, coords, None, sty))
    o_sa.square().sum().backward()
    for n, p in sa.named_parameters():
        out["sa/" + n] = p.grad.numpy().copy()
    cfeat = torch.from_numpy(z["fp_cfeat"]).requires_grad_(True)
    o_fp = fp((coords, c_sa.detach(), cfeat, feat.detach(), None, sty))[0]
    o_fp.square().sum().backward()
    for n, p in fp.named_parameters():
        out["fp/" + n] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "block_grads.npz"), **out)
    print(len(out), "parameter gradients ->", os.path.join(HERE, "block_grads.npz"))


if __name__ == "__main__":
    main()
