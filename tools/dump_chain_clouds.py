"""Dump the latent point clouds x_t of the local prior's real 1000-step DDIM chain at a few steps (graph replay with the
trajectory kept) to gpurun_out/chain_clouds.npz: tools/tile_shape_estimate.py --clouds evaluates the empty-tile /
empty-block / active-voxel fractions of the voxel convolutions on them (what the sparse plan sees IN the chain, as opposed
to the Gaussian / flat micro-benchmark clouds)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd.config import released_prior_cfg
from lion_amd.models.lion import LION

STEPS = [0, 1, 5, 20, 50, 100, 200, 400, 600, 800, 950, 999]
B = int(os.environ.get("B", "32"))
dev = torch.device("cuda", 0)
cfg = released_prior_cfg("airplane")
torch.manual_seed(0)
lion = LION(cfg, device=dev)
lion.priors.eval()
lion.vae.eval()
shapes = lion.vae.latent_shape()
d = lion.diffusion
with torch.no_grad():
    torch.manual_seed(1234)
    g, _ = d.run_ddim(lion.priors[0], B, shapes[0], 1.0, False, is_image=False, ddim_step=1000, condition_input=None,
                      keep_trajectory=False)
    style = lion.vae.global2style(g)
    x, traj = d.run_ddim(lion.priors[1], B, shapes[1], 1.0, False, is_image=False, ddim_step=1000, condition_input=style,
                         keep_trajectory=True)
torch.cuda.synchronize()
local = lion.priors[1]
out = {}
for s in STEPS:
    xt = traj[s]
    pts = xt.view(B, local.num_points, local.num_classes)[:, :, :3]   # [B, N, 3]
    out["step_%04d" % s] = pts.float().cpu().numpy()
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/chain_clouds.npz", **out)
print("saved", {k: v.shape for k, v in out.items()})
