"""Runs W warm-up + K timed DDIM steps of both priors at B=32 for rocprofv3; a chamfer launch (never
used by sampling) brackets the timed region so tools/kstats.py --between-markers can drop MIOpen's
find/tuning kernels and the warm-up."""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd.config import released_prior_cfg
from lion_amd.models.lion import LION
from lion_amd import diffusion_ops
from lion_amd.chamfer3d import chamfer_3DDist_nograd

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--only", default="both")
ap.add_argument("--evolve", type=int, default=0, help="run this many chain steps first (the latents drift along the chain)")
a = ap.parse_args()
torch.manual_seed(0)
cfg = released_prior_cfg()
lion = LION(cfg); lion.priors.eval(); lion.vae.eval()
d = lion.diffusion; B = a.batch; dev = torch.device("cuda")
sh = lion.vae.latent_shape(); steps = d.ddim_schedule(1000, 1000)
mk = chamfer_3DDist_nograd(); mx = torch.rand(1, 64, 3, device=dev)

def run(model, x, cond, n, first=0):
    for i in range(first, first + n):
        t = steps[i]; s, c, sg = d.ddim_coefficients(t, steps[i + 1], 1.0)
        eps = model(x=x, t=torch.full((B,), float(t + 1), device=dev), condition_input=cond, clip_feat=None).float().contiguous()
        x = diffusion_ops.ddim_update(x, eps, torch.randn_like(x), s, c, sg)
    return x

with torch.no_grad():
    xg = torch.randn([B] + sh[0], device=dev); xl = torch.randn([B] + sh[1], device=dev)
    style = lion.vae.global2style(torch.randn([B] + sh[0], device=dev))
    run(lion.priors[0], xg, None, a.warmup); run(lion.priors[1], xl, style, a.warmup)
    if a.evolve:
        xg = run(lion.priors[0], xg, None, a.evolve); xl = run(lion.priors[1], xl, style, a.evolve)
    torch.cuda.synchronize(); mk(mx, mx); torch.cuda.synchronize()
    if a.only in ("both", "global"): run(lion.priors[0], xg, None, a.steps, a.evolve)
    if a.only in ("both", "local"): run(lion.priors[1], xl, style, a.steps, a.evolve)
    torch.cuda.synchronize(); mk(mx, mx); torch.cuda.synchronize()
print("done")
