"""torch.profiler table of one eager local-prior + global-prior step (op names, shapes, counts)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from lion_amd.config import released_prior_cfg
from lion_amd.models.lion import LION
torch.manual_seed(0)
cfg = released_prior_cfg(); lion = LION(cfg); lion.priors.eval(); lion.vae.eval()
B = 32; dev = torch.device("cuda"); sh = lion.vae.latent_shape()
sh = lion.vae.latent_shape()
xg = torch.randn([B] + sh[0], device=dev); xl = torch.randn([B] + sh[1], device=dev)
t = torch.full((B,), 500.0, device=dev)
with torch.no_grad():
    for _ in range(2):
        eg = lion.priors[0](x=xg, t=t, condition_input=None, clip_feat=None)
        cond = lion.vae.global2style(xg)
        el = lion.priors[1](x=xl, t=t, condition_input=cond, clip_feat=None)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        eg = lion.priors[0](x=xg, t=t, condition_input=None, clip_feat=None)
        el = lion.priors[1](x=xl, t=t, condition_input=cond, clip_feat=None)
        torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=70, max_name_column_width=40, max_shapes_column_width=70))
