"""10 forwards of the global prior (PriorSEDrop, 309 MB of fp32 weights streamed per forward) at B=32."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd.config import released_prior_cfg
from lion_amd.models import import_model
cfg = released_prior_cfg()
torch.manual_seed(0)
m = import_model(cfg.latent_pts.style_prior)(cfg.sde, cfg.latent_pts.style_dim, cfg).cuda().eval()
x = torch.randn(32, 128, 1, 1, device="cuda"); t = torch.full((32,), 500.0, device="cuda")
with torch.no_grad():
    for _ in range(10): m(x=x, t=t, condition_input=None, clip_feat=None)
torch.cuda.synchronize()
