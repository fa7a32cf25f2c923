import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd.config import released_prior_cfg
from lion_amd.models.vae_adain import Model as VAE
from lion_amd import training
from lion_amd.dist import BucketedGradAverager
use_avg = int(sys.argv[1]); B = int(sys.argv[2])
torch.manual_seed(0)
cfg = released_prior_cfg("chair")
vae = VAE(cfg).cuda().train()
params = list(vae.parameters())
opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.99))
avg = BucketedGradAverager(params) if use_avg else None
torch.manual_seed(1234)
x = torch.randn(B, 2048, 3, device="cuda")
for i in range(8):
    loss, out = training.vae_train_step(vae, opt, x, step=0, averager=avg)
    gn = sum((p.grad.float().norm() ** 2 for p in params if p.grad is not None)).sqrt().item()
    bad = [n for n, p in vae.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(i, float(loss), "rec", float(out['msg/rec']), "kl", float(out['msg/kl'].mean()), "gradnorm", gn, "nonfinite grads:", bad[:3], len(bad))
