import sys, os, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import fill_
import oracle
import lion_amd.functional.backend as bk
from lion_amd.config import released_prior_cfg
from lion_amd.models import distributions
from lion_amd.models.vae_adain import Model
distributions.Normal.sample = lambda self, t=1.0: (self.mu + 0.3 * self.sigma, None)
cfg = released_prior_cfg(); cfg.data.tr_max_sample_points = 1024; cfg.ddpm.dropout = 0.0; cfg.trainer.anneal_kl = 0
torch.manual_seed(0)
vae = Model(cfg); fill_(vae)
x = torch.randn(1, 1024, 3) * 0.5
g = copy.deepcopy(vae).cuda().train()
out = g.get_loss(x.cuda(), it=0); out['loss'].mean().backward()
hip = bk._backend; bk._backend = oracle.TorchBackend()
c = copy.deepcopy(vae).train(); outc = c.get_loss(x, it=0); outc['loss'].mean().backward()
bk._backend = hip
print("loss", out['loss'].mean().item(), outc['loss'].mean().item())
rows = []
for (n, pg), (_, pc) in zip(g.named_parameters(), c.named_parameters()):
    if pc.grad is None: continue
    s = pc.grad.abs().max().item()
    if s == 0: continue
    rows.append(((pg.grad.cpu() - pc.grad).abs().max().item() / s, n, s))
rows.sort(reverse=True)
for r in rows[:25]: print(f"{r[0]:.3e} {r[1]} scale {r[2]:.3e}")
import collections
by = collections.defaultdict(list)
for e, n, s in rows: by[n.split('.')[0]].append(e)
for k, v in by.items(): print(k, "max", max(v), "median", sorted(v)[len(v)//2], "n", len(v))
