import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from conftest import fill_
from lion_amd.config import released_prior_cfg
from lion_amd.models.shapelatent_modules import PointNetPlusEncoder
z = np.load(os.path.join(R, "tests/golden/style_encoder.npz"))
m = PointNetPlusEncoder(zdim=128, input_dim=3, args=released_prior_cfg()); fill_(m); m.cuda().train()
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
x = torch.from_numpy(z["x"]).cuda()
o = m(x); w = torch.from_numpy(z["w"]).cuda()
((o['mu_1d'] * w).sum() + (o['sigma_1d'] * w).sum()).backward()
print("mu err", np.abs(o['mu_1d'].detach().cpu().numpy() - z["mu"]).max(), np.abs(z["mu"]).max())
P = dict(m.named_parameters())
for k in z.files:
    if k.startswith("g_"):
        ref = z[k]; got = P[k[2:]].grad.cpu().numpy(); print(k, np.abs(got - ref).max() / np.abs(ref).max())
