"""CPU tier: models/adagn.py StylePlan under autograd (round 6) -- the AdaGN style projections of a network batched into one
Linear per forward in TRAINING too: same values and gradients as one Linear per layer, and a module the forward never calls keeps
grad = None (it is left out of the batch: the first training forward records who asks)."""
import copy

import torch

from lion_amd.config import released_prior_cfg
from lion_amd.models import adagn as A


class Net(torch.nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.a = A.AdaGN(1, cfg, 16)
        self.b = A.AdaGN(1, cfg, 32)
        self.unused = A.AdaGN(1, cfg, 8)
        self.lin = torch.nn.Conv1d(16, 32, 1)
        self.plan = None

    def forward(self, x, style):
        if self.plan is None:
            self.plan = A.StylePlan(self)
        with self.plan.projected(style):
            h = self.a(x, style)
            h = self.b(self.lin(h), style)
            return self.a(h[:, :16] + x, style)      # a module used twice in one forward


def _run(net, x, style, n_forwards):
    for _ in range(n_forwards):
        net.zero_grad(set_to_none=True)
        xs, ss = x.clone().requires_grad_(True), style.clone().requires_grad_(True)
        y = net(xs, ss)
        y.square().mean().backward()
    return y.detach(), xs.grad, ss.grad, {k: (None if p.grad is None else p.grad.clone()) for k, p in net.named_parameters()}


def test_batched_style_projection_under_autograd_equals_per_layer(monkeypatch):
    cfg = released_prior_cfg()
    torch.manual_seed(0)
    net = Net(cfg).double()
    for p in net.parameters():
        torch.nn.init.normal_(p, std=0.3)
    ref_net = copy.deepcopy(net)
    x, style = torch.randn(3, 16, 10, dtype=torch.float64), torch.randn(3, cfg.latent_pts.style_dim, dtype=torch.float64)
    monkeypatch.setattr(A, "TRAIN_BATCHED", False)
    y0, gx0, gs0, gp0 = _run(ref_net, x, style, 1)
    monkeypatch.setattr(A, "TRAIN_BATCHED", True)
    y1, gx1, gs1, gp1 = _run(net, x, style, 1)              # the recording forward: still per layer
    assert net.plan._used is not None and len(net.plan._used) == 2
    y2, gx2, gs2, gp2 = _run(net, x, style, 1)              # batched
    for got in ((y1, gx1, gs1, gp1), (y2, gx2, gs2, gp2)):
        assert torch.allclose(got[0], y0, rtol=1e-12, atol=1e-14)
        assert torch.allclose(got[1], gx0, rtol=1e-10, atol=1e-14) and torch.allclose(got[2], gs0, rtol=1e-10, atol=1e-14)
        for k, g in gp0.items():
            if g is None:
                assert got[3][k] is None, k                 # the module nobody called: no gradient, as per layer
            else:
                assert torch.allclose(got[3][k], g, rtol=1e-10, atol=1e-14), k
    assert all(gp2[k] is None for k in gp2 if k.startswith("unused."))


def test_inference_style_plan_unchanged():
    cfg = released_prior_cfg()
    torch.manual_seed(1)
    net = Net(cfg)
    x, style = torch.randn(2, 16, 6), torch.randn(2, cfg.latent_pts.style_dim)
    with torch.no_grad():
        y_plan = net(x, style)
        net.plan = type("NoPlan", (), {"projected": lambda self, s: __import__("contextlib").nullcontext()})()
        y_plain = net(x, style)
    assert torch.allclose(y_plan, y_plain, rtol=1e-5, atol=1e-6)
