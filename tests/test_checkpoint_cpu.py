"""Checkpoint files: the reference's two formats (trainers/base_trainer.py:90-146, trainers/train_prior.py:294-350,
models/lion.py:30-35) written and read by lion_amd.checkpoint; released-weights layout from tests/golden."""
import json
import os

import pytest
import torch
import torch.nn as nn

from lion_amd import checkpoint as ck

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _net():
    return nn.Sequential(nn.Linear(4, 8), nn.GroupNorm(2, 8), nn.Linear(8, 3))


def _trained(net, steps=2):
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    for _ in range(steps):
        opt.zero_grad()
        net(torch.randn(5, 4)).square().sum().backward()
        opt.step()
    return opt


def _same(a, b):
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_vae_format_round_trip(tmp_path):
    torch.manual_seed(0)
    net = _net()
    opt = _trained(net)
    path = ck.save_vae_checkpoint(str(tmp_path), net, opt, epoch=7, step=1234, appendix={"note": "x"})
    assert path == os.path.join(str(tmp_path), "checkpoints", "epoch_7_iters_1234.pt")
    raw = torch.load(path)
    assert set(raw) == {"opt", "model", "epoch", "step", "note"}  # base_trainer.py:91-98; no scaler -> no key
    net2 = _net()
    opt2 = torch.optim.Adam(net2.parameters(), lr=1e-2)
    assert ck.load_vae_checkpoint(path, net2, opt2) == (7, 1234)
    _same(net.state_dict(), net2.state_dict())
    s1, s2 = opt.state_dict()["state"], opt2.state_dict()["state"]
    assert all(torch.equal(s1[i]["exp_avg"], s2[i]["exp_avg"]) for i in s1)
    with pytest.raises(KeyError):
        ck.load_vae_checkpoint(path, _net(), grad_scalar=torch.amp.GradScaler("cpu", enabled=True))
    assert not os.path.exists(path + ".tmp")


def test_vae_format_accepts_wrapped_and_legacy_files(tmp_path):
    torch.manual_seed(1)
    net = _net()
    sd = net.state_dict()
    for n, (key, prefix) in enumerate((("model", "module."), ("model_state", "model.module."), ("model_state", ""))):
        p = str(tmp_path / f"old{n}.pt")
        torch.save({key: {prefix + k: v for k, v in sd.items()}, "epoch": 3}, p)
        net2 = _net()
        assert ck.load_vae_checkpoint(p, net2) == (3, 0)  # no 'step' in old files -> 0 (base_trainer.py:137)
        _same(sd, net2.state_dict())
    assert ck.strip_wrapper_prefixes({"module.model.module.w": 1}) == {"model.module.w": 1}  # one prefix per key


def test_prior_format_round_trip(tmp_path):
    torch.manual_seed(2)
    dae, vae = nn.ModuleList([_net(), _net()]), _net()
    dopt, vopt = _trained(dae[0]), _trained(vae)
    dsch = torch.optim.lr_scheduler.CosineAnnealingLR(dopt, 10)
    vsch = torch.optim.lr_scheduler.CosineAnnealingLR(vopt, 10)
    dsch.step()
    scaler = torch.amp.GradScaler("cpu", init_scale=2.0 ** 10, enabled=True)
    path = ck.save_prior_checkpoint(str(tmp_path), dae, vae, epoch=4, step=99, dae_optimizer=dopt,
                                    vae_optimizer=vopt, dae_scheduler=dsch, vae_scheduler=vsch, grad_scalar=scaler)
    raw = torch.load(path)
    assert set(raw) == {"epoch", "global_step", "grad_scalar", "dae_state_dict", "dae_optimizer", "dae_scheduler",
                        "vae_state_dict", "vae_optimizer", "vae_scheduler"}  # train_prior.py:333-339
    assert raw["epoch"] == 5 and os.path.basename(path) == "epoch_4_iters_99.pt"  # stored epoch = finished + 1
    dae2, vae2 = nn.ModuleList([_net(), _net()]), _net()
    dopt2 = torch.optim.Adam(dae2[0].parameters(), lr=1e-2)
    dsch2 = torch.optim.lr_scheduler.CosineAnnealingLR(dopt2, 10)
    assert ck.load_prior_checkpoint(path, dae2, vae2, dae_optimizer=dopt2, dae_scheduler=dsch2) == (5, 99)
    _same(dae.state_dict(), dae2.state_dict())
    _same(vae.state_dict(), vae2.state_dict())
    assert dsch2.last_epoch == dsch.last_epoch and dopt2.param_groups[0]["lr"] == dopt.param_groups[0]["lr"]
    # a sampling-only file (weights, no optimizers) loads for sampling and refuses to resume training
    p2 = ck.save_prior_checkpoint(str(tmp_path), dae, vae, epoch=0, step=0, save_name="weights.pt")
    assert ck.load_prior_checkpoint(p2, dae2, vae2) == (1, 0)
    with pytest.raises(KeyError):
        ck.load_prior_checkpoint(p2, dae2, vae2, dae_optimizer=dopt2)
    _same(ck.load_pretrained_vae(ck.save_vae_checkpoint(str(tmp_path), vae, vopt, 0, 0), _net()).state_dict(),
          vae.state_dict())


def test_released_checkpoint_layout_loads_unchanged(tmp_path):
    """A file with the reference's exact parameter names / shapes (ModuleList([global, local]) under
    'dae_state_dict', the VAE under 'vae_state_dict') loads strictly into LION's modules (models/lion.py:30-35)."""
    from lion_amd.config import released_prior_cfg
    from lion_amd.models.latent_points_ada_localprior import PVCNN2Prior
    from lion_amd.models.score_sde.resnet import PriorSEDrop
    from lion_amd.models.vae_adain import Model
    lay = json.load(open(os.path.join(GOLDEN, "state_dict_layouts.json")))
    gen = torch.Generator().manual_seed(3)
    rnd = lambda shape: torch.randn(shape, generator=gen) if shape else torch.zeros((), dtype=torch.long)
    dae_sd = {f"{i}.{k}": rnd(s) for i, name in enumerate(("PriorSEDrop", "PVCNN2Prior")) for k, s in lay[name].items()}
    vae_sd = {k: rnd(s) for k, s in lay["vae_adain.Model"].items()}
    path = str(tmp_path / "released.pt")
    torch.save({"dae_state_dict": dae_sd, "vae_state_dict": vae_sd, "epoch": 8000, "global_step": 1}, path)
    cfg = released_prior_cfg()
    priors = nn.ModuleList([PriorSEDrop(cfg.sde, cfg.latent_pts.style_dim, cfg),
                            PVCNN2Prior(cfg.sde, cfg.shapelatent.latent_dim, cfg)])
    vae = Model(cfg)
    assert ck.load_prior_checkpoint(path, priors, vae) == (8000, 1)
    got = priors.state_dict()
    for k, v in dae_sd.items():
        assert torch.equal(got[k], v.to(got[k].dtype)), k
    got = vae.state_dict()
    for k, v in vae_sd.items():
        assert torch.equal(got[k], v.to(got[k].dtype)), k
