"""Generates tests/golden/*.npz|json by IMPORTING THE REFERENCE (read-only, /root/reference) in the
build container.  The GPU box has no reference tree, so the vectors are committed.

How the reference is made importable on a CPU-only box without its CUDA extensions:
  * ``loguru`` -> tests/golden/_stubs/loguru.py;
  * ``third_party.pvcnn.functional.backend`` -> a module whose ``_backend`` is oracle.TorchBackend
    (the 12 pybind entry points restated on the CPU).  Everything ABOVE that boundary -- autograd
    wrappers, Voxelization, PVConv, SA/FP modules, PVCNN2Unet, priors, VAE, beta schedules,
    chamfer_python -- is the reference's own code running under PyTorch-CPU;
  * ``utils.model_helper`` / ``utils.utils`` (comet/wandb/CUDA-JIT imports) -> tiny stand-ins that
    provide ``import_model`` only.
Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import importlib
import json
import os
import sys
import types
import zlib

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "_stubs"), REF, ROOT]

import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle  # noqa: E402

# -- stand-ins registered before any reference import -------------------------------------------
bk = types.ModuleType("third_party.pvcnn.functional.backend")
bk._backend = oracle.TorchBackend()
sys.modules["third_party.pvcnn.functional.backend"] = bk
mh = types.ModuleType("utils.model_helper")


def _import_model(path):
    mod, cls = path.rsplit(".", 1)
    return getattr(importlib.import_module(mod), cls)


mh.import_model = _import_model
mh.loss_fn = None
sys.modules["utils.model_helper"] = mh
sys.modules["utils.utils"] = types.ModuleType("utils.utils")


def fill_(module, seed=0):
    """Deterministic weights derived from the parameter NAME: the test side regenerates them without
    shipping tensors.  Scales keep activations O(1) through ~30 layers."""
    with torch.no_grad():
        for name, t in sorted(module.state_dict().items()):
            if not t.is_floating_point():
                continue
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + seed) & 0x7fffffff)
            r = torch.randn(t.shape, generator=g)
            if t.dim() >= 2:
                fan_in = t[0].numel()
                t.copy_(r * (0.8 / np.sqrt(fan_in)))
            elif name.endswith("norm.weight") or "normalize" in name and name.endswith("weight"):
                t.copy_(1.0 + 0.1 * r)
            elif name.endswith("emd.bias"):   # AdaGN: factor half around 1 (as the reference inits it), bias half around 0
                v = 0.1 * r
                v[: t.numel() // 2] += 1.0
                t.copy_(v)
            else:
                t.copy_(0.1 * r)


def ref_cfg(category="airplane", clip=False):
    from default_config import cfg as base
    c = base.clone()
    c.merge_from_file(os.path.join(REF, "config", f"{category}_prior_cfg.yml"))
    if clip:
        c.clipforge.enable = 1
        c.latent_pts.style_prior = "models.score_sde.resnet.PriorSEClip"
    return c


def keys_of(m):
    return {k: list(v.shape) for k, v in m.state_dict().items()}


def main():
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    out = {}

    # ---- (a) state_dict layouts ---------------------------------------------------------------
    cfg = ref_cfg()
    from models.latent_points_ada_localprior import PVCNN2Prior
    from models.score_sde.resnet import PriorSEDrop, PriorSEClip
    from models.vae_adain import Model as VAE
    local = PVCNN2Prior(cfg.sde, cfg.shapelatent.latent_dim, cfg)
    glob = PriorSEDrop(cfg.sde, cfg.latent_pts.style_dim, cfg)
    vae = VAE(cfg)
    ccfg = ref_cfg(clip=True)
    local_c = PVCNN2Prior(ccfg.sde, ccfg.shapelatent.latent_dim, ccfg)
    glob_c = PriorSEClip(ccfg.sde, ccfg.latent_pts.style_dim, ccfg)
    layouts = {"PVCNN2Prior": keys_of(local), "PriorSEDrop": keys_of(glob), "vae_adain.Model": keys_of(vae),
               "PVCNN2Prior_clip": keys_of(local_c), "PriorSEClip": keys_of(glob_c)}
    json.dump(layouts, open(os.path.join(HERE, "state_dict_layouts.json"), "w"))

    # ---- (b) whole-model forwards with name-derived weights ------------------------------------
    for m in (local, glob, vae, local_c, glob_c):
        fill_(m)
        m.eval()
    B = 1
    x_l = torch.from_numpy(rng.standard_normal((B, 8192, 1, 1)).astype(np.float32))
    x_g = torch.from_numpy(rng.standard_normal((2, 128, 1, 1)).astype(np.float32))
    style = torch.from_numpy(rng.standard_normal((B, 128, 1, 1)).astype(np.float32))
    clipf = torch.from_numpy(rng.standard_normal((2, 512)).astype(np.float32))
    t_l = torch.tensor([417.0] * B)
    t_g = torch.tensor([3.0, 980.0])
    with torch.no_grad():
        y_l = local(x=x_l, t=t_l, condition_input=style, clip_feat=None)
        y_g = glob(x=x_g, t=t_g, condition_input=None, clip_feat=None)
        y_lc = local_c(x=x_l, t=t_l, condition_input=style, clip_feat=clipf[:B])
        y_gc = glob_c(x=x_g, t=t_g, condition_input=None, clip_feat=clipf)
        pts = vae.sample(num_samples=B, decomposed_eps=[style.view(B, 128), x_l.view(B, 8192)])
    np.savez_compressed(os.path.join(HERE, "model_forward.npz"),
                        x_l=x_l.numpy(), x_g=x_g.numpy(), style=style.numpy(), clip=clipf.numpy(),
                        t_l=t_l.numpy(), t_g=t_g.numpy(), y_l=y_l.numpy(), y_g=y_g.numpy(),
                        y_lc=y_lc.numpy(), y_gc=y_gc.numpy(), vae_points=pts.numpy())

    # ---- (c) block-level forwards + backwards (training path) -----------------------------------
    from models import pvcnn2_ada as ref_ada
    blocks = {}
    pv = ref_ada.PVConv(16, 32, 3, 8, with_se=True, attention=True, dropout=0.0, cfg=cfg)
    sa = ref_ada.PointNetSAModule(32, 0.5, 16, 16, [32, 48], cfg=cfg)
    fp = ref_ada.PointNetFPModule(48 + 16, [32, 24], cfg=cfg)
    for m in (pv, sa, fp):
        fill_(m)
        m.train()  # training mode exercises inds/wgts + every backward kernel; dropout is 0
    feat = torch.from_numpy(rng.standard_normal((2, 16, 128)).astype(np.float32)).requires_grad_(True)
    coords = torch.from_numpy(rng.standard_normal((2, 3, 128)).astype(np.float32))
    sty = torch.from_numpy(rng.standard_normal((2, 128)).astype(np.float32))
    o_pv = pv((feat, coords, None, sty))[0]
    o_pv.square().sum().backward()
    blocks.update(pv_feat=feat.detach().numpy(), pv_coords=coords.numpy(), pv_style=sty.numpy(),
                  pv_out=o_pv.detach().numpy(), pv_gfeat=feat.grad.numpy().copy(),
                  pv_gconv0=pv.voxel_layers[0].weight.grad.numpy().copy())
    feat.grad = None
    o_sa, c_sa, _, _ = sa((feat, coords, None, sty))
    o_sa.square().sum().backward()
    blocks.update(sa_out=o_sa.detach().numpy(), sa_centers=c_sa.detach().numpy(), sa_gfeat=feat.grad.numpy().copy())
    cfeat = torch.from_numpy(rng.standard_normal((2, 48, 32)).astype(np.float32)).requires_grad_(True)
    o_fp = fp((coords, c_sa.detach(), cfeat, feat.detach(), None, sty))[0]
    o_fp.square().sum().backward()
    blocks.update(fp_cfeat=cfeat.detach().numpy(), fp_out=o_fp.detach().numpy(), fp_gcfeat=cfeat.grad.numpy().copy())
    np.savez_compressed(os.path.join(HERE, "blocks.npz"), **blocks)

    # ---- (d) Voxelization.forward (P1) of the reference under PyTorch-CPU -----------------------
    p1 = {}
    for N, r in [(2048, 32), (1024, 16), (256, 8)]:
        co = torch.from_numpy((rng.standard_normal((4, 3, N)) * 0.6 + 0.2).astype(np.float32))
        vox = ref_ada.Voxelization(r)
        _, norm = vox(None, co)
        p1[f"co_{N}_{r}"] = co.numpy()
        p1[f"norm_{N}_{r}"] = norm.numpy()
        p1[f"vox_{N}_{r}"] = torch.round(norm).to(torch.int32).numpy()
    np.savez_compressed(os.path.join(HERE, "p1_voxelization.npz"), **p1)

    # ---- (e) chamfer_python.distChamfer (the oracle of the reference's own unit_test.py) ---------
    from third_party.ChamferDistancePytorch import chamfer_python
    a = torch.rand(4, 100, 3)
    b = torch.rand(4, 200, 3)
    d1, d2, i1, i2 = chamfer_python.distChamfer(a, b)
    np.savez_compressed(os.path.join(HERE, "chamfer_python.npz"), a=a.numpy(), b=b.numpy(), d1=d1.numpy(),
                        d2=d2.numpy(), i1=i1.numpy().astype(np.int32), i2=i2.numpy().astype(np.int32))

    # ---- (f) diffusion constants + DDIM coefficients with the reference's expressions -----------
    from utils.diffusion import make_beta_schedule
    dif = {}
    for mode in ("linear", "cust", "quad", "warmup10", "const"):
        try:
            dif[f"betas_{mode}"] = make_beta_schedule(mode, 1e-4, 0.02, 1000).numpy()
        except Exception:
            pass
    betas = dif["betas_linear"]
    alphas = 1.0 - betas
    ab = np.cumprod(alphas)
    Alpha_bar = torch.from_numpy(ab).float()      # diffusion_pvd.py:124-140
    steps = sorted([int(np.floor(i * ((1000 - 1.0) / (1000 - 1.0)))) for i in range(1000)], reverse=True)
    coef = []
    for i, tau in enumerate(steps):               # diffusion_pvd.py:434-447, verbatim arithmetic
        if i == len(steps) - 1:
            alpha_next = torch.tensor(1.0)
            sigma = torch.tensor(0.0)
        else:
            alpha_next = Alpha_bar[steps[i + 1]]
            sigma = 1.0 * torch.sqrt((1 - alpha_next) / (1 - Alpha_bar[tau]) * (1 - Alpha_bar[tau] / alpha_next))
        s = torch.sqrt(alpha_next / Alpha_bar[tau])
        c = torch.sqrt(1 - alpha_next - sigma ** 2) - torch.sqrt(1 - Alpha_bar[tau]) * torch.sqrt(alpha_next / Alpha_bar[tau])
        coef.append([float(s), float(c), float(sigma)])
    dif["alpha_bars_f32"] = Alpha_bar.numpy()
    dif["ddim_coef_1000"] = np.array(coef, np.float64)
    np.savez_compressed(os.path.join(HERE, "diffusion_constants.npz"), **dif)
    # ---- (g) VAE style encoder (non-ada pvcnn2 blocks) forward + backward, N = 1024 ---------------
    from models.shapelatent_modules import PointNetPlusEncoder
    se = PointNetPlusEncoder(zdim=128, input_dim=3, args=cfg)
    fill_(se)
    se.train()
    for mod in se.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    xs = torch.from_numpy((rng.standard_normal((1, 1024, 3)) * 0.5).astype(np.float32))
    o = se(xs)
    wv = torch.from_numpy(rng.standard_normal((1, 128)).astype(np.float32))
    ((o['mu_1d'] * wv).sum() + (o['sigma_1d'] * wv).sum()).backward()
    names = ['layers.0.0.voxel_layers.0.weight', 'layers.0.1.voxel_layers.6.fc.0.weight',
             'layers.0.2.mlps.0.layers.0.weight', 'layers.1.0.voxel_layers.4.weight', 'mlp.weight']
    sed = dict(x=xs.numpy(), w=wv.numpy(), mu=o['mu_1d'].detach().numpy(), sigma=o['sigma_1d'].detach().numpy())
    P = dict(se.named_parameters())
    for n in names:
        sed['g_' + n] = P[n].grad.numpy()
    np.savez_compressed(os.path.join(HERE, "style_encoder.npz"), **sed)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
