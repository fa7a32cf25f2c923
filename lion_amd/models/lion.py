"""``LION`` facade -- mirror of the reference's ``models/lion.py:17-91``: VAE + [global prior, local
prior], ``load_model`` (``dae_state_dict`` / ``vae_state_dict``) and ``sample``.

The reference's demo path drives ``diffusers.DDPMScheduler`` (external, not vendored, not installed
here).  ``DDPMSchedulerShim`` provides the three members ``LION.sample`` uses
(``set_timesteps``, ``timesteps``, ``step(...).prev_sample``) implementing the DDPM ancestral step
with beta_t variance, i.e. exactly what the in-tree ``DiffusionDiscretized.run_denoising_diffusion``
does (utils/diffusion_pvd.py:224-303) -- parity at this boundary is "unpinned" by the reference
(its variance_type string 'fixedlarge' is not one diffusers knows), see DESIGN.md.
"""
from types import SimpleNamespace

import torch

from . import import_model
from .latent_points_ada_localprior import PVCNN2Prior as LocalPrior
from .vae_adain import Model as VAE
from ..checkpoint import load_prior_checkpoint
from ..diffusion import DiffusionDiscretized
from .. import diffusion_ops


class DDPMSchedulerShim:
    def __init__(self, diffusion: DiffusionDiscretized):
        self.d = diffusion
        self.timesteps = None
        self.alphas_cumprod = diffusion._h_alpha_bars

    def set_timesteps(self, n, device='cuda'):
        T = self.d._diffusion_steps
        assert n == T, "the shim implements the full ancestral chain only"
        self.timesteps = list(range(T - 1, -1, -1))

    def step(self, noise_pred, t, x, generator=None):
        t = int(t)
        is0, k_outer, k_a, k_b, scale = self.d.ddpm_coefficients(t)
        z = None if is0 else torch.randn(x.shape, device=x.device, generator=generator)
        out = diffusion_ops.ddpm_update(x.contiguous(), noise_pred.float().contiguous(), z, is0,
                                        k_outer, k_a, k_b, scale, 1.0)
        return SimpleNamespace(prev_sample=out)


class LION(object):
    def __init__(self, cfg, device='cuda'):
        self.device = torch.device(device)
        self.vae = VAE(cfg).to(self.device)
        GlobalPrior = import_model(cfg.latent_pts.style_prior)
        global_prior = GlobalPrior(cfg.sde, cfg.latent_pts.style_dim, cfg).to(self.device)
        local_prior = LocalPrior(cfg.sde, cfg.shapelatent.latent_dim, cfg).to(self.device)
        self.priors = torch.nn.ModuleList([global_prior, local_prior])
        self.diffusion = DiffusionDiscretized(None, None, cfg, device=self.device)
        self.scheduler = DDPMSchedulerShim(self.diffusion)

    def load_model(self, model_path):
        load_prior_checkpoint(model_path, self.priors, self.vae, map_location=self.device)
        print(f'INFO finish loading from {model_path}')

    @torch.no_grad()
    def sample(self, num_samples=10, clip_feat=None, save_img=False):
        """1000 ancestral steps of the global prior, 1000 of the local prior, one decode
        (reference :38-80)."""
        self.priors.eval()
        self.vae.eval()
        self.scheduler.set_timesteps(self.diffusion._diffusion_steps, device=self.device)
        latent_shape = self.vae.latent_shape()
        global_prior, local_prior = self.priors[0], self.priors[1]
        assert not local_prior.mixed_prediction and not global_prior.mixed_prediction
        output_dict, sampled = {}, []
        condition_input = None
        for prior, shp, key in ((global_prior, latent_shape[0], 'z_global'),
                                (local_prior, latent_shape[1], 'z_local')):
            x = torch.randn(size=[num_samples] + shp, device=self.device)
            for t in self.scheduler.timesteps:
                t_tensor = torch.full((num_samples,), t + 1, dtype=torch.int64, device=self.device)
                eps = prior(x=x, t=t_tensor.float(), condition_input=condition_input, clip_feat=clip_feat)
                x = self.scheduler.step(eps, t, x).prev_sample
            sampled.append(x)
            output_dict[key] = x
            if condition_input is None:
                condition_input = self.vae.global2style(x)
        output_dict['points'] = self.vae.sample(num_samples=num_samples, decomposed_eps=sampled)
        return output_dict
