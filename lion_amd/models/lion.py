"""``LION`` facade -- mirror of the reference's ``models/lion.py:17-91``: VAE + [global prior, local
prior], ``load_model`` (``dae_state_dict`` / ``vae_state_dict``) and ``sample``.

The reference's demo path drives ``diffusers.DDPMScheduler`` (external, not vendored, not installed
here).  ``DDPMSchedulerShim`` provides the three members ``LION.sample`` uses
(``set_timesteps``, ``timesteps``, ``step(...).prev_sample``) implementing the DDPM ancestral step.
The noise variance follows what diffusers 0.11.1 (env.yaml:155) does with the string the reference
passes: ``variance_type=cfg.ddpm.model_var_type`` = 'fixedlarge' (default_config.py:221) is NOT one of
diffusers' names ('fixed_large', 'fixed_small', ...), so ``DDPMScheduler._get_variance`` matches no
branch and returns the posterior variance beta_t (1-abar_{t-1})/(1-abar_t) unclamped -- i.e. the
demo path samples with the SMALL variance although the config says large.  The shim reproduces that
('fixed_large' -> beta_t and 'fixed_small' -> clamped posterior are available by their diffusers
names).  The mean is the in-tree posterior mean (utils/diffusion_pvd.py:475-486), algebraically equal
to diffusers' x0-based form; floats differ at rounding level -- parity at this boundary stays
"unpinned" by the reference (no test, dependency not vendored), see DESIGN.md / INTEGRATION.md.
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
    def __init__(self, diffusion: DiffusionDiscretized, variance_type: str = 'fixedlarge'):
        self.d = diffusion
        self.timesteps = None
        self.alphas_cumprod = diffusion._h_alpha_bars
        self.variance_type = variance_type
        self._post = diffusion._betas_post_init.detach().cpu()  # [t] = beta_t (1-abar_{t-1})/(1-abar_t), t >= 1

    def _noise_scale(self, t: int) -> float:
        """sqrt of the variance diffusers 0.11.1 ``_get_variance`` yields for this variance_type."""
        if self.variance_type == 'fixed_large':
            return float(torch.sqrt(self.d._h_betas[t]))
        var = self._post[t]
        if self.variance_type == 'fixed_small':
            var = torch.clamp(var, min=1e-20)
        return float(torch.sqrt(var))   # every other string, 'fixedlarge' included: no branch matches

    def set_timesteps(self, n, device='cuda'):
        T = self.d._diffusion_steps
        assert n == T, "the shim implements the full ancestral chain only"
        self.timesteps = list(range(T - 1, -1, -1))

    def step(self, noise_pred, t, x, generator=None):
        t = int(t)
        is0, k_outer, k_a, k_b, _ = self.d.ddpm_coefficients(t)
        scale = 0.0 if is0 else self._noise_scale(t)
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
        self.scheduler = DDPMSchedulerShim(self.diffusion, variance_type=cfg.ddpm.model_var_type)

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
            # the reference's loop (:55-70: prior forward + scheduler.step per timestep) as one graphed chain:
            # the scheduler's mean / variance rule feeds the chain's coefficient table (lion_amd/chain.py)
            x, _ = self.diffusion.run_denoising_diffusion(
                prior, num_samples, shp, condition_input=condition_input, clip_feat=clip_feat,
                keep_trajectory=False, noise_scale=self.scheduler._noise_scale)
            prior.eval()
            sampled.append(x)
            output_dict[key] = x
            if condition_input is None:
                condition_input = self.vae.global2style(x)
        output_dict['points'] = self.vae.sample(num_samples=num_samples, decomposed_eps=sampled)
        return output_dict
