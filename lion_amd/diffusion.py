"""Discrete diffusion process -- mirror of the reference's ``utils/diffusion_pvd.py``
(``DiffusionDiscretized`` :17; constants :118-142; ``run_denoising_diffusion`` :224-303;
``run_ddim`` :390-473; ``get_q_posterior_mean`` :475-486) and ``utils/diffusion.py:28-65``
(``make_beta_schedule``).

What changes on MI355X: by default (``graph=True``, GPU tensors, eval) a chain is S replays of ONE captured
hipGraph [step prologue -> denoiser forward -> fused update with on-chip Philox noise], nothing per step comes
from the host (``lion_amd/chain.py``); the reference issues ~10 elementwise launches per step on top of the
~500 of the denoiser and draws the DDIM noise on the CPU, copying it over every step (:465-466).  The scalar
coefficients are computed exactly as the reference computes them -- float32 0-d tensor arithmetic on the same
float32 schedule -- so for identical (x, eps_hat, z) the update is bit-identical (``lion_ddim_update`` /
``lion_ddpm_update`` and the chain kernel share the arithmetic).  The noise STREAM differs by construction
(Philox on the device vs torch's CPU generator): ``graph=False, noise='cpu'`` reproduces the reference's stream
for seed-for-seed comparisons, ``graph=False`` alone the eager per-step loop.
"""
from __future__ import annotations

import numpy as np
import torch

from . import chain as _chain
from . import diffusion_ops


def make_beta_schedule(schedule, start, end, n_timestep):
    """float64 beta schedule (values as in utils/diffusion.py:28-65)."""
    if schedule == "cust":  # airplane PVD schedule: 10 % linear warm-up to beta_T, then constant
        betas = end * np.ones(n_timestep, dtype=np.float64)
        warm = int(n_timestep * 0.1)
        betas[:warm] = np.linspace(start, end, warm, dtype=np.float64)
    elif schedule == "quad":
        betas = torch.linspace(start ** 0.5, end ** 0.5, n_timestep, dtype=torch.float64).numpy() ** 2
    elif schedule == "linear":
        betas = torch.linspace(start, end, n_timestep, dtype=torch.float64).numpy()
    elif schedule in ("warmup10", "warmup50"):
        frac = 0.1 if schedule == "warmup10" else 0.5
        betas = end * np.ones(n_timestep, dtype=np.float64)
        warm = int(n_timestep * frac)
        betas[:warm] = np.linspace(start, end, warm, dtype=np.float64)
    elif schedule == "const":
        betas = end * np.ones(n_timestep, dtype=np.float64)
    elif schedule == "jsd":
        betas = 1.0 / np.linspace(n_timestep, 1, n_timestep, dtype=np.float64)
    else:
        raise NotImplementedError(schedule)
    return torch.from_numpy(np.asarray(betas, dtype=np.float64))


def _mixed(model, pred, mixing_component):
    """utils/utils.py:1299-1305."""
    if getattr(model, "mixed_prediction", False):
        coeff = torch.sigmoid(model.mixing_logit)
        return (1 - coeff) * mixing_component + coeff * pred
    return pred


class DiffusionDiscretized(object):
    """Constants and samplers of the discrete (DDPM) process; same constructor as the reference."""

    def __init__(self, args, var_fun, cfg, device="cuda"):
        self.cfg = cfg
        self.device = torch.device(device)
        self._diffusion_steps = cfg.ddpm.num_steps
        self._denoising_stddevs = 'beta'
        self.p2_gamma = cfg.ddpm.p2_gamma
        self.p2_k = cfg.ddpm.p2_k
        self.use_p2_weight = cfg.ddpm.use_p2_weight
        self.betas = make_beta_schedule(cfg.ddpm.sched_mode, cfg.ddpm.beta_1, cfg.ddpm.beta_T,
                                        cfg.ddpm.num_steps).numpy()
        # float64 numpy -> float32 (reference :118-142).  Host copies feed the scalar arithmetic.
        alphas = 1.0 - self.betas
        alpha_bars = np.cumprod(alphas)
        snr = 1.0 / (1 - alpha_bars) - 1
        betas_post = self.betas[1:] * (1.0 - alpha_bars[:-1]) / (1.0 - alpha_bars[1:])
        betas_post_init = np.append(betas_post[0], betas_post)
        f32 = lambda a: torch.from_numpy(a).float()
        self._h_betas, self._h_alphas, self._h_alpha_bars = f32(self.betas), f32(alphas), f32(alpha_bars)
        self._betas_init = self._h_betas.to(self.device)
        self.snr = f32(snr).to(self.device)
        self._alphas = self._h_alphas.to(self.device)
        self._alpha_bars = self._h_alpha_bars.to(self.device)
        self._betas_post_init = f32(betas_post_init).to(self.device)
        self._chains = _chain.ChainCache()

    # ---- training-side quantities (reference :45-113) ------------------------------------------
    def _iw(self, B, timestep):
        timestep = timestep + 1  # [1, T]
        alpha_bars = torch.gather(self._alpha_bars, 0, timestep - 1)
        weight_init = torch.sqrt(alpha_bars)[:, None, None, None]
        weight_noise_power = (1.0 - alpha_bars)[:, None, None, None]
        if self.use_p2_weight:
            loss_weight = torch.gather(1 / (self.p2_k + self.snr) ** self.p2_gamma, 0, timestep - 1).view(B)
        else:
            loss_weight = 1.0
        return timestep, weight_noise_power, weight_init, loss_weight, None, None

    def iw_quantities_t(self, B, timestep, *args):
        return self._iw(B, timestep.view(B))

    def iw_quantities(self, B, *args):
        rho = torch.rand(size=[B], device=self.device) * self._diffusion_steps
        return self._iw(B, rho.type(torch.int64))

    def sample_q(self, x_init, noise, var_t, m_t):
        assert len(x_init.shape) == 4 and len(var_t.shape) == 4 and len(m_t.shape) == 4
        assert x_init.shape[0] == m_t.shape[0]
        return m_t * x_init + torch.sqrt(var_t) * noise

    def cross_entropy_const(self, ode_eps):
        return 0

    def get_p_log_scales(self, timestep, stddev_type):
        if stddev_type == 'beta':
            return 0.5 * torch.log(torch.gather(self._betas_init, 0, timestep - 1))
        if stddev_type == 'beta_post':
            return 0.5 * torch.log(torch.gather(self._betas_post_init, 0, timestep - 1))
        if stddev_type == 'learn':
            return None
        raise ValueError('Unknown stddev_type: {}'.format(stddev_type))

    def get_mixing_component(self, x_noisy, timestep, enabled):
        if not enabled:
            return None
        alpha_bars = torch.gather(self._alpha_bars, 0, timestep - 1)
        return torch.sqrt(1.0 - alpha_bars).view(-1, 1, 1, 1) * x_noisy

    def mixing_component(self, eps, var, t, enabled):
        return self.get_mixing_component(eps, t, enabled)

    def get_q_posterior_mean(self, x_noisy, prediction, t):
        """torch formulation of :475-486 (the samplers below use the fused kernel instead)."""
        if t == 0:
            return 1.0 / torch.sqrt(self._alpha_bars[0]) * \
                (x_noisy - torch.sqrt(1.0 - self._alpha_bars[0]) * prediction)
        return 1.0 / torch.sqrt(self._alphas[t]) * \
            (x_noisy - self._betas_init[t] * prediction / torch.sqrt(1.0 - self._alpha_bars[t]))

    # ---- step coefficients: float32 0-d tensor arithmetic, exactly the reference's expressions ----
    def ddim_coefficients(self, t, t_next, kappa):
        """(s, c, sigma) of x <- x*s + c*eps + sigma*z for the step t -> t_next (t_next None = last).
        reference :434-447."""
        ab = self._h_alpha_bars
        if t_next is None:
            alpha_next = torch.tensor(1.0)
            sigma = torch.tensor(0.0)
        else:
            alpha_next = ab[t_next]
            sigma = kappa * torch.sqrt((1 - alpha_next) / (1 - ab[t]) * (1 - ab[t] / alpha_next))
        s = torch.sqrt(alpha_next / ab[t])
        c = torch.sqrt(1 - alpha_next - sigma ** 2) - torch.sqrt(1 - ab[t]) * torch.sqrt(alpha_next / ab[t])
        return float(s), float(c), float(sigma)

    def ddpm_coefficients(self, t):
        """(t_is_zero, k_outer, k_a, k_b, scale) for the ancestral step at t (reference :475-486, :265-266)."""
        ab, al, be = self._h_alpha_bars, self._h_alphas, self._h_betas
        scale = torch.exp(0.5 * torch.log(be[t]))
        if t == 0:
            return True, float(1.0 / torch.sqrt(ab[0])), float(torch.sqrt(1.0 - ab[0])), 1.0, float(scale)
        return False, float(1.0 / torch.sqrt(al[t])), float(be[t]), float(torch.sqrt(1.0 - ab[t])), float(scale)

    @staticmethod
    def ddim_schedule(diffusion_steps, S, skip_type='uniform'):
        """descending list of the visited timesteps (reference :408-419)."""
        if S == 1:
            tau = [0]   # (the reference's expression divides by S - 1)
        elif skip_type == 'uniform':
            c = (diffusion_steps - 1.0) / (S - 1.0)
            tau = [int(np.floor(i * c)) for i in range(S)]
        elif skip_type == 'quad':
            tau = [int(s) for s in list(np.linspace(0, np.sqrt(diffusion_steps * 0.8), S) ** 2)]
        else:
            raise NotImplementedError(skip_type)
        return sorted(tau, reverse=True)

    def ddim_table(self, steps, kappa):
        """[S, 8] float32 rows {t_model, s, c, sigma, 0...} of a DDIM chain over the descending timesteps `steps`: what a
        captured chain reads its per-step scalars from (lion_amd/chain.py)"""
        table = np.zeros((len(steps), 8), np.float32)
        for i, t in enumerate(steps):
            last = i == len(steps) - 1
            s_, c_, sigma_ = self.ddim_coefficients(t, None if last else steps[i + 1], kappa)
            table[i, :4] = (t + 1, s_, c_, sigma_)
        return table

    def _noise(self, size, source, device):
        if source == 'cpu':  # the reference's stream: torch.randn(size).to(device), :465-466
            return torch.randn(size).to(device)
        return torch.randn(size, device=device)

    # ---- samplers ------------------------------------------------------------------------------
    @torch.no_grad()
    def run_denoising_diffusion(self, model, num_samples, shape, temp=1.0, enable_autocast=False,
                                is_image=False, prior_var=1.0, condition_input=None, given_noise=None,
                                clip_feat=None, cls_emb=None, grid_emb=None, graph=True, keep_trajectory=True,
                                noise_scale=None):
        """Ancestral DDPM sampling, T model evaluations (reference :224-303).  ``noise_scale(t)`` overrides the
        per-step noise standard deviation sqrt(beta_t) (LION.sample's scheduler variance)."""
        model.eval()
        dev = self.device
        size = [num_samples] + list(shape)
        x_noisy = torch.randn(size=size, device=dev) if given_noise is None else given_noise[0].to(dev)
        x_noisy = x_noisy.contiguous()
        output_list = {'pred_x': []}
        kwargs = {'grid_emb': grid_emb} if grid_emb is not None else {}
        if cls_emb is not None:
            condition_input = cls_emb if condition_input is None else torch.cat([condition_input, cls_emb], dim=1)
        x_image = None
        if graph and given_noise is None and _chain.graphable(model, x_noisy, enable_autocast, kwargs):
            T = self._diffusion_steps
            table = np.zeros((T, 8), np.float32)
            for i, t in enumerate(reversed(range(T))):
                is0, k_outer, k_a, k_b, scale = self.ddpm_coefficients(t)
                if noise_scale is not None and not is0:
                    scale = float(noise_scale(t))
                table[i] = (t + 1, k_outer, k_a, k_b, scale, temp, 1.0 if is0 else 0.0, 0.0)
            ch = self._chains.get(model, num_samples, shape, condition_input, clip_feat, dev, _chain.DDPM, T)
            x_image = ch.run(x_noisy, table, _chain.draw_seed(), condition_input, clip_feat,
                             trajectory=output_list['pred_x'] if keep_trajectory else None,
                             trajectory_before_last=True)
            if is_image:
                x_image = (x_image.clamp(min=-1., max=1.) + 1.0) / 2.0
            model.train()
            return x_image, output_list
        for t in reversed(range(0, self._diffusion_steps)):
            timestep = torch.full((num_samples,), t + 1, dtype=torch.int64, device=dev)
            mixing = self.get_mixing_component(x_noisy, timestep, enabled=getattr(model, 'mixed_prediction', False))
            with torch.autocast("cuda", enabled=enable_autocast):
                pred = model(x=x_noisy, t=timestep.float(), condition_input=condition_input,
                             clip_feat=clip_feat, **kwargs)
                eps_hat = _mixed(model, pred, mixing).float().contiguous()
            is0, k_outer, k_a, k_b, scale = self.ddpm_coefficients(t)
            if is0:
                x_image = diffusion_ops.ddpm_update(x_noisy, eps_hat, None, True, k_outer, k_a, k_b, scale, temp)
            else:
                z = (torch.randn(size=size, device=dev) if given_noise is None
                     else given_noise[1][t].to(dev)).contiguous()
                x_noisy = diffusion_ops.ddpm_update(x_noisy, eps_hat, z, False, k_outer, k_a, k_b, scale, temp)
            output_list['pred_x'].append(x_noisy)
        if is_image:
            x_image = (x_image.clamp(min=-1., max=1.) + 1.0) / 2.0
        model.train()
        return x_image, output_list

    @torch.no_grad()
    def run_ddim(self, model, num_samples, shape, temp=1.0, enable_autocast=False, is_image=True,
                 prior_var=1.0, condition_input=None, ddim_step=100, skip_type='uniform', kappa=1.0,
                 clip_feat=None, grid_emb=None, x_noisy=None, dae_index=-1, noise='device',
                 keep_trajectory=True, graph=True, given_noise=None, state_hook=None):
        """DDIM sampling with ``ddim_step`` model evaluations; kappa is DDIM's eta (reference :390-473).
        noise='cpu': EVERY draw of the chain -- the start and the per-step noise -- comes from torch's CPU generator, so
        that one seed gives one chain on any device (the reference draws the per-step noise there, :465-466, and the
        start on its device).  given_noise = (start, [z_0, z_1, ...]): run the eager loop on exactly these draws (the
        counterpart of run_denoising_diffusion's given_noise; tests replay a graphed chain's recorded noise with it).
        state_hook(i, x): called before the i-th model evaluation with the chain's latent, which it may overwrite in place
        (device-side launches only; known-region replacement, bench.py's forced clouds)."""
        model.eval()
        dev = self.device
        size = [num_samples] + list(shape)
        if given_noise is not None:
            x_noisy, noise = given_noise[0], 'given'
        if x_noisy is None:
            x_noisy = torch.randn(size=size).to(dev) if noise == 'cpu' else torch.randn(size=size, device=dev)
        x_noisy = x_noisy.to(dev).contiguous()
        steps = self.ddim_schedule(self._diffusion_steps, ddim_step, skip_type)
        kwargs = {'grid_emb': grid_emb} if grid_emb is not None else {}
        output_list = []
        if graph and noise == 'device' and _chain.graphable(model, x_noisy, enable_autocast, kwargs):
            table = self.ddim_table(steps, kappa)
            ch = self._chains.get(model, num_samples, shape, condition_input, clip_feat, dev, _chain.DDIM,
                                  self._diffusion_steps)
            x_noisy = ch.run(x_noisy, table, _chain.draw_seed(), condition_input, clip_feat,
                             trajectory=output_list if keep_trajectory else None, state_hook=state_hook)
            model.train()
            return x_noisy, output_list
        for i, t in enumerate(steps):
            last = i == len(steps) - 1
            if last:
                assert t == 0
            if state_hook is not None:
                state_hook(i, x_noisy)
            timestep = torch.full((num_samples,), t + 1, dtype=torch.int64, device=dev)
            mixing = self.get_mixing_component(x_noisy, timestep, enabled=getattr(model, 'mixed_prediction', False))
            s, c, sigma = self.ddim_coefficients(t, None if last else steps[i + 1], kappa)
            with torch.autocast("cuda", enabled=enable_autocast):
                pred = model(x=x_noisy, t=timestep.float(), condition_input=condition_input,
                             clip_feat=clip_feat, **kwargs)
                eps_hat = _mixed(model, pred, mixing).float().contiguous()
            # the reference draws (and adds, scaled by sigma == 0) noise on the last step as well
            if noise == 'given':
                z = given_noise[1][i].to(dev).contiguous() if sigma != 0.0 else None
            else:
                z = self._noise(size, noise, dev).contiguous() if (sigma != 0.0 or noise == 'cpu') else None
            x_noisy = diffusion_ops.ddim_update(x_noisy, eps_hat, z if sigma != 0.0 else None, s, c, sigma)
            if keep_trajectory:
                output_list.append(x_noisy)
        model.train()
        return x_noisy, output_list
