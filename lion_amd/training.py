"""The two training steps of BASELINE.json configs 3-5, reduced to their arithmetic:
  vae_train_step    trainers/hvae_trainer.py:90-154  (get_loss -> backward -> gradient averaging -> step)
  prior_train_step  trainers/train_2prior.py:195-410 (frozen VAE encode -> q(eps_t|eps) -> both denoisers
                    -> MSE to the noise -> backward -> gradient averaging -> step), pvd_mse_loss path
                    (every released config) with optional CLIP feature (config 5).
Gradient averaging goes through lion_amd.dist (bucketed, overlapped RCCL all-reduce) instead of the
reference's single post-backward flat all-reduce.  Logging / LR schedules / snapshots stay with the
caller (out of scope).  ``EMA`` mirrors utils/ema.py:31-120 (EMA of the weights
kept in the optimizer state, swapped in for evaluation)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _wcache
from .dist import BucketedGradAverager, average_gradients


def _finish(params, averager, distributed):
    if averager is not None:
        averager.finish()
    elif distributed:
        average_gradients(params, True)


def vae_train_step(vae, optimizer, x, step=0, averager: BucketedGradAverager | None = None,
                   distributed=False, noisy_input=None):
    vae.train()
    if averager is not None:
        averager.zero_grad()
    else:
        optimizer.zero_grad(set_to_none=False)
    out = vae.get_loss(x, it=step, noisy_input=noisy_input)
    loss = out['loss'].mean()
    loss.backward()
    _finish(list(vae.parameters()), averager, distributed)
    optimizer.step()
    return loss.detach(), out


def prior_train_step(vae, dae, diffusion, optimizer, x, averager: BucketedGradAverager | None = None,
                     distributed=False, clip_feat=None):
    """dae: ModuleList [global prior, local prior]; the VAE is frozen (cfg.sde.train_vae = False)."""
    vae.eval()
    dae.train()
    B = x.shape[0]
    with torch.no_grad():
        eps = vae.encode(x)[0]                                   # [B, 128 + N*(3+D)]
    if averager is not None:
        averager.zero_grad()
    else:
        optimizer.zero_grad(set_to_none=False)
    t_p, var_t_p, m_t_p, _, _, _ = diffusion.iw_quantities(B)
    losses = []
    decomposed = [e.unsqueeze(-1).unsqueeze(-1) for e in vae.decompose_eps(eps)]
    for latent_id, e in enumerate(decomposed):
        noise = torch.randn_like(e)
        e_t = diffusion.sample_q(e, noise, var_t_p, m_t_p)
        if latent_id == 0:
            pred = dae[0](e_t, t_p.float(), x0=e, condition_input=None, clip_feat=clip_feat)
        else:
            cond = vae.global2style(decomposed[0])
            pred = dae[1](e_t, t_p.float(), x0=e, condition_input=cond, clip_feat=clip_feat)
        mix = diffusion.mixing_component(e_t, var_t_p, t_p, enabled=getattr(dae[latent_id], 'mixed_prediction', False))
        if mix is not None:
            coeff = torch.sigmoid(dae[latent_id].mixing_logit)
            pred = (1 - coeff) * mix + coeff * pred
        losses.append(F.mse_loss(pred.contiguous().view(B, -1), noise.view(B, -1), reduction='mean'))
    loss = sum(losses)
    loss.backward()
    _finish(list(dae.parameters()), averager, distributed)
    optimizer.step()
    return loss.detach(), [l.detach() for l in losses]


class EMA(torch.optim.Optimizer):
    """Optimizer wrapper keeping an exponential moving average of every parameter in
    ``state[p]['ema']`` (utils/ema.py:31-120); ``swap_parameters_with_ema`` exchanges weights and EMA
    (evaluation uses the EMA weights, trainers/train_prior.py:653-656)."""

    def __init__(self, opt, ema_decay):
        self.ema_decay = ema_decay
        self.apply_ema = ema_decay > 0.0
        self.optimizer = opt
        self.state = opt.state
        self.param_groups = opt.param_groups

    def step(self, *args, **kwargs):
        ret = self.optimizer.step(*args, **kwargs)
        if not self.apply_ema:
            return ret
        for group in self.optimizer.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                state = self.optimizer.state[p]
                if 'ema' not in state:
                    state['ema'] = p.data.clone()
                state['ema'].mul_(self.ema_decay).add_(p.data, alpha=1.0 - self.ema_decay)
        return ret

    def zero_grad(self, set_to_none=True):
        return self.optimizer.zero_grad(set_to_none=set_to_none)

    def state_dict(self):
        return self.optimizer.state_dict()

    def load_state_dict(self, sd):
        self.optimizer.load_state_dict(sd)
        self.state = self.optimizer.state
        self.param_groups = self.optimizer.param_groups

    def swap_parameters_with_ema(self, store_params_in_ema):
        if not self.apply_ema:
            return
        for group in self.optimizer.param_groups:
            for p in group['params']:
                if not p.requires_grad:
                    continue
                ema = self.optimizer.state[p].get('ema')
                if ema is None:
                    continue
                # written through the parameter (bumps p._version): packed-weight caches, style plans and
                # captured graphs key on it; `.data.copy_` would leave the HIP kernels on the old packed copies
                with torch.no_grad():
                    if store_params_in_ema:
                        tmp = p.detach().clone()
                        p.copy_(ema)
                        ema.copy_(tmp)
                    else:
                        p.copy_(ema)
        _wcache.invalidate_all()
