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


def _zero(optimizer, averager):
    if averager is not None:
        averager.zero_grad()
    else:
        optimizer.zero_grad(set_to_none=True)   # autograd adopts the backward's gradient tensors: no fill, no accumulate kernels


def vae_forward_backward(vae, optimizer, x, step=0, averager=None, noisy_input=None, kl_weight=None):
    """zero_grad + get_loss + backward (hvae_trainer.py:90-140).  kl_weight: a 0-d device tensor overriding the
    annealed KL weight the model derives from `step` on the host (a captured step reads it from memory)."""
    vae.train()
    _zero(optimizer, averager)
    out = vae.get_loss(x, it=step, noisy_input=noisy_input, kl_weight=kl_weight)
    loss = out['loss'].mean()
    loss.backward()
    return loss.detach(), out


def vae_train_step(vae, optimizer, x, step=0, averager: BucketedGradAverager | None = None,
                   distributed=False, noisy_input=None, kl_weight=None):
    loss, out = vae_forward_backward(vae, optimizer, x, step, averager, noisy_input, kl_weight)
    _finish(list(vae.parameters()), averager, distributed)
    optimizer.step()
    return loss, out


def prior_forward_backward(vae, dae, diffusion, optimizer, x, averager=None, clip_feat=None):
    """frozen VAE encode -> q(eps_t | eps) -> both denoisers -> MSE to the noise -> backward
    (train_2prior.py:195-410, pvd_mse_loss path)."""
    vae.eval()
    dae.train()
    B = x.shape[0]
    with torch.no_grad():
        eps = vae.encode(x)[0]                                   # [B, 128 + N*(3+D)]
    _zero(optimizer, averager)
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
    return loss.detach(), [l.detach() for l in losses]


def prior_train_step(vae, dae, diffusion, optimizer, x, averager: BucketedGradAverager | None = None,
                     distributed=False, clip_feat=None):
    """dae: ModuleList [global prior, local prior]; the VAE is frozen (cfg.sde.train_vae = False)."""
    loss, losses = prior_forward_backward(vae, dae, diffusion, optimizer, x, averager, clip_feat)
    _finish(list(dae.parameters()), averager, distributed)
    optimizer.step()
    return loss, losses


def _detached(obj):
    """the auxiliary outputs of a step without their autograd history (a captured step must not keep the graph of the
    previous pass alive: its buffers belong to the capture's memory pool)"""
    if torch.is_tensor(obj):
        return obj.detach()
    if isinstance(obj, dict):
        return {k: _detached(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_detached(v) for v in obj)
    return obj


class GraphedTrainStep:
    """One data-parallel training step -- zero_grad, forward, backward, gradient averaging, optimizer step -- as hipGraph
    replays, at any world size.  Eagerly a VAE step is ~7.8 k kernel launches and the GPU idles 80 % of the time behind
    the host; the reference's trainers (hvae_trainer.py:90-154, train_2prior.py:195-410 + utils/utils.py:717-748) are
    launch-bound the same way.

      forward_backward(**inputs) -> (loss, aux): zero_grad + forward + backward on STATIC input tensors (the caller's
          batch is copied into them before every replay); `vae_forward_backward` / `prior_forward_backward` bound to
          their model, optimizer and averager.
      mode 'whole' (world 1, or a backend whose collectives are stream-capturable: nccl = RCCL): ONE graph holds the
          step including the bucket all-reduces the averager's hooks launch from its side stream during the backward
          (they become a parallel branch of the graph) and the optimizer step.
      mode 'split' (gloo, or LION_TRAIN_GRAPH=split): [zero_grad + forward + backward] graph -> the buckets averaged
          eagerly (`averager.reduce_all()`) -> [optimizer step] graph.  No overlap of communication with the backward,
          but no per-kernel launch cost either.
      mode 'eager' (LION_TRAIN_GRAPH=off, or capture failed: `self.launch` says why): the plain step.
    The optimizer must be capturable (torch.optim.Adam(..., capturable=True)); scalars that change from step to step
    (annealed KL weight) are 0-d device tensors in `inputs`, refreshed with `set_scalar`.
    Construction runs `warmup` eager steps and one or two more for the capture on whatever `inputs` holds -- real
    updates, Adam moments and step counters included.  With `restore=True` (default) the parameters and the optimizer
    state (and the buffers of `modules`, if given) are snapshotted first and written back IN PLACE afterwards (the captured graph keeps pointing at the same
    state tensors), so the caller's first step starts from the weights, moments and step count it handed in, exactly as
    the eager / reference trainer would (hvae_trainer.py:90-154); `restore=False` keeps the consumed steps.
    The gradient layout is frozen
    at capture: parameters that received no gradient in the captured step are skipped by every replay (their
    ``grad`` stays None, as the reference's averaging leaves them)."""

    def __init__(self, forward_backward, inputs: dict, params, optimizer, averager=None, mode=None, warmup=3,
                 restore=True, modules=()):
        import os
        import torch.distributed as dist
        self.fb, self.inputs, self.params = forward_backward, dict(inputs), list(params)
        self.opt, self.avg = optimizer, averager
        # module BUFFERS the warm-up / capture steps may mutate (running statistics, counters): rewound with the parameters.
        # The released LION networks have none (GroupNorm everywhere); a caller whose model does passes `modules=[model]`.
        self._buffers = [b for m in modules for b in m.buffers()]
        world = dist.get_world_size() if dist.is_initialized() else 1
        backend = dist.get_backend() if dist.is_initialized() else None
        if mode is None:
            mode = os.environ.get("LION_TRAIN_GRAPH", "auto")
        if mode == "auto":
            mode = "whole" if (world == 1 or backend == "nccl") else "split"
        self.mode, self.world = mode, world
        self.loss = self.aux = None
        self._graphs = []
        dev = next(t.device for t in self.inputs.values() if torch.is_tensor(t))
        snapshot = self._snapshot() if restore else None
        try:
            self._build(mode, dev, world, backend, warmup)
        finally:
            if snapshot is not None:
                self._restore(snapshot)
                torch.cuda.synchronize(dev) if dev.type == "cuda" else None

    def _opt_params(self):
        seen, out = set(), []
        for group in self.opt.param_groups:
            for p in group['params']:
                if id(p) not in seen:
                    seen.add(id(p))
                    out.append(p)
        for p in self.params:
            if id(p) not in seen:
                seen.add(id(p))
                out.append(p)
        return out

    def _snapshot(self):
        """parameters and optimizer state as they are handed in (clones; `None` marks state the optimizer had not created yet)"""
        snap = []
        for p in self._opt_params():
            st = self.opt.state.get(p, None)
            snap.append((p, p.detach().clone(),
                         None if not st else {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}))
        self._buffer_snapshot = [b.detach().clone() for b in self._buffers]
        return snap

    def _restore(self, snap):
        """write the snapshot back THROUGH the live tensors: the captured graphs hold their addresses.  State that did not
        exist before construction (lazily created by the warm-up steps) goes back to its initial value, zero."""
        with torch.no_grad():
            for b, value in zip(self._buffers, getattr(self, "_buffer_snapshot", [])):
                b.copy_(value)
            for p, value, st in snap:
                p.copy_(value)
                live = self.opt.state.get(p, None)
                if not live:
                    continue
                for k, v in live.items():
                    if torch.is_tensor(v):
                        if st is not None and torch.is_tensor(st.get(k)):
                            v.copy_(st[k])
                        else:
                            v.zero_()
                    elif st is not None and k in st:
                        live[k] = st[k]
                    elif isinstance(v, (int, float)):
                        live[k] = type(v)(0)
        _wcache.invalidate_all()   # packed-weight caches key on parameter versions: the copies above bumped them

    def _build(self, mode, dev, world, backend, warmup):
        for _ in range(max(warmup, 1)):          # eager steps: optimizer state, packed-weight caches, kernel attributes
            self._eager()
        torch.cuda.synchronize(dev)
        if mode in ("off", "eager"):
            self.mode, self.launch = "eager", "eager (requested)"
            return
        try:
            if mode == "whole":
                self._capture_whole(dev)
                self.launch = ("hipGraph replay of the whole step (forward, backward, %s, optimizer)"
                               % ("bucket all-reduces inside the graph" if world > 1 else "gradient buckets"))
            else:
                self._capture_split(dev)
                self.launch = ("hipGraph replay of [forward + backward] -> eager bucket all-reduce (%s) -> hipGraph "
                               "replay of [optimizer step]" % backend)
        except Exception as e:   # capture is an optimisation: say so and run the eager step
            import sys
            import traceback
            traceback.print_exc(file=sys.stderr)
            torch.cuda.synchronize(dev)
            self._graphs = []
            self.mode, self.launch = "eager", f"eager (capture in mode '{mode}' failed: {type(e).__name__})"
            if self.avg is not None:   # an aborted capture leaves gradient views pointing into the dead graph pool
                self.avg.launch_in_hooks = True
                self.avg._reset()
                self.avg.zero_grad()

    # -- the three ways to run a step ---------------------------------------------------------------------------
    def _update(self):
        if self.avg is not None:
            self.avg.finish()
        elif self.world > 1:
            average_gradients(self.params, True)
        self.opt.step()

    def _eager(self):
        loss, aux = self.fb(**self.inputs)
        self.loss, self.aux = loss.detach(), _detached(aux)
        del loss, aux
        self._update()

    def _side_run(self, dev, fn):
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)

    def _capture_whole(self, dev):
        self._side_run(dev, self._eager)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._eager()
        self._graphs = [g]

    def _capture_split(self, dev):
        if self.avg is not None:
            self.avg.launch_in_hooks = False

        def fb():
            loss, aux = self.fb(**self.inputs)
            self.loss, self.aux = loss.detach(), _detached(aux)
            if self.avg is not None:
                self.avg.bind_all()   # the copies into the buckets belong to the replayed graph, not to the eager finish() below
        self._side_run(dev, lambda: (fb(), self._update()))
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            fb()
        # fix the gradient layout as the captured backward left it (untouched parameters: grad = None), outside any
        # capture; the all-reduce this performs runs on the warm-up gradients and is overwritten by the first replay
        if self.avg is not None:
            self.avg.finish()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, pool=g1.pool()):
            self.opt.step()
        self._graphs = [g1, g2]

    # -- public ----------------------------------------------------------------------------------------------------
    def set_scalar(self, name, value):
        self.inputs[name].fill_(float(value))

    def __call__(self, **batch):
        """copy `batch` (tensors by input name) into the static inputs, run one step, return the loss (a static 0-d
        tensor: read it before the next call)"""
        for k, v in batch.items():
            dst = self.inputs[k]
            if torch.is_tensor(dst):
                dst.copy_(v, non_blocking=True)
            else:
                raise KeyError(f"{k} is not a tensor input of this step")
        if self.mode == "eager":
            self._eager()
        elif self.mode == "whole":
            self._graphs[0].replay()
        else:
            self._graphs[0].replay()
            if self.avg is not None:
                self.avg.reduce_all()
            elif self.world > 1:
                average_gradients(self.params, True)
            self._graphs[1].replay()
        return self.loss


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
        # lion_amd.optim.Adam updates state[p]['ema'] inside its own launch: hand the decay over, keep the swap / state_dict side
        from .optim import Adam as _OwnAdam
        self._folded = isinstance(opt, _OwnAdam)
        if self._folded:
            opt.ema_decay = float(ema_decay)

    def step(self, *args, **kwargs):
        ret = self.optimizer.step(*args, **kwargs)
        if not self.apply_ema or self._folded:
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
