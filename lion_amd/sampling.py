"""Two-prior sampling driver -- mirror of ``generate_samples_vada_2prior``
(trainers/train_2prior.py:50-127): global prior -> style -> local prior -> VAE decode, and the
multi-GPU sharding of a generation job (independent shapes, no data-path collective; the reference
splits iterations over ranks but reseeds every rank identically, base_trainer.py:447-463 -- here the
rank is folded into the seed)."""
from __future__ import annotations

import torch


@torch.no_grad()
def generate_samples_vada_2prior(shape, dae, diffusion, vae, num_samples, enable_autocast=False,
                                 temp=1.0, ddim_step=0, clip_feat=None, ddim_skip_type='uniform',
                                 ddim_kappa=1.0, noise='device', step_callback=None, graph=True, given_noise=None,
                                 state_hook=None):
    """shape: vae.latent_shape(); dae: [global prior, local prior].  Returns (points [B,N,3], info).
    graph=True (default): every chain is replayed from one captured hipGraph per prior (lion_amd/chain.py);
    graph=False: the eager per-step loop; noise='cpu' draws the start and every step's noise from torch's CPU generator
    (one seed = one chain on any device: what the sampler-level parity test runs on the GPU and on the host);
    given_noise = [(start, [z per step]) per prior] replays recorded draws through the eager loop (DDIM only).
    state_hook(prior_index, step_index, x): may overwrite a chain's latent in place before a model evaluation (DDIM only;
    device-side launches, no host synchronisation)."""
    condition_input = None
    all_eps = []
    for i in range(len(dae)):
        if ddim_step > 0:
            eps, _ = diffusion.run_ddim(dae[i], num_samples, shape[i], temp, enable_autocast,
                                        is_image=False, ddim_step=ddim_step,
                                        condition_input=condition_input, clip_feat=clip_feat,
                                        skip_type=ddim_skip_type, kappa=ddim_kappa, noise=noise,
                                        keep_trajectory=False, graph=graph,
                                        given_noise=None if given_noise is None else given_noise[i],
                                        state_hook=None if state_hook is None else
                                        (lambda k, x, _i=i: state_hook(_i, k, x)))
        else:
            eps, _ = diffusion.run_denoising_diffusion(dae[i], num_samples, shape[i], temp,
                                                       enable_autocast, is_image=False,
                                                       condition_input=condition_input,
                                                       clip_feat=clip_feat, graph=graph, keep_trajectory=False)
        condition_input = eps
        if i == 0:
            condition_input = vae.global2style(condition_input)
        all_eps.append(eps)
        if step_callback is not None:
            step_callback(i)
    eps = vae.compose_eps(all_eps)
    info = {'print/sample_mean_global': eps.view(num_samples, -1).mean(-1).mean(),
            'print/sample_var_global': eps.view(num_samples, -1).var(-1).mean()}
    points = vae.sample(num_samples=num_samples, decomposed_eps=vae.decompose_eps(eps))
    return points, info


def shard_batch(total: int, rank: int, world: int) -> int:
    """shapes this rank generates: contiguous split, remainder to the low ranks."""
    return total // world + (1 if rank < total % world else 0)


def rank_seed(base_seed: int, rank: int, iteration: int = 0) -> int:
    return int(base_seed) + 1000003 * int(rank) + int(iteration)


def gather_samples(local_points: torch.Tensor, world: int):
    """all_gather of [B_local, N, 3] samples at the end of a generation job
    (trainers/base_trainer.py:484-487); the only communication of the sampling path."""
    import torch.distributed as dist
    if world == 1 or not dist.is_initialized():
        return local_points
    sizes = [torch.zeros(1, dtype=torch.int64, device=local_points.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local_points.shape[0]], dtype=torch.int64, device=local_points.device))
    mx = int(max(s.item() for s in sizes))
    pad = torch.zeros((mx,) + tuple(local_points.shape[1:]), dtype=local_points.dtype, device=local_points.device)
    pad[: local_points.shape[0]] = local_points
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[: int(s.item())] for b, s in zip(bufs, sizes)], dim=0)
