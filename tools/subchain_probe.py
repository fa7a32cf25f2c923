"""Does a B = 32 sampling step get shorter when the batch runs as K independent sub-batch chains on K streams?
One DDIM step at B = 1 takes 3.95 ms, at B = 32 6.4 ms: most of a step is latency (300 dependent launches), which a second,
independent chain could fill.  For K in {1, 2, 4}: K captured chains of the local prior at B / K shapes each (lion_amd/chain.py,
each with its own main + geometry stream), replayed step by step from one host thread; time per step over 40 steps = the
time for ALL 32 shapes.  Also the global prior (one hipGraph per step)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd import chain as _chain
from lion_amd.config import released_prior_cfg
from lion_amd.models.lion import LION

B = int(os.environ.get("B", "32"))
STEPS = int(os.environ.get("STEPS", "40"))
dev = torch.device("cuda", 0)
cfg = released_prior_cfg("airplane")
torch.manual_seed(0)
lion = LION(cfg, device=dev)
lion.priors.eval(); lion.vae.eval()
shapes = lion.vae.latent_shape()
d = lion.diffusion
sched = d.ddim_schedule(d._diffusion_steps, 1000, 'uniform')
table = np.zeros((STEPS, 8), np.float32)
for i in range(STEPS):
    s_, c_, sg_ = d.ddim_coefficients(sched[i], sched[i + 1], 1.0)
    table[i, :4] = (sched[i] + 1, s_, c_, sg_)
with torch.no_grad():
    style = lion.vae.global2style(torch.randn([B] + shapes[0], device=dev))
for which, model, shape, cond in (("local prior", lion.priors[1], shapes[1], style), ("global prior", lion.priors[0], shapes[0], None)):
    for K in (1, 2, 4):
        bs = B // K
        streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
        chains = []
        torch.cuda.synchronize()
        for k in range(K):
            with torch.cuda.stream(streams[k]):
                cs = None if cond is None else cond[k * bs:(k + 1) * bs].contiguous()
                ch = _chain.GraphedChain(model, bs, shape, cs, None, dev, _chain.DDIM, 1000)
                ch.prepare(torch.randn([bs] + shape, device=dev), table, 1234 + k, cs, None)
                chains.append(ch)
        torch.cuda.synchronize()
        times = []
        for rep in range(3):
            for k in range(K):
                with torch.cuda.stream(streams[k]):
                    chains[k].counter.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(STEPS):
                for k in range(K):
                    with torch.cuda.stream(streams[k]):
                        chains[k].replay()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) / STEPS * 1e3)
        ok = all(bool(torch.isfinite(c.x).all()) for c in chains)
        print(f"{which}: B = {B} as {K} chain(s) of {bs}: {min(times):.3f} ms per step for all {B} shapes (runs {[round(t, 3) for t in times]}), finite {ok}", flush=True)
        del chains
        torch.cuda.empty_cache()
