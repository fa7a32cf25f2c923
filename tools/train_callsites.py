"""Which lines of lion_amd launch the small ATen kernels of a training step?  One eager step under torch.profiler with stacks;
every aten:: op with device time is attributed to the innermost lion_amd/ frame of its stack and summed per (call site, op):
launch counts and device microseconds.  usage: train_callsites.py [--mode vae|prior] [--B 32] [--rows 60]"""
import argparse, collections, os, sys
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd import training
from lion_amd.config import released_prior_cfg
from lion_amd.dist import BucketedGradAverager

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="vae"); ap.add_argument("--B", type=int, default=32); ap.add_argument("--rows", type=int, default=60)
ap.add_argument("--op", default=None, help="list EVERY call of this aten op (e.g. aten::copy_, aten::clone), device time or not: memcpy-backed copies carry none")
a = ap.parse_args()
dev = torch.device("cuda"); torch.manual_seed(0)
x = torch.randn(a.B, 2048, 3, device=dev)
if a.mode == "vae":
    from lion_amd.models.vae_adain import Model as VAE
    model = VAE(released_prior_cfg("chair")).to(dev).train(); params = list(model.parameters())
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.99), fused=True); av = BucketedGradAverager(params)
    step = lambda: training.vae_train_step(model, opt, x, step=0, averager=av, distributed=False)
else:
    from lion_amd.models.lion import LION
    lion = LION(released_prior_cfg("car"), device=dev); lion.vae.eval()
    for p_ in lion.vae.parameters(): p_.requires_grad_(False)
    model = lion.priors.train(); params = list(model.parameters())
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.99), fused=True); av = BucketedGradAverager(params)
    def step():
        training.prior_forward_backward(lion.vae, model, lion.diffusion, opt, x, averager=av); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
tot_n = tot_t = 0
for e in prof.events():
    dt = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
    if not e.name.startswith("aten::"): continue
    if a.op is not None:
        if e.name != a.op: continue
    elif dt <= 0: continue
    site = "?"
    for fr in (e.stack or []):
        if "lion_amd/" in fr and "torch/" not in fr:
            site = fr.split("lion_amd/")[-1].strip(); break
    if site == "?":   # launched by the autograd engine: name the backward node instead
        par = e.cpu_parent
        while par is not None:
            if par.name.startswith("autograd::engine::evaluate_function: "):
                site = "backward of " + par.name.split(": ", 1)[1]; break
            par = par.cpu_parent
    k = (site[:70], e.name)
    agg[k][0] += 1; agg[k][1] += dt; tot_n += 1; tot_t += dt
print(f"aten ops with device time: {tot_n} calls, {tot_t/1e3:.1f} ms")
print(f"{'call site':70s} {'op':28s} {'calls':>6s} {'ms':>8s}")
for (site, op), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.rows]:
    print(f"{site:70s} {op:28s} {n:6d} {t/1e3:8.2f}")
