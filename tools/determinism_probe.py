"""Is the product sampler run-to-run deterministic at B = 32, and if not, which stage / module is not?

Runs generate_samples_vada_2prior's stages (global chain, style, local chain, decode) twice from the same torch seed
(graphed path, the default) and compares every stage with torch.equal; then runs ONE eager forward of the local prior
and of the VAE decoder twice with forward hooks on every leaf module and reports the first module whose output differs
between the two passes.  Output: a small JSON report on stdout (tools/README.md)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd.config import released_prior_cfg   # noqa: E402
from lion_amd.models.lion import LION            # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(5)
lion = LION(released_prior_cfg("airplane"))
lion.priors.eval()
lion.vae.eval()
d, vae = lion.diffusion, lion.vae
sh = vae.latent_shape()
report = {"B": B, "K": K}


def stages(graph):
    torch.manual_seed(11)
    out = {}
    e0, _ = d.run_ddim(lion.priors[0], B, sh[0], 1.0, False, is_image=False, ddim_step=K, condition_input=None,
                       keep_trajectory=False, graph=graph)
    out["global_eps"] = e0.clone()
    style = vae.global2style(e0)
    out["style"] = style.clone()
    e1, _ = d.run_ddim(lion.priors[1], B, sh[1], 1.0, False, is_image=False, ddim_step=K, condition_input=style,
                       keep_trajectory=False, graph=graph)
    out["local_eps"] = e1.clone()
    eps = vae.compose_eps([e0, e1])
    out["points"] = vae.sample(num_samples=B, decomposed_eps=vae.decompose_eps(eps)).clone()
    return out


def diff(a, b):
    if torch.equal(a, b):
        return 0.0
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


with torch.no_grad():
    for graph in (True, False):
        runs = [stages(graph) for _ in range(3)]
        report["graph" if graph else "eager"] = {k: [diff(runs[i][k], runs[0][k]) for i in (1, 2)] for k in runs[0]}
    g, e = stages(True), stages(False)
    # the graphed chain draws Philox noise on the device, the eager loop torch.randn: only stage 0's START is shared

    def hooked(model, call):
        outs = []
        hs = []
        for name, m in model.named_modules():
            if not list(m.children()):
                hs.append(m.register_forward_hook(
                    lambda mod, inp, out, name=name: outs.append((name, (out[0] if isinstance(out, tuple) else out).detach().clone()))))
        y = call()
        for h in hs:
            h.remove()
        return y.clone(), outs

    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn([B] + sh[1], device="cuda", generator=gen)
    style = vae.global2style(torch.randn([B] + sh[0], device="cuda", generator=gen))
    tt = torch.full((B,), 500.0, device="cuda")
    lp = lion.priors[1]
    passes = [hooked(lp, lambda: lp(x=x, t=tt, condition_input=style, clip_feat=None).float()) for _ in range(3)]
    bad = []
    for p in passes[1:]:
        for (n0, a), (n1, b_) in zip(passes[0][1], p[1]):
            if not torch.equal(a, b_):
                bad.append((n0, diff(b_, a)))
                break
    report["local_prior_eager_forward"] = {"final": [diff(p[0], passes[0][0]) for p in passes[1:]], "first_bad_module": bad}
    eps = [torch.randn([B] + sh[0], device="cuda", generator=gen), torch.randn([B] + sh[1], device="cuda", generator=gen)]
    passes = [hooked(vae, lambda: vae.sample(num_samples=B, decomposed_eps=eps)) for _ in range(3)]
    bad = []
    for p in passes[1:]:
        for (n0, a), (n1, b_) in zip(passes[0][1], p[1]):
            if not torch.equal(a, b_):
                bad.append((n0, diff(b_, a)))
                break
    report["vae_decode_eager"] = {"final": [diff(p[0], passes[0][0]) for p in passes[1:]], "first_bad_module": bad}
    gp = lion.priors[0]
    xg = torch.randn([B] + sh[0], device="cuda", generator=gen)
    passes = [gp(x=xg, t=tt, condition_input=None, clip_feat=None).float().clone() for _ in range(3)]
    report["global_prior_eager_forward"] = [diff(p, passes[0]) for p in passes[1:]]
print(json.dumps(report, indent=1))
