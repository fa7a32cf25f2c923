"""Voxelize on clumped clouds: (a) how crowded the voxels of the benchmark's own trajectory are (per voxelize call of a
20-step product chain: max points per voxel, occupied voxels, sum of squared counts), (b) the fused kernel's time at
(C, 2048, 32) on Gaussian, flat and clumped synthetic clouds (graph replay, HIP events).  Round 4: the in-step r = 32
voxelize calls ran 2.1x their stand-alone (Gaussian-cloud) time."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lion_amd.functional.backend as bkm   # noqa: E402
from lion_amd.functional.backend import _backend as bk   # noqa: E402

B = 32
rep = {}


def tgraph(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def clouds(kind, n, gen):
    x = torch.randn(B, 3, n, device="cuda", generator=gen)
    if kind == "flat":
        x[:, 1] *= 0.15; x[:, 2] *= 0.6
    elif kind.startswith("clump"):
        f = float(kind[5:])     # clumpF: all but 2 % of the points scaled by F (the outliers set the normalisation)
        k = int(n * 0.98)
        x[:, :, :k] *= f
    return x


if "--no-micro" not in sys.argv:
    gen = torch.Generator(device="cuda").manual_seed(0)
    micro = {}
    for C, N, r in ((4, 2048, 32), (64, 2048, 32), (128, 1024, 16)):
        for kind in ("gauss", "flat", "clump0.3", "clump0.1", "clump0.03"):
            co = clouds(kind, N, gen)
            ft = torch.randn(B, C, N, device="cuda", generator=gen)
            _, _, _, cnt = bk.voxelize_points_forward(ft, co, r, True, 0.0)
            us = tgraph(lambda: bk.voxelize_points_forward(ft, co, r, True, 0.0))
            micro[f"{C},{N},{r} {kind}"] = {"us": round(us, 1), "max_count": int(cnt.max()),
                                            "occupied_per_cloud": float((cnt > 0).sum() / B),
                                            "sum_c2_per_cloud": float((cnt.float() ** 2).sum() / B)}
            print(f"{C},{N},{r} {kind}", micro[f"{C},{N},{r} {kind}"], flush=True)
    rep["micro"] = micro

if "--no-traj" not in sys.argv:
    from lion_amd.config import released_prior_cfg
    from lion_amd.models.lion import LION
    from lion_amd.sampling import generate_samples_vada_2prior
    torch.manual_seed(0)
    lion = LION(released_prior_cfg("airplane")); lion.priors.eval(); lion.vae.eval()
    log = []

    class Rec:
        def __getattr__(self, name):
            f = getattr(bk, name)
            if name != "voxelize_points_forward":
                return f

            def w(features, coords, r, *a, **k):
                out = f(features, coords, r, *a, **k)
                cnt = out[3]
                log.append((0 if features is None else features.shape[1], coords.shape[2], int(r), int(cnt.max()),
                            float((cnt > 0).sum() / cnt.shape[0]), float((cnt.float() ** 2).sum() / cnt.shape[0])))
                return out
            return w
    K = int(os.environ.get("K", "20"))
    bkm._backend = Rec()
    try:
        torch.manual_seed(1234)
        with torch.no_grad():
            generate_samples_vada_2prior(lion.vae.latent_shape(), lion.priors, lion.diffusion, lion.vae, B, ddim_step=K, graph=False)
    finally:
        bkm._backend = bk
    per = 14
    rows = [x for x in log if x[1] in (2048,) and x[2] == 32]
    rep["trajectory"] = {"K": K, "calls": len(log),
                         "r32_calls (C, N, r, max_count, occupied, sum_c2) every 4th step": rows[::16][:40]}
    print(json.dumps(rep["trajectory"]))
json.dump(rep, open(os.environ.get("OUT", "/dev/stdout"), "w"), indent=1)
