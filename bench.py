#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: shapes/sec @1000-step DDIM, B x 2048 pts (+ voxelize HBM GB/s).

Workload (BASELINE.json configs[1]): unconditional airplane prior sampling, B=32 x 2048 points per
GPU, the 1000-step DDIM chain of generate_samples_vada_2prior (trainers/train_2prior.py:50-127):
1000 denoiser steps of the global prior (PriorSEDrop, 77 M params) + 1000 of the local prior
(PVCNN2Prior, 14 PVConv / 4 SA / 4 FP per forward) + one VAE decode.  Weights are random-init of the
released architecture (no checkpoints offline), latents are synthetic N(0, I): data = "synthetic".

A "step" = one DDIM step of BOTH priors over the batch (model forward + fused update each), taken
from the head of the real 1000-step chain (t = 999, 998, ...).  --steps 1000 is the full chain.
    value = n_gpus * B / (1000 * ms_per_step / 1e3 + decode_seconds)        [shapes/s]
Each rank samples its own B shapes (independent units, no data-path collective): scaling = weak.

Extra objects on the JSON line:
  roofline      dominant kernel of the step (3x3x3 Conv3d 64->64 @32^3, 97 % of the FLOPs) timed
                with HIP events on the launch stream: achieved TFLOP/s vs the 157.3 TF fp32 MFMA peak;
  roofline_voxelize  the kernel the metric names: fused voxelize (64, 2048, 32), algorithmic bytes
                 / measured time vs 8 TB/s HBM;
  cpu_baseline  the same step on the host cores (PyTorch-CPU dense layers + the C oracle operators).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
MFMA_F32_PEAK_TF = 157.3    # fp32-input MFMA dense peak


def ev_time(fn, iters, warm=2):
    """average seconds per call, HIP events on torch's current stream (the launch stream)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def build_models(cfg, device):
    from lion_amd.models.lion import LION
    torch.manual_seed(0)
    lion = LION(cfg, device=device)
    lion.priors.eval()
    lion.vae.eval()
    return lion


def cpu_baseline(cfg, budget_s=25.0):
    """One DDIM step of both priors at B=2 on the host cores; bounded to ~budget_s of CPU work."""
    import numpy as np
    import oracle
    import lion_amd.functional.backend as bk
    from lion_amd.diffusion import DiffusionDiscretized
    from lion_amd.models import import_model
    from lion_amd.models.latent_points_ada_localprior import PVCNN2Prior
    saved = bk._backend
    bk._backend = oracle.TorchBackend()      # cpu_baseline leg only: the oracle is the thing TIMED here
    try:
        torch.manual_seed(0)
        B = 2
        glob = import_model(cfg.latent_pts.style_prior)(cfg.sde, cfg.latent_pts.style_dim, cfg).eval()
        local = PVCNN2Prior(cfg.sde, cfg.shapelatent.latent_dim, cfg).eval()
        d = DiffusionDiscretized(None, None, cfg, device="cpu")
        orc = oracle.lib()
        xg, xl = torch.randn(B, 128, 1, 1), torch.randn(B, 8192, 1, 1)
        style = torch.randn(B, 128, 1, 1)
        t = torch.full((B,), 1000.0)
        s, c, sg = d.ddim_coefficients(999, 998, 1.0)

        def step():
            with torch.no_grad():
                eg = glob(x=xg, t=t, condition_input=None, clip_feat=None)
                orc.ddim_update(xg.numpy(), eg.numpy(), np.random.standard_normal(xg.shape).astype(np.float32), s, c, sg)
                el = local(x=xl, t=t, condition_input=style, clip_feat=None)
                orc.ddim_update(xl.numpy(), el.numpy(), np.random.standard_normal(xl.shape).astype(np.float32), s, c, sg)

        step()  # warm-up (thread pools, allocator)
        t0 = time.perf_counter()
        n = 0
        while True:
            step()
            n += 1
            el_ = time.perf_counter() - t0
            if el_ > budget_s or n >= 20:
                break
        sec_per_step = el_ / n
        return {"value": B / (1000.0 * sec_per_step), "unit": "shapes/s", "cores": torch.get_num_threads(),
                "kind": "port",
                "sample": f"{n} DDIM steps (global+local prior forward + update) at B={B}x2048 on the host: "
                          f"{sec_per_step*1e3:.0f} ms/step; shapes/s = B/(1000*step), decode excluded; "
                          f"dense layers PyTorch-CPU ({torch.get_num_threads()} threads), point-voxel ops oracle/liboracle.so (OpenMP)"}
    finally:
        bk._backend = saved


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="DDIM steps timed; 1000 = the full real chain (~20 s)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="shapes per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-sparse", action="store_true",
                    help="run the first conv of every PVConv densely (no exact skip of all-zero input tiles)")
    args = ap.parse_args()
    if args.no_sparse:
        from lion_amd.models import pvcnn2_ada
        pvcnn2_ada.SPARSE_CONV1 = False

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", init_method="env://")

    from lion_amd import _lib
    _lib.load()  # fail loudly if the HIP extension is missing
    from lion_amd.config import released_prior_cfg
    from lion_amd import diffusion_ops
    from lion_amd.sampling import rank_seed

    cfg = released_prior_cfg("airplane")
    lion = build_models(cfg, dev)
    d = lion.diffusion
    B, K, W = args.batch, args.steps, args.warmup
    assert 1 <= K <= 1000
    torch.manual_seed(rank_seed(1234, rank))
    shapes = lion.vae.latent_shape()
    steps = d.ddim_schedule(1000, 1000, "uniform")  # 999 .. 0
    glob, local = lion.priors[0], lion.priors[1]

    def ddim_steps(model, x, cond, first, count):
        for i in range(first, first + count):
            t = steps[i]
            last = i == len(steps) - 1
            s, c, sg = d.ddim_coefficients(t, None if last else steps[i + 1], 1.0)
            ts = torch.full((B,), float(t + 1), device=dev)
            eps = model(x=x, t=ts, condition_input=cond, clip_feat=None).float().contiguous()
            z = torch.randn_like(x) if sg != 0.0 else None
            x = diffusion_ops.ddim_update(x, eps, z, s, c, sg)
        return x

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    with torch.no_grad():
        xg = torch.randn([B] + shapes[0], device=dev)
        xl = torch.randn([B] + shapes[1], device=dev)
        style = lion.vae.global2style(torch.randn([B] + shapes[0], device=dev))
        if not args.no_graph:
            # one hipGraph per denoiser (untimed, like any other one-off setup): ~740 launches per
            # local-prior forward are replayed without host launch overhead
            from lion_amd.graph import GraphedDenoiser
            t0_ = torch.full((B,), 1000.0, device=dev)
            try:
                glob_g = GraphedDenoiser(glob, xg, t0_, None)
                local_g = GraphedDenoiser(local, xl, t0_, style)
                glob, local = glob_g, local_g
            except Exception as e:  # a failed capture must not cost the measurement: run eagerly and say so
                print(f"bench: hipGraph capture failed ({e!r}); falling back to eager launches", file=sys.stderr, flush=True)
                torch.cuda.synchronize()
                args.no_graph = True
        # warm-up (untimed): W steps of each prior
        ddim_steps(glob, xg, None, 0, W)
        ddim_steps(local, xl, style, 0, W)
        sync_all()
        t0 = time.perf_counter()
        xg = ddim_steps(glob, xg, None, 0, K)          # K steps of the global chain
        style = lion.vae.global2style(xg)
        xl = ddim_steps(local, xl, style, 0, K)        # K steps of the local chain
        sync_all()
        elapsed = time.perf_counter() - t0
        # decode (once per 1000 steps), timed on its own
        lion.vae.sample(num_samples=B, decomposed_eps=[xg, xl])
        sync_all()
        t1 = time.perf_counter()
        pts = lion.vae.sample(num_samples=B, decomposed_eps=[xg, xl])
        sync_all()
        decode_s = time.perf_counter() - t1
    assert tuple(pts.shape) == (B, 2048, 3)

    if world > 1:
        tt = torch.tensor([elapsed, decode_s], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, decode_s = float(tt[0]), float(tt[1])
    ms_per_step = elapsed / K * 1e3
    value = world * B / (1000.0 * ms_per_step / 1e3 + decode_s)

    out = None
    if rank == 0:
        # ---- roofline: dominant kernel (Conv3d 64->64, 3^3, 32^3 grid, B=32: 29 of 59.7 GFLOP/shape) --
        from lion_amd.functional.backend import _backend as bk
        with torch.no_grad():
            conv = None
            for m in lion.priors[1].modules():
                if isinstance(m, torch.nn.Conv3d) and m.in_channels == 64 and m.out_channels == 64:
                    conv = m
                    break
            xin = torch.randn(B, 64, 32, 32, 32, device=dev)
            from lion_amd.conv_ops import conv3d_module
            tconv = ev_time(lambda: conv3d_module(conv, xin), 20, warm=5)
            flops = 2.0 * 27 * 64 * 64 * 32 ** 3 * B
            roof = {"kernel": "conv3d_k3_kernel: Conv3d 3x3x3 64->64 @32^3, B=32 (PVConv voxel branch; fp32-MFMA implicit GEMM, csrc/conv3d.hip)", "bound": "mfma",
                    "achieved": flops / tconv / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                    "frac": flops / tconv / 1e12 / MFMA_F32_PEAK_TF, "traffic": None,
                    "us_per_launch": tconv * 1e6}
            C, N, r = 64, 2048, 32
            co = torch.randn(B, 3, N, device=dev)
            ft = torch.randn(B, C, N, device=dev)
            tv = ev_time(lambda: bk.voxelize_points_forward(ft, co, r, True, 0.0), 20)
            vbytes = 4.0 * B * (3 * N + C * N + C * r ** 3 + N + r ** 3) + 4.0 * B * 3 * N
            roofv = {"kernel": "voxelize_points (P1+K1+K2) C=64 N=2048 r=32: vox_fused_kernel",
                     "bound": "hbm", "achieved": vbytes / tv / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": vbytes / tv / 1e9 / HBM_PEAK_GBS, "traffic": None, "us_per_call": tv * 1e6,
                     "algorithmic_bytes": vbytes}
            _, nc, _, _ = bk.voxelize_points_forward(None, co, r, True, 0.0)
            gridv = torch.randn(B, C, r ** 3, device=dev)
            td = ev_time(lambda: bk.trilinear_devoxelize_forward(r, False, nc, gridv), 20)
            dbytes = 4.0 * B * (3 * N + C * min(r ** 3, 8 * N) + C * N)       # 8 corners per point
            dphys = 4.0 * B * (3 * N + C * r ** 3 + C * N)                    # what moves: the grid is read once
            roofd = {"kernel": "trilinear_devoxelize C=64 N=2048 r=32: devox_slab_kernel (LDS-DMA slabs)",
                     "bound": "hbm", "achieved": dbytes / td / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": dbytes / td / 1e9 / HBM_PEAK_GBS, "traffic": dphys, "us_per_call": td * 1e6,
                     "algorithmic_bytes": dbytes,
                     "note": "traffic = bytes physically moved (whole grid read once): %.0f GB/s" % (dphys / td / 1e9)}
            try:  # HBM bytes per call from the separate rocprofv3 --pmc passes (cannot be collected live)
                tj = json.load(open(os.path.join(ROOT, "profiles", "r01_voxelize_traffic.json")))
                roofv["traffic"] = tj["hbm_bytes_per_call"]
            except Exception:
                pass
        out = {
            "metric": "shapes/sec @1000-step DDIM, Bx2048pts", "value": value, "unit": "shapes/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: unconditional airplane prior sampling, 1000-step DDIM chain "
                                   "(global PriorSEDrop + local PVCNN2Prior) + VAE decode",
                       "shapes_per_gpu": B, "points": 2048, "chain_steps": 1000,
                       "timed_steps_of_chain": K, "extrapolated": K != 1000, "decode_seconds": decode_s,
                       "parallelism": f"{world} independent rank(s), no data-path collective",
                       "launch": "eager" if args.no_graph else "hipGraph replay of each denoiser forward",
                       "conv1_empty_tile_skip": not args.no_sparse},
            "roofline": roof, "roofline_voxelize": roofv, "roofline_devoxelize": roofd,
        }
        if world > 1:  # the host baseline belongs to the 1-GPU line (other ranks would idle behind it)
            out["cpu_baseline"] = {"value": None, "unit": "shapes/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": "not timed at --gpus > 1; see the --gpus 1 line"}
        elif not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg)
            except Exception as e:  # the baseline must never take the benchmark down
                out["cpu_baseline"] = {"value": None, "unit": "shapes/s", "cores": os.cpu_count(),
                                       "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
