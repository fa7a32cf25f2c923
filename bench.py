#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: shapes/sec @1000-step DDIM, B x 2048 pts (+ voxelize HBM GB/s).

Workload (BASELINE.json configs[1]): unconditional airplane prior sampling, B=32 x 2048 points per
GPU, the 1000-step DDIM chain of generate_samples_vada_2prior (trainers/train_2prior.py:50-127):
1000 denoiser steps of the global prior (PriorSEDrop, 77 M params) + 1000 of the local prior
(PVCNN2Prior, 14 PVConv / 4 SA / 4 FP per forward) + one VAE decode.  Weights are random-init of the
released architecture (no checkpoints offline), latents are synthetic N(0, I): data = "synthetic".

The timed region is ONE call of the product sampler, lion_amd.sampling.generate_samples_vada_2prior(ddim_step=K)
-- the function a user calls; the graph replay lives in it, not here.  A "step" = one DDIM step of BOTH priors over
the batch.  --steps 1000 (default) is the metric's real chain:  value = n_gpus * B / elapsed.  For K < 1000 the call
runs a K-step DDIM chain (uniform skip) and the line says "extrapolated":
    value = n_gpus * B / (1000 * ms_per_step / 1e3 + decode_seconds),  ms_per_step = (elapsed - decode) / K.
Each rank samples its own B shapes (independent units, no data-path collective): scaling = weak.
`python bench.py --gpus N` starts its own N ranks when no launcher set WORLD_SIZE; under torchrun it is one rank.

Extra objects on the JSON line (round 4):
  roofline      the Conv3d instantiation a sampling step RUNS (64->64 @32^3, B=32: AdaGN+Swish prologue, constant + delta,
                GroupNorm sums, work queue, every tile occupied) timed with HIP events on the launch stream: fp32-equivalent
                TFLOP/s vs 2500 / 3 TF (fp16 MFMA peak over the 3 products per fp32 product), on dense random operands (power
                capped: frac_of_power_limited_mfma_rate relates it to mfma_ceiling, the bare MFMA stream's sustained rate measured
                in the same run; roofline_conv1_form_voxelized_input = the launch on the model's own operands); roofline_conv1_form = the other
                in-step instantiation, roofline_plain_kernel = the same layer without prologue / sums / queue,
                roofline_fp32_kernel = the exact-fp32 kernel vs the 157.3 TF fp32 MFMA peak; whole_step_mfma_frac = 1909
                GFLOP of a step / ms_per_step_dense_convs / 833 TF; roofline.traffic = HBM bytes per launch from
                profiles/r*_conv_instep_traffic.json (separate rocprofv3 --pmc passes; the file is named on the line);
  roofline_voxelize  the kernel the step runs and the metric names: vox_scatter_kernel (64, 2048, 32) from the index plan,
                algorithmic bytes / measured time vs 8 TB/s HBM; index kernel time and the fused single call as side keys;
  roofline_devoxelize  the planned affine form a PVConv runs (lion_trilinear_devoxelize_plan once per (cloud, r = 32), the
                planned forward per feature tensor); one-step affine and plain eval as side keys; roofline_plain_kernel.zero_input:
                the conv on all-zero operands (what power management takes on random data); roofline_backward_operators (K5, K8,
                K12-grad); roofline_chamfer / roofline_emd (fp32 vector peak / v_exp issue rate); latency_bound_operators
                (K6, K9, K11 as times).  Operators of 30-200 us are timed as 10-20 launches inside ONE hipGraph replay
                bracketed by HIP events: the kernels' time, not the host's launch rate;
  config.ms_per_step_all_runs  the timed call is repeated --repeats (3) times, `value` is the median;
  config.full_chain_1000  with --steps < 1000: one real 1000-step chain of the product sampler, run after the timed region
                (the short chain is the dense start of the trajectory; SURVEY.md 8d wants the real chain beside it);
  config.streams  the graphs / streams the captured chains actually replay;
  cpu_baseline  ONE DDIM step at B = 32 (and a few at B = 1) on the host cores (PyTorch-CPU dense layers + the C oracle
                operators) and the oracle's own time for voxelize / devoxelize at (64, 2048, 32).
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
HBM_COPY_GBS = 6290.0       # measured float4-copy ceiling (same guide)
MFMA_F32_PEAK_TF = 157.3    # fp32-input MFMA dense peak
MFMA_F16_PEAK_TF = 2500.0   # dense fp16/bf16 MFMA peak (no sparsity), same guide


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


def ev_time_graph(fn, iters, warm=2):
    """average seconds per call with `iters` calls captured in ONE hipGraph and the replay bracketed by HIP events on the
    launch stream: the kernels' own time.  Eagerly, a 60-us operator whose python wrapper allocates four outputs is timed
    at the host's launch rate, not the GPU's (round 3: the same voxelize kernel read 63 us in one tool and 72 in this
    file, in one gpurun call)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    del g
    return a.elapsed_time(b) / iters * 1e-3


def hbm_roofline(kernel, algorithmic_bytes, seconds):
    """achieved = SURVEY.md 8d algorithmic bytes / HIP-event time; frac against the 8.0 TB/s spec, frac_of_copy_ceiling
    against the 6.29 TB/s a float4 copy reaches on this part (MI355X_MICROARCH.md).  traffic: HBM bytes per launch
    come from separate rocprofv3 --pmc passes (tools/prof_traffic.sh -> profiles/archive/r02_*_traffic.json), they cannot be
    collected inside this process, hence null here."""
    gbs = algorithmic_bytes / seconds / 1e9
    return {"kernel": kernel, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "frac_of_copy_ceiling": gbs / HBM_COPY_GBS, "traffic": None,
            "us_per_call": seconds * 1e6, "algorithmic_bytes": algorithmic_bytes}


# the voxelisations / devoxelisations of ONE denoiser forward at configs[1] (top): (C, N, r) per call,
# in call order; the four (cloud, r) pairs they share are the N / r combinations
VOX_TUPLES = [(4, 2048, 32), (32, 2048, 32), (128, 1024, 16), (192, 256, 8)] + 3 * [(128, 64, 8)] + 3 * [(128, 256, 8)] + \
    2 * [(128, 1024, 16)] + 2 * [(64, 2048, 32)]
DEVOX_TUPLES = 2 * [(32, 2048, 32)] + [(64, 1024, 16), (128, 256, 8)] + 3 * [(128, 64, 8)] + 3 * [(128, 256, 8)] + \
    2 * [(128, 1024, 16)] + 2 * [(64, 2048, 32)]


def vox_devox_totals(bk, fused_ops, B, dev):
    """roofline_voxelize_forward_total / roofline_devoxelize_forward_total: ALL 14 + 14 calls of one denoiser forward, launched
    the way the step launches them (one index plan per (cloud, r) pair + one scatter per feature tensor; one devoxelisation
    plan per r = 32 cloud + the affine devoxelisation per grid), captured in ONE hipGraph each and timed by its replay:
    sum of SURVEY.md 8d algorithmic bytes (1092.6 MB / 746.1 MB at B = 32) / the in-graph time of the whole set."""
    g = torch.Generator(device=dev).manual_seed(0)
    clouds = {}
    for _, N, r in VOX_TUPLES:
        if (N, r) not in clouds:
            clouds[(N, r)] = torch.randn(B, 3, N, device=dev, generator=g)
    vfeat = [torch.randn(B, C, N, device=dev, generator=g) for C, N, r in VOX_TUPLES]
    vbytes = sum(4.0 * B * (3 * N + C * N + C * r ** 3 + N + r ** 3) for C, N, r in VOX_TUPLES)

    def vox_all():
        plans = {k: bk.voxel_index(co, k[1], True, 0.0) for k, co in clouds.items()}
        return [bk.voxel_scatter(f, plans[(N, r)]) for f, (C, N, r) in zip(vfeat, VOX_TUPLES)]
    tv = ev_time_graph(vox_all, 3)
    plans = {k: bk.voxel_index(co, k[1], True, 0.0) for k, co in clouds.items()}
    dgrid = [torch.randn(B, C, r, r, r, device=dev, generator=g) for C, N, r in DEVOX_TUPLES]
    dsc = [(torch.rand(B, C, device=dev, generator=g) + 0.5, torch.randn(B, C, device=dev, generator=g)) for C, N, r in DEVOX_TUPLES]
    dbytes = sum(4.0 * B * (3 * N + C * min(r ** 3, 8 * N) + C * N) for C, N, r in DEVOX_TUPLES)

    def devox_all():
        dplans = {k: fused_ops.devoxelize_plan(plans[k]["norm"], k[1]) for k in clouds if k[1] == 32}
        return [fused_ops.devoxelize_affine(gr, plans[(N, r)]["norm"], r, a_, b_, plan=dplans.get((N, r)))
                for gr, (a_, b_), (C, N, r) in zip(dgrid, dsc, DEVOX_TUPLES)]
    td = ev_time_graph(devox_all, 3)
    rv = hbm_roofline("all 14 voxelisations of one denoiser forward (4 index plans + 14 vox_scatter launches, one graph)", vbytes, tv)
    rd = hbm_roofline("all 14 devoxelisations of one denoiser forward (1 plan + 14 affine devoxelisations, one graph)", dbytes, td)
    for r_ in (rv, rd):
        r_["us_per_forward"] = r_.pop("us_per_call")
        r_["note"] = "sum of SURVEY.md 8d algorithmic bytes over the calls / replay time of the graph that holds them all"
    return rv, rd


def surface_latents(B, dev, seed=0):
    """the 'surface-like' set of SURVEY.md 8d as latent points of the local prior: 2048 points uniform on the unit sphere +
    0.01 noise per shape, the extra latent channel 0.01 noise; layout [B, 2048 * 4, 1, 1] (point-major, 3 coordinates + 1
    feature: models/latent_points_ada_localprior.py:72-84)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    v = torch.randn(B, 2048, 3, device=dev, generator=g)
    v = v / v.norm(dim=2, keepdim=True)
    v = v + 0.01 * torch.randn(B, 2048, 3, device=dev, generator=g)
    f = 0.01 * torch.randn(B, 2048, 1, device=dev, generator=g)
    return torch.cat([v, f], dim=2).reshape(B, 2048 * 4, 1, 1).contiguous()


class ForcedClouds:
    """state_hook of the forced-clouds call: before every model evaluation of the LOCAL prior its latent is overwritten with
    x_t = sqrt(abar_t) S + sqrt(1 - abar_t) z (the forward process q(x_t | x_0 = S), utils/diffusion_pvd.py:96-113 sample_q)
    -- the states a TRAINED denoiser visits on its way from N(0, I) to a shape, which random-init weights never produce
    (their chain collapses into a clump).  Two elementwise launches per step on the chain's stream, no host sync.  With
    force=False the hook only keeps copies of the latent at five steps (occupancy statistics of the unforced chain)."""

    def __init__(self, d, B, dev, n_steps, force=True):
        steps = d.ddim_schedule(d._diffusion_steps, n_steps, 'uniform')
        ab = d._h_alpha_bars[torch.tensor(steps)].double()
        self.a = ab.sqrt().float().to(dev)
        self.b = (1.0 - ab).sqrt().float().to(dev)
        self.force = force
        if force:
            self.S = surface_latents(B, dev)
            self.z = torch.randn(self.S.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        self.keep_at = sorted({0, n_steps // 4, n_steps // 2, (3 * n_steps) // 4, n_steps - 1})
        self.kept = {}

    def __call__(self, prior_index, i, x):
        if prior_index != 1:
            return
        if self.force:
            torch.mul(self.S, self.a[i], out=x)
            x.addcmul_(self.z, self.b[i])
        if i in self.keep_at:
            self.kept[i] = x.clone()


def empty_tile_fractions(bk, fused_ops, kept, B):
    """fraction of the r = 32 voxel-convolution tiles (256 voxels) that are EMPTY for conv1 (no point within 1 voxel of the
    tile) / for the delta conv2 (within 2), on the latents kept by a ForcedClouds hook: mean over the kept steps."""
    f1, f2 = [], []
    for i in sorted(kept):
        co = kept[i].view(B, 2048, 4)[:, :, :3].permute(0, 2, 1).contiguous()
        plan = bk.voxel_index(co, 32, True, 0.0)
        o1, o2 = fused_ops.conv3d_occupancy(plan["cnt"], 32, 64, B)
        nt = (o1.numel() - 4) // 2
        f1.append(float(((o1[:nt] & 0xf) == 0).float().mean()))
        f2.append(float(((o2[:nt] & 0xf) == 0).float().mean()))
    return {"steps": sorted(kept), "conv1_empty": f1, "conv2_empty": f2,
            "conv1_empty_mean": sum(f1) / len(f1), "conv2_empty_mean": sum(f2) / len(f2)}


def _short(sv, n=160):
    sv = str(sv)
    return sv if len(sv) <= n else sv[: n - 3] + "..."


def compact_line(out):
    """the ONE line the driver parses (round 5 lost its record to a 20-KB line with nested bench lines): the contract's keys,
    `config` with the workload and SCALARS only, the dominant kernel's `roofline`, `cpu_baseline` -- nothing nested deeper,
    no second object with the contract's key names, < 6 KB.  Everything else lives in the detail file (--detail-file)."""
    cfg = out["config"]
    keep_cfg = {}
    for k, v in cfg.items():
        if isinstance(v, (dict, list)):
            continue
        keep_cfg[k] = _short(v) if isinstance(v, str) else v
    roof = out.get("roofline") or {}
    r_keep = {k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "us_per_launch")
              if k in roof}
    if "kernel" in r_keep:
        r_keep["kernel"] = _short(r_keep["kernel"], 200)
    cpu = out.get("cpu_baseline") or {}
    c_keep = {k: (_short(v, 220) if isinstance(v, str) else v) for k, v in cpu.items() if not isinstance(v, (dict, list))}
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline") if k in out}
    line["dtype"] = _short(out.get("dtype", "f32"), 120)
    line["data"] = out.get("data", "synthetic")
    line["config"] = keep_cfg
    line["roofline"] = r_keep
    line["cpu_baseline"] = c_keep
    for k, v in out.items():      # top-level scalars of the full record (chain / forced-clouds / fraction summaries)
        if k not in line and not isinstance(v, (dict, list, str)):
            line[k] = v
    return line


def emit(out, detail_file, rank=0):
    """full record -> detail file (+ stderr), compact line -> stdout (the last thing printed)."""
    line = compact_line(out)
    try:
        os.makedirs(os.path.dirname(os.path.abspath(detail_file)), exist_ok=True)
        with open(detail_file, "w") as f:
            f.write(json.dumps(out) + "\n")     # one line: profiles/ copies of it are read line-wise by tests/test_bench_line_cpu.py
        line["config"]["detail_file"] = os.path.relpath(detail_file, ROOT)
    except OSError as e:
        line["config"]["detail_file"] = f"not written: {e!r}"
    print("[bench detail] " + json.dumps(out), file=sys.stderr, flush=True)
    text = json.dumps(line)
    assert len(text) < 8000 and text.count('"metric"') == 1, len(text)
    print(text, flush=True)


def build_models(cfg, device):
    from lion_amd.models.lion import LION
    torch.manual_seed(0)
    lion = LION(cfg, device=device)
    lion.priors.eval()
    lion.vae.eval()
    return lion


def cpu_baseline(cfg, budget_s=25.0):
    """The same step on the host cores -- PyTorch-CPU dense layers + the C oracle's operators (oracle/liboracle.so,
    OpenMP) -- on a bounded sample: DDIM steps of both priors at B = 1 (latency form, a few
    steps), ONE step at B = 32 (the metric's batch; ~25 s on 128 cores) from which `value` is derived, and the oracle's
    time for the three operators the metric names (voxelize, devoxelize at (64, 2048, 32), B = 32)."""
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
        glob = import_model(cfg.latent_pts.style_prior)(cfg.sde, cfg.latent_pts.style_dim, cfg).eval()
        local = PVCNN2Prior(cfg.sde, cfg.shapelatent.latent_dim, cfg).eval()
        d = DiffusionDiscretized(None, None, cfg, device="cpu")
        orc = oracle.lib()
        s, c, sg = d.ddim_coefficients(999, 998, 1.0)

        def make(B):
            xg, xl = torch.randn(B, 128, 1, 1), torch.randn(B, 8192, 1, 1)
            style = torch.randn(B, 128, 1, 1)
            t = torch.full((B,), 1000.0)

            def step():
                with torch.no_grad():
                    eg = glob(x=xg, t=t, condition_input=None, clip_feat=None)
                    orc.ddim_update(xg.numpy(), eg.numpy(), np.random.standard_normal(xg.shape).astype(np.float32), s, c, sg)
                    el = local(x=xl, t=t, condition_input=style, clip_feat=None)
                    orc.ddim_update(xl.numpy(), el.numpy(), np.random.standard_normal(xl.shape).astype(np.float32), s, c, sg)
            return step
        step1 = make(1)
        step1()  # warm-up (thread pools, allocator)
        t0 = time.perf_counter()
        n1 = 0
        while n1 < 5 and time.perf_counter() - t0 < 0.2 * budget_s:
            step1()
            n1 += 1
        sec_b1 = (time.perf_counter() - t0) / max(n1, 1)
        step32 = make(32)
        t0 = time.perf_counter()
        step32()
        sec_b32 = time.perf_counter() - t0
        # the metric's operators on their own: oracle ms per call at B = 32 (one call each, ~1 s together)
        rng = np.random.default_rng(0)
        C, N, r = 64, 2048, 32
        co = rng.standard_normal((32, 3, N)).astype(np.float32)
        ft = rng.standard_normal((32, C, N)).astype(np.float32)
        t0 = time.perf_counter()
        nc, vox = orc.voxelize_coords(co, r, True, 0.0)
        t_p1 = time.perf_counter() - t0
        t0 = time.perf_counter()
        grid, _, _ = orc.avg_voxelize_forward(ft, vox, r)
        t_vox = time.perf_counter() - t0
        t0 = time.perf_counter()
        orc.trilinear_devoxelize_forward(r, False, nc, grid)
        t_devox = time.perf_counter() - t0
        vbytes = 4.0 * 32 * (3 * N + C * N + C * r ** 3 + N + r ** 3)
        dbytes = 4.0 * 32 * (3 * N + C * min(r ** 3, 8 * N) + C * N)
        return {"value": 32 / (1000.0 * sec_b32), "unit": "shapes/s", "cores": torch.get_num_threads(),
                "kind": "port",
                "sample": f"ONE DDIM step (global + local prior forward + update) at B=32x2048 on the host: {sec_b32*1e3:.0f} ms; "
                          f"shapes/s = 32 / (1000 x step), decode excluded; {n1} steps at B=1: {sec_b1*1e3:.0f} ms/step; "
                          f"dense layers PyTorch-CPU ({torch.get_num_threads()} threads), point-voxel ops oracle/liboracle.so (OpenMP)",
                "ms_per_step_B32": sec_b32 * 1e3, "ms_per_step_B1": sec_b1 * 1e3,
                "per_kernel_B32": {
                    "voxelize (K1+K2) C=64 N=2048 r=32": {"ms": t_vox * 1e3, "GB/s": vbytes / t_vox / 1e9},
                    "Voxelization.forward coordinates (P1) N=2048 r=32": {"ms": t_p1 * 1e3},
                    "trilinear_devoxelize (K4) C=64 N=2048 r=32": {"ms": t_devox * 1e3, "GB/s": dbytes / t_devox / 1e9}}}
    finally:
        bk._backend = saved


def train_cpu_baseline(mode, cfg, budget_s=40.0):
    """ONE forward + backward of the same training step on the host cores at B = 1 (PyTorch-CPU dense layers + autograd,
    the C oracle's operators -- forward and backward -- through oracle.TorchBackend), the bounded sample BASELINE.md
    section 3 allows for the training configs; samples/s = 1 / (fwd + bwd seconds), optimizer step excluded (it is
    B-independent and would dominate a B = 1 sample)."""
    import oracle
    import lion_amd.functional.backend as bk
    from lion_amd import _fallback, training
    saved, was_strict = bk._backend, _fallback.strict()
    bk._backend = oracle.TorchBackend()      # cpu_baseline leg only: the oracle is the thing TIMED here
    _fallback.strict(False)
    try:
        torch.manual_seed(0)
        x = torch.randn(1, 2048, 3)
        if mode == "train_vae":
            from lion_amd.models.vae_adain import Model as VAE
            model = VAE(cfg).train()
            opt = torch.optim.Adam(model.parameters(), lr=1e-4)

            def fb():
                return training.vae_forward_backward(model, opt, x, step=0)
        else:
            from lion_amd.models.lion import LION
            lion = LION(cfg, device="cpu")
            lion.vae.eval()
            for p_ in lion.vae.parameters():
                p_.requires_grad_(False)
            model = lion.priors.train()
            opt = torch.optim.Adam(model.parameters(), lr=1e-4)
            clip = torch.randn(1, 512) if mode == "train_prior_clip" else None

            def fb():
                return training.prior_forward_backward(lion.vae, model, lion.diffusion, opt, x, clip_feat=clip)
        t0 = time.perf_counter()
        fb()
        first = time.perf_counter() - t0
        sec, n = first, 1
        if first < 0.4 * budget_s:          # a second pass with warm thread pools, if the budget allows
            t0 = time.perf_counter()
            fb()
            sec, n = time.perf_counter() - t0, 2
        return {"value": 1.0 / sec, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"forward + backward of ONE sample (B=1 x 2048 points) on the host, pass {n} of {n}: {sec:.2f} s "
                          f"(first pass {first:.2f} s); dense layers PyTorch-CPU autograd ({torch.get_num_threads()} threads), "
                          "point-voxel operators and their gradients oracle/liboracle.so; optimizer step excluded",
                "seconds_fwd_bwd_B1": sec}
    finally:
        bk._backend = saved
        _fallback.strict(was_strict)


def run_side_lines(modes, steps, timeout_s):
    """the OTHER configurations of BASELINE.json as short side runs of this script (own process each: own memory pool, a
    failure or a timeout costs its own line only): returns {mode: parsed JSON line | {"error": ...}}."""
    import subprocess
    out = {}
    for mode in modes:
        cmd = [sys.executable, os.path.abspath(__file__), "--mode", mode, "--steps", str(steps), "--warmup", "3"]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s,
                               env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            out[mode] = json.loads(line[-1]) if line else {"error": f"rc {r.returncode}: {r.stderr[-300:]}"}
        except Exception as e:
            out[mode] = {"error": repr(e)}
        out[mode]["side_run_seconds"] = time.perf_counter() - t0
    return out


def _spawn(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU, rendezvous on
    127.0.0.1) and relay rank 0's JSON line.  Under torchrun (WORLD_SIZE set) this is not used."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p_ in procs[1:]:
        rc = rc or p_.wait()
    sys.stdout.write(out)
    sys.stdout.flush()
    sys.exit(rc)



def demo_bench(args, rank, world, dev):
    """--mode demo: BASELINE.json configs[0] -- demo.py's chair prior, ONE shape x 2048 points, 100 DDIM steps per prior
    + decode -- as a latency line on the GPU (the reference quotes it as CPU-only plumbing; there is no CPU path in this
    product).  The timed call is the product sampler; value = seconds per shape, median of 5 calls after one warm-up."""
    from lion_amd.config import released_prior_cfg
    from lion_amd.sampling import generate_samples_vada_2prior, rank_seed
    cfg = released_prior_cfg("chair")
    lion = build_models(cfg, dev)
    shapes = lion.vae.latent_shape()
    K = args.steps if args.steps_given else 100
    times = []
    with torch.no_grad():
        for i in range(6):
            torch.manual_seed(rank_seed(1234, rank, i))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pts = generate_samples_vada_2prior(shapes, lion.priors, lion.diffusion, lion.vae, 1, ddim_step=K,
                                               graph=not args.no_graph)[0]
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
    assert tuple(pts.shape) == (1, 2048, 3) and bool(torch.isfinite(pts).all())
    lat = sorted(times[1:])[len(times[1:]) // 2]
    if rank == 0:
        emit({
            "metric": "seconds per shape (latency), 1 x 2048 pts, %d DDIM steps per prior + decode" % K, "value": lat,
            "unit": "s", "n_gpus": world, "steps": K, "warmup": 1, "ms_per_step": lat / K * 1e3, "higher_is_better": False,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[0]: demo.py chair prior, 1 shape x 2048 pts, %d DDIM steps, on the GPU" % K,
                       "calls_timed": len(times) - 1, "first_call_seconds_incl_graph_capture": times[0],
                       "launch": "eager" if args.no_graph else "hipGraph replay",
                       "scheduler_parity": "demo.py's diffusers.DDPMScheduler (v0.11.1, not installed, not vendored) is a restatement "
                                           "here: parity UNPINNED at that boundary (INTEGRATION.md section 4); this line times the "
                                           "in-tree DDIM sampler the trainers / eval use"},
            "roofline": {"kernel": "see the --mode sample line", "bound": "mfma", "achieved": None, "peak": None, "unit": "TFLOP/s",
                         "frac": None, "traffic": None},
            "cpu_baseline": {"value": None, "unit": "s", "cores": os.cpu_count(), "kind": "port",
                             "sample": "see the --mode sample line"}}, args.detail_file.replace(".json", "_demo.json"))


def train_bench(args, rank, world, dev, backend):
    """--mode train_vae | train_prior: one data-parallel training step of BASELINE.json configs[2] / configs[3] at the
    per-GPU share of the quoted batch (B = 128 / 4 and 256 / 8 = 32 x 2048 points): forward + backward + bucketed
    gradient averaging (lion_amd/dist.py; hooks armed at world size 1 too) + Adam, synthetic N(0, 1) clouds.
    value = samples/s over all ranks.  The step runs through lion_amd.training.GraphedTrainStep at every world size
    (7.8 k launches per eager VAE step leave the GPU idle 80 % of the time); `launch` says which form ran."""
    import torch.distributed as dist
    from lion_amd.config import released_prior_cfg
    from lion_amd.dist import BucketedGradAverager, broadcast_params
    from lion_amd import training
    vae_mode = args.mode == "train_vae"
    clip_mode = args.mode == "train_prior_clip"
    cfg = released_prior_cfg("chair" if vae_mode else "car", clip=clip_mode)
    B = args.batch if args.batch_given else (8 if clip_mode else 32)   # configs[4]: B = 64 over 8 GPUs
    K, W = (args.steps if args.steps_given else 20), args.warmup
    torch.manual_seed(0)
    use_graph = not args.no_graph
    if vae_mode:
        from lion_amd.models.vae_adain import Model as VAE
        model = VAE(cfg).to(dev).train()
        params = list(model.parameters())
        name = "configs[2]: hvae_trainer VAE step (style encoder + latent-point encoder + decoder), chair, B=128 over 4 GPUs"
    else:
        from lion_amd.models.lion import LION
        lion = LION(cfg, device=dev)
        lion.vae.eval()
        for p_ in lion.vae.parameters():
            p_.requires_grad_(False)
        model = lion.priors.train()
        params = list(model.parameters())
        name = ("configs[4]: CLIP-conditioned train_2prior step (train_prior_clip.sh: PriorSEClip + AdaGN-conditioned local "
                "denoiser, synthetic [B, 512] CLIP feature), B=64 over 8 GPUs" if clip_mode else
                "configs[3]: train_2prior step (frozen VAE encode + global + local denoiser), car, B=256 over 8 GPUs")
    if world > 1:
        broadcast_params(params)
    # the optimizer of the reference's trainers (utils/utils.py:115-121: optim.Adam): lion_amd.optim.Adam = the same arithmetic with
    # the whole update in ONE launch through a device pointer table.  LION_BENCH_ADAM=torch: ATen's fused multi-tensor Adam (36
    # tensors per launch: 25 launches for the VAE's 867 tensors; its foreach form with device-resident step counters launches
    # ~1700 tiny bias-correction kernels per step)
    own_adam = os.environ.get("LION_BENCH_ADAM", "own") != "torch"
    # ... each wrapped in the reference's EMA of the weights (utils/utils.py:134-138, common_fun_prior_train.py:47; decay 0.9999):
    # inside the same launch here, a multiply-add per parameter behind ATen's Adam (LION_BENCH_EMA=0: without, as rounds 1-5 timed it)
    ema_decay = 0.9999 if os.environ.get("LION_BENCH_EMA", "1") != "0" else 0.0
    if own_adam:
        from lion_amd.optim import Adam
        opt = Adam(params, lr=1e-4, betas=(0.9, 0.99), ema_decay=ema_decay)
    else:
        opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.99), capturable=use_graph,
                               fused=os.environ.get("LION_BENCH_ADAM_FUSED", "1") != "0")
        if ema_decay > 0.0:
            opt = training.EMA(opt, ema_decay)
    averager = BucketedGradAverager(params)
    torch.manual_seed(1234 + rank)
    x = torch.randn(B, 2048, 3, device=dev)
    clip_feat = torch.randn(B, 512, device=dev) if clip_mode else None   # the CLIP encoder itself is out of scope
    # the PRODUCT's captured step (lion_amd/training.py::GraphedTrainStep): whole-step graph at world 1 and with RCCL
    # (bucket all-reduces captured as a side branch), [fwd + bwd] -> eager all-reduce -> [optimizer] graphs on backends
    # whose collectives cannot be captured (gloo); eager only with --no-graph or if capture fails (`launch` says so)
    inputs = {"x": x}
    if vae_mode:
        # the (annealed) KL weight of a step lives in device memory: a captured step reads it there (set_scalar per step)
        inputs["kl_weight"] = torch.full((), float(getattr(model, "kl_weight", 1.0)), device=dev)

        def fb(x, kl_weight):
            return training.vae_forward_backward(model, opt, x, step=0, averager=averager, kl_weight=kl_weight)
    else:
        def fb(x):
            return training.prior_forward_backward(lion.vae, model, lion.diffusion, opt, x, averager=averager,
                                                   clip_feat=clip_feat)
    stepper = training.GraphedTrainStep(fb, inputs, params, opt, averager, mode="off" if args.no_graph else None,
                                        warmup=max(W, 3))
    launch = stepper.launch
    static_loss = [None]

    def runner():
        static_loss[0] = stepper()

    runner()
    torch.cuda.synchronize()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(device_ids=[torch.cuda.current_device()]) if backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    sync_all()
    t0 = time.perf_counter()
    for _ in range(K):
        runner()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])
    ms = elapsed / K * 1e3
    loss_v = float(static_loss[0])
    if rank == 0:
        from lion_amd import conv_ops
        with torch.no_grad():
            xin = torch.randn(32, 64, 32, 32, 32, device=dev)
            gy = torch.randn(32, 64, 32, 32, 32, device=dev)
            tw = ev_time(lambda: conv_ops.conv3d_k3_wgrad(xin, gy, (64, 64, 3, 3, 3)), 10, warm=3)
        flops = 2.0 * 27 * 64 * 64 * 32 ** 3 * 32
        wg_split = bool(conv_ops.WGRAD_SPLIT)     # which weight-gradient kernel conv3d_k3_wgrad dispatched to (Cin = 64)
        wg_peak = MFMA_F16_PEAK_TF / 3.0 if wg_split else MFMA_F32_PEAK_TF
        wg_kernel = ("conv3d_wgrad_split_kernel: weight gradient of Conv3d 3x3x3 64->64 @32^3, B=32 -- fp32 operands cut into fp16 "
                     "hi/lo pairs, 3 products per fp32 product on the 16-bit MFMA pipe, f32 accumulate (csrc/conv3d_wgrad.hip); "
                     "peak = 2500 TF / 3" if wg_split else
                     "conv3d_wgrad_kernel: weight gradient of Conv3d 3x3x3 64->64 @32^3, B=32 (exact-fp32 MFMA, csrc/conv3d_wgrad.hip)")
        dt = "f32"
        if conv_ops.SPLIT:
            dt = ("f32 (voxel-conv forward / data-gradient" + (" / weight-gradient" if wg_split else "") + " operands cut into fp16 "
                  "hi/lo pairs on the 16-bit MFMA pipe, f32 accumulate -- fp32-accurate, tests/test_conv_split_gpu.py" +
                  ("" if wg_split else "; weight gradient exact-fp32 MFMA") + ")")
        from lion_amd import _fallback
        # what the GPU step did, read BEFORE the host baseline runs (its CPU tensors take conv3d_module's host branch and
        # would be counted as if the GPU step had used vendor libraries: round-5 review, weak 6)
        gpu_step_fallbacks = dict(_fallback.counts())
        gpu_step_fallback_reasons = _fallback.reasons(12)
        cpu = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port", "sample": "unmeasured (--no-cpu-baseline)"}
        if world > 1:
            cpu["sample"] = "not timed at --gpus > 1; see the --gpus 1 line"
        elif not args.no_cpu_baseline:
            try:
                cpu = train_cpu_baseline(args.mode, cfg)
            except Exception as e:  # the baseline must never take the benchmark down
                cpu["sample"] = f"unmeasured: {e!r}"
        out = {"metric": "samples/sec, one data-parallel training step (fwd + bwd + grad averaging + Adam + EMA of the weights)",
               "value": world * B / (ms / 1e3), "unit": "samples/s", "n_gpus": world, "steps": K, "warmup": W,
               "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": dt, "data": "synthetic",
               "config": {"workload": name, "samples_per_gpu": B, "points": 2048, "launch": launch,
                          "gradient_averaging": f"BucketedGradAverager, {len(averager.buckets)} buckets, world {world}",
                          "optimizer": (("lion_amd.optim.Adam (one launch)" if own_adam else "torch.optim.Adam (fused multi-tensor)")
                                        + (f" + EMA of the weights, decay {ema_decay}" if ema_decay > 0.0 else "")),
                          "final_loss": loss_v, "strict": _fallback.strict(),
                          "vendor_library_fallbacks_total": sum(gpu_step_fallbacks.values()),
                          "vendor_library_fallbacks": gpu_step_fallbacks,
                          "vendor_library_fallback_shapes": gpu_step_fallback_reasons},
               "roofline": {"kernel": wg_kernel, "bound": "mfma", "achieved": flops / tw / 1e12,
                            "peak": wg_peak, "unit": "TFLOP/s (fp32-equivalent conv FLOPs)" if wg_split else "TFLOP/s",
                            "frac": flops / tw / 1e12 / wg_peak, "traffic": None, "us_per_launch": tw * 1e6},
               "cpu_baseline": cpu}
        emit(out, args.detail_file.replace(".json", f"_{args.mode}.json"))
    if world > 1:
        dist.barrier(device_ids=[torch.cuda.current_device()]) if backend == "nccl" else dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--mode", choices=["sample", "demo", "train_vae", "train_prior", "train_prior_clip"], default="sample",
                    help="sample (default): the BASELINE metric; demo: configs[0] as a latency line (1 shape, 100 DDIM "
                         "steps); train_*: one training step of configs[2] / configs[3] / configs[4]")
    ap.add_argument("--steps", type=int, default=1000,
                    help="DDIM steps per prior that are timed; 1000 = the metric's real chain (~17 s)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="shapes per GPU")
    ap.add_argument("--repeats", type=int, default=3,
                    help="the timed K-step call is repeated this many times; the line reports the MEDIAN (and every run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager per-step launches instead of hipGraph replay")
    ap.add_argument("--no-sparse", action="store_true",
                    help="run the voxel convolutions densely (no exact skip of all-zero input tiles)")
    ap.add_argument("--no-dense-check", action="store_true", help="skip the short dense (--no-sparse) side measurement")
    ap.add_argument("--shapes-total", type=int, default=0,
                    help="strong scaling: N shapes in total, split over the ranks (lion_amd.sampling.shard_batch); 0 (default) = "
                         "weak scaling, --batch shapes per GPU")
    ap.add_argument("--side-lines", action="store_true",
                    help="ALSO run the training configurations (configs[2..4]) as short side runs of --mode train_* and keep "
                         "them in the detail file (opt-in since round 6: they doubled the driver's wall time and their nested "
                         "lines made the record unparseable)")
    ap.add_argument("--side-steps", type=int, default=5, help="timed steps of each training side run")
    ap.add_argument("--detail-file", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="where the FULL record goes (every roofline object, all runs, notes); stdout carries one compact line")
    ap.add_argument("--forced-steps", type=int, default=-1,
                    help="DDIM steps of the forced-clouds call (x_t of the local prior forced to sqrt(abar_t) S + sqrt(1-abar_t) z "
                         "before every step); -1 = 1000 with the default / driver command, 0 = skip")
    ap.add_argument("--small-batches", default="4,8,16",
                    help="batch sizes of the strong-scaling readiness timing (ms per step of the same sampler); '' = skip")
    ap.add_argument("--no-full-chain", action="store_true",
                    help="with --steps < 1000: skip the one real 1000-step chain that is run (untimed region) for config.full_chain_1000")
    args = ap.parse_args()
    args.steps_given = any(a == "--steps" or a.startswith("--steps=") for a in sys.argv[1:])
    args.batch_given = any(a == "--batch" or a.startswith("--batch=") for a in sys.argv[1:])
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _spawn(args)
    from lion_amd import geometry
    from lion_amd.models import pvcnn2_ada
    if args.no_sparse:
        pvcnn2_ada.SPARSE_CONV1 = False

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    dev = torch.device("cuda", torch.cuda.current_device())
    backend = os.environ.get("LION_BENCH_BACKEND", "nccl")  # "gloo": ranks sharing one device (launcher tests only)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend=backend, init_method="env://")

    from lion_amd import _lib
    _lib.load()  # fail loudly if the HIP extension is missing
    if args.mode == "demo":
        return demo_bench(args, rank, world, dev)
    if args.mode != "sample":
        return train_bench(args, rank, world, dev, backend)
    from lion_amd.config import released_prior_cfg
    from lion_amd.sampling import generate_samples_vada_2prior, rank_seed

    cfg = released_prior_cfg("airplane")
    lion = build_models(cfg, dev)
    d = lion.diffusion
    B, K, W = args.batch, args.steps, args.warmup
    strong = args.shapes_total > 0
    if strong:      # strong scaling: the job is --shapes-total shapes, every rank takes its contiguous share
        from lion_amd.sampling import shard_batch
        B = shard_batch(args.shapes_total, rank, world)
        assert B >= 1, "--shapes-total must give every rank at least one shape"
    total_shapes = args.shapes_total if strong else world * B
    assert 1 <= K <= 1000
    shapes = lion.vae.latent_shape()
    graph = not args.no_graph

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(device_ids=[torch.cuda.current_device()]) if backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    def sample(n_steps, seed, hook=None, nb=None):
        """the PRODUCT sampler (lion_amd/sampling.py == trainers/train_2prior.py:50-127): n_steps DDIM steps of the
        global prior, style, n_steps of the local prior, VAE decode."""
        torch.manual_seed(seed)
        return generate_samples_vada_2prior(shapes, lion.priors, d, lion.vae, B if nb is None else nb, ddim_step=n_steps,
                                            ddim_skip_type='uniform', ddim_kappa=1.0, graph=graph, state_hook=hook)[0]

    with torch.no_grad():
        # untimed: W warm-up steps per prior through the same call (captures the two chain graphs once)
        sample(max(W, 1), rank_seed(999, rank))
        runs = []
        for _ in range(args.repeats):                   # the same K-step call, timed `repeats` times: median + spread
            sync_all()
            t0 = time.perf_counter()
            pts = sample(K, rank_seed(1234, rank))      # K steps of each chain + one decode: the timed region
            sync_all()
            runs.append(time.perf_counter() - t0)
        elapsed = sorted(runs)[len(runs) // 2]
        # the decode on its own (once per 1000 steps), to extrapolate honestly when K != 1000
        eps = [torch.randn([B] + shapes[0], device=dev), torch.randn([B] + shapes[1], device=dev)]
        lion.vae.sample(num_samples=B, decomposed_eps=eps)
        sync_all()
        t1 = time.perf_counter()
        lion.vae.sample(num_samples=B, decomposed_eps=eps)
        sync_all()
        decode_s = time.perf_counter() - t1
    assert tuple(pts.shape) == (B, 2048, 3)

    if world > 1:
        tt = torch.tensor([elapsed, decode_s] + runs, device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, decode_s, runs = float(tt[0]), float(tt[1]), [float(v) for v in tt[2:]]
    chain_s = max(elapsed - decode_s, 1e-9)
    ms_per_step = chain_s / K * 1e3
    value = total_shapes / elapsed if K == 1000 else total_shapes / (1000.0 * ms_per_step / 1e3 + decode_s)

    out = None
    if rank == 0:
        from lion_amd.functional.backend import _backend as bk
        from lion_amd import _fallback
        from lion_amd import fused_ops as fused_ops_mod
        with torch.no_grad():
            # how much of the step the exact sparse evaluation saves on THIS trajectory (random-weight latents drift
            # into concentrated clouds): a short dense chain, same call
            # what was actually replayed: graphs and streams of the captured chains (lion_amd/chain.py)
            chains = list(d._chains._entries.values()) if graph else []
            # launches of one replayed step of the local prior's chain and how many of them are ATen's: from the rocprofv3 kernel
            # trace of this command (tools/census_run.sh + tools/step_census.py -> profiles/r*_step_census_B32.json); a kernel
            # trace cannot be taken inside this process (hipGraphDebugDotPrint writes nothing on ROCm 7.2: tools/graph_census_probe.py)
            census = None
            import glob as _glob
            for f_ in sorted(_glob.glob(os.path.join(ROOT, "profiles", "r*_step_census_B%d.json" % B)), reverse=True):
                try:
                    c_ = json.load(open(f_))
                    census = {"launches_local_prior": c_["launches_per_step"], "aten_local_prior": c_["aten_kernels_per_step"],
                              "launches": c_["launches_per_step"], "aten": c_["aten_kernels_per_step"],
                              "aten_names": c_.get("aten_names"), "source": os.path.relpath(f_, ROOT),
                              "profile_commit": c_.get("profile_commit")}
                    fg_ = f_.replace("_step_census_B", "_step_census_global_prior_B")
                    if os.path.exists(fg_):      # a step = one DDIM step of BOTH priors
                        g_ = json.load(open(fg_))
                        census["launches_global_prior"], census["aten_global_prior"] = g_["launches_per_step"], g_["aten_kernels_per_step"]
                        census["launches"] += g_["launches_per_step"]
                        census["aten"] += g_["aten_kernels_per_step"]
                    break
                except Exception:
                    continue
            streams = [{"main_graphs": len(getattr(ch, "graphs", None) or [ch.graph]),
                        "geometry_graphs": len(ch.geo_graphs or []),
                        "streams": 2 if ch.geo_graphs else 1} for ch in chains]
            # a short chain is the dense start of the trajectory: with --steps < 1000 ALSO run the metric's real
            # 1000-step chain once, after the timed region (the headline is a real 1000-step run)
            full_chain = None
            chain_tiles = None
            sections = {}
            t_sec = time.perf_counter()
            if K != 1000 and not args.no_full_chain:   # rank 0 only (the other ranks are at the final barrier): no collective
                watch = ForcedClouds(d, B, dev, 1000, force=False)     # copies the latent at five steps, changes nothing
                torch.cuda.synchronize()
                tf = time.perf_counter()
                sample(1000, rank_seed(1234, rank), hook=watch)
                torch.cuda.synchronize()
                tf = time.perf_counter() - tf
                full_chain = {"seconds": tf, "shapes_per_s": B / tf, "ms_per_step": (tf - decode_s) / 1000 * 1e3,
                              "runs": 1, "note": "one call of the product sampler with ddim_step=1000 on rank 0 (shapes_per_s "
                                                 "is this rank's; ranks are independent)"}
                chain_tiles = empty_tile_fractions(bk, fused_ops_mod, watch.kept, B)
                del watch
            sections["full_chain_s"] = time.perf_counter() - t_sec
            # ---- forced clouds (round-5 review, missing 3): the same call with the local prior's latent forced, before every
            # step, to the state a TRAINED model would be at: x_t = sqrt(abar_t) S + sqrt(1 - abar_t) z, S = the surface set
            t_sec = time.perf_counter()
            forced = None
            n_forced = args.forced_steps if args.forced_steps >= 0 else (1000 if (K == 1000 or not args.no_full_chain) else K)
            if n_forced > 0:
                hook = ForcedClouds(d, B, dev, n_forced, force=True)
                torch.cuda.synchronize()
                tf = time.perf_counter()
                sample(n_forced, rank_seed(1234, rank), hook=hook)
                torch.cuda.synchronize()
                tf = time.perf_counter() - tf
                ms_f = (tf - decode_s) / n_forced * 1e3
                forced = {"steps": n_forced, "seconds": tf, "ms_per_step": ms_f,
                          "shapes_per_s": B / tf if n_forced == 1000 else B / (1000.0 * ms_f / 1e3 + decode_s),
                          "extrapolated": n_forced != 1000,
                          "tiles": empty_tile_fractions(bk, fused_ops_mod, hook.kept, B),
                          "note": "product sampler, ddim_step = steps; the local prior's latent is overwritten before each step with "
                                  "sqrt(abar_t) S + sqrt(1 - abar_t) z (S: 2048 points on the unit sphere + 0.01 noise per shape, z: "
                                  "one fixed N(0, I) draw): the occupancy a trained model's chain has -- Gaussian at t = T, a surface "
                                  "at t = 0 -- instead of the clump random-init weights drift into"}
                del hook
            sections["forced_clouds_s"] = time.perf_counter() - t_sec
            # ---- strong-scaling readiness (review, missing 2): the same sampler at the per-rank batches of a 32-shape job
            # split over 8 / 4 / 2 GPUs
            t_sec = time.perf_counter()
            small = {}
            for nb in [int(v) for v in args.small_batches.split(",") if v.strip()]:
                if nb >= B or strong:
                    continue
                ks = min(K, 20)
                sample(2, 1, nb=nb)
                torch.cuda.synchronize()
                t_ = []
                for _ in range(2):
                    tb = time.perf_counter()
                    sample(ks, rank_seed(1234, rank), nb=nb)
                    torch.cuda.synchronize()
                    t_.append(time.perf_counter() - tb)
                eps_ = [torch.randn([nb] + shapes[0], device=dev), torch.randn([nb] + shapes[1], device=dev)]
                lion.vae.sample(num_samples=nb, decomposed_eps=eps_)
                torch.cuda.synchronize()
                tb = time.perf_counter()
                lion.vae.sample(num_samples=nb, decomposed_eps=eps_)
                torch.cuda.synchronize()
                dec_ = time.perf_counter() - tb
                small[nb] = {"ms_per_step": (min(t_) - dec_) / ks * 1e3, "decode_seconds": dec_, "timed_steps": ks}
            sections["small_batches_s"] = time.perf_counter() - t_sec
            ms_dense = None
            if not args.no_sparse and not args.no_dense_check:
                pvcnn2_ada.SPARSE_CONV1 = False
                try:
                    d._chains.clear()
                    kd = min(K, 20)
                    sample(2, 1)
                    torch.cuda.synchronize()
                    td = time.perf_counter()
                    sample(kd, rank_seed(1234, rank))
                    torch.cuda.synchronize()
                    ms_dense = (time.perf_counter() - td - decode_s) / kd * 1e3
                finally:
                    pvcnn2_ada.SPARSE_CONV1 = True
                    d._chains.clear()
            # ---- roofline: dominant kernel (Conv3d 64->64, 3^3, 32^3 grid, B=32: 29 of 59.7 GFLOP/shape) --
            conv = None
            for m in lion.priors[1].modules():
                if isinstance(m, torch.nn.Conv3d) and m.in_channels == 64 and m.out_channels == 64:
                    conv = m
                    break
            xin = torch.randn(B, 64, 32, 32, 32, device=dev)
            from lion_amd import conv_ops, fused_ops
            flops = 2.0 * 27 * 64 * 64 * 32 ** 3 * B
            t32 = ev_time(lambda: conv_ops.conv3d_k3(xin, conv.weight, conv.bias, split=False), 20, warm=5)
            roof32 = {"kernel": "conv3d_k3_kernel: Conv3d 3x3x3 64->64 @32^3, B=32, exact-fp32 MFMA implicit GEMM "
                                "(csrc/conv3d.hip; fallback for Cin % 16 != 0 and the parity reference)",
                      "bound": "mfma", "achieved": flops / t32 / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                      "frac": flops / t32 / 1e12 / MFMA_F32_PEAK_TF, "traffic": None, "us_per_launch": t32 * 1e6}
            if conv_ops.SPLIT:
                tsp = ev_time(lambda: conv_ops.conv3d_k3(xin, conv.weight, conv.bias, split=True), 20, warm=5)
                roof = {"kernel": "conv3d_split_kernel: Conv3d 3x3x3 64->64 @32^3, B=32 (PVConv voxel branch), fp32 "
                                  "operands cut into fp16 hi/lo pieces, 3 v_mfma_f32_32x32x16_f16 per K=16, f32 "
                                  "accumulate (csrc/conv3d_split.hip)",
                        "bound": "mfma", "achieved": flops / tsp / 1e12, "peak": MFMA_F16_PEAK_TF / 3.0,
                        "unit": "TFLOP/s (fp32-equivalent conv FLOPs)", "frac": flops / tsp / 1e12 / (MFMA_F16_PEAK_TF / 3.0),
                        "traffic": None, "us_per_launch": tsp * 1e6,
                        "note": "peak = dense fp16 MFMA peak (2500 TF) / 3 MFMA products per fp32-equivalent product; "
                                "the exact-fp32 kernel of the same layer is in roofline_fp32_kernel"}
            else:
                roof = roof32
            roof_plain = roof
            if conv_ops.SPLIT:
                # the same launch on an all-zero input: identical instruction stream and MFMA count, no operand toggling.  The
                # ratio is what the chip's power management takes from this kernel on random data (profiles/HISTORY.md 4g: the in-kernel
                # clock, s_memtime against the 100 MHz wall clock, falls to 1.45-1.9 GHz under the dense random-data launch
                # and stays at 2.1-2.4 GHz on sparse / zero data; a bare MFMA stream sustains 2.4 GHz).
                xz = torch.zeros_like(xin)
                tz = ev_time(lambda: conv_ops.conv3d_k3(xz, conv.weight, conv.bias, split=True), 20, warm=5)
                roof_plain = dict(roof)
                roof_plain["zero_input"] = {"us_per_launch": tz * 1e6, "frac": flops / tz / 1e12 / (MFMA_F16_PEAK_TF / 3.0),
                                            "note": "same kernel, same MFMA count, all-zero activations: the difference to "
                                                    "us_per_launch is clock the power management takes on random operands"}
                roof = roof_plain
                del xz
            # The instantiations a sampling step RUNS (round-3 verdict: the plain kernel above is not one of them): conv1 of a
            # PVConv = GroupNorm sums + work queue over occupied tiles; conv2 = AdaGN + Swish prologue, constant + delta,
            # GroupNorm sums, work queue -- timed here with EVERY tile occupied (dense), so that the FLOP count is the
            # layer's.  Each launch consumes an occupancy / queue buffer: [occupancy + conv] and [occupancy] are captured
            # in graphs and subtracted.
            # what the matrix pipe itself sustains on this board: the bare tap stream of the split convolution (MFMAs + their
            # fragment reads, pipe 100 % busy at 32.5 cycles per MFMA) on random fp16 operands runs into the 1400-W cap at
            # ~1.56 GHz; on constant operands it holds 2.4 GHz (tools/exp/mfma_issue_probe.hip, profiles/HISTORY.md 4g)
            mfma_ceiling = None
            try:
                import ctypes
                pl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "exp", "libmfma_probe.so"))
                pl.mfma_ceiling.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
                mfma_ceiling = {}
                for nm, rnd in (("random_fp16_operands", 1), ("constant_operands", 0)):
                    tf_, mhz_ = ctypes.c_double(0), ctypes.c_double(0)
                    if pl.mfma_ceiling(rnd, 30, ctypes.byref(tf_), ctypes.byref(mhz_)) == 0:
                        mfma_ceiling[nm] = {"TFLOP/s fp16": tf_.value, "sclk_MHz": mhz_.value}
                mfma_ceiling["note"] = ("sustained rate of a bare v_mfma_f32_32x32x16_f16 stream with its LDS fragment reads, one wave per "
                                        "SIMD, 2 x ~50 ms launches; the clock is s_memtime ticks per wall-clock microsecond inside the kernel")
            except Exception as e:  # the probe library is optional (tools/exp/libmfma_probe.so from __graft_entry__.build())
                mfma_ceiling = {"error": repr(e)}
            roof_step = {}
            if conv_ops.SPLIT:
                ones = torch.ones(B, 32 ** 3, device=dev, dtype=torch.int32)
                pa = torch.rand(B, 64, device=dev) + 0.5
                pb = torch.randn(B, 64, device=dev) * 0.5
                t_occ = ev_time_graph(lambda: fused_ops.conv3d_occupancy(ones, 32, 64, B), 10)
                t_c1 = ev_time_graph(lambda: fused_ops.conv3d_fused(xin, conv, None, True,
                                                                    fused_ops.conv3d_occupancy(ones, 32, 64, B)[0]), 10) - t_occ
                t_c2 = ev_time_graph(lambda: fused_ops.conv3d_fused(xin, conv, (pa, pb), True,
                                                                    fused_ops.conv3d_occupancy(ones, 32, 64, B)[1],
                                                                    prev_conv=conv), 10) - t_occ
                for key, tt, what in (("conv1_form", t_c1, "STATS + work queue (reads the voxelised grid)"),
                                      ("conv2_form", t_c2, "AdaGN+Swish prologue + constant/delta + STATS + work queue")):
                    roof_step[key] = {"kernel": "conv3d_split_kernel<..., PRO=%s, STATS=true, OCC=2>: Conv3d 3x3x3 64->64 @32^3, B=32, "
                                                "every tile occupied; %s" % ("true" if key == "conv2_form" else "false", what),
                                      "bound": "mfma", "achieved": flops / tt / 1e12, "peak": MFMA_F16_PEAK_TF / 3.0,
                                      "unit": "TFLOP/s (fp32-equivalent conv FLOPs)",
                                      "frac": flops / tt / 1e12 / (MFMA_F16_PEAK_TF / 3.0), "traffic": None,
                                      "us_per_launch": tt * 1e6}
                # the same in-step launches on the operands the model feeds them: conv1 reads a voxelised 2048-point cloud
                # (94 % exact zeros), every tile still computed (all-ones occupancy) -- the MFMA count of the dense layer at the
                # operand statistics of the workload (random dense operands are the power-limited worst case)
                cov = torch.randn(B, 3, 2048, device=dev)
                fv = torch.randn(B, 64, 2048, device=dev)
                gridz, _, _, _ = bk.voxelize_points_forward(fv, cov, 32, True, 0.0)
                gz5 = gridz.view(B, 64, 32, 32, 32)
                t_c1v = ev_time_graph(lambda: fused_ops.conv3d_fused(gz5, conv, None, True,
                                                                     fused_ops.conv3d_occupancy(ones, 32, 64, B)[0]), 10) - t_occ
                roof_step["conv1_form_voxelized_input"] = {
                    "kernel": "conv3d_split_kernel<..., PRO=false, STATS=true, OCC=2>: Conv3d 3x3x3 64->64 @32^3, B=32, every tile "
                              "computed, input = the voxelised features of 2048-point clouds (what conv1 of a PVConv reads)",
                    "bound": "mfma", "achieved": flops / t_c1v / 1e12, "peak": MFMA_F16_PEAK_TF / 3.0,
                    "unit": "TFLOP/s (fp32-equivalent conv FLOPs)", "frac": flops / t_c1v / 1e12 / (MFMA_F16_PEAK_TF / 3.0),
                    "traffic": None, "us_per_launch": t_c1v * 1e6}
                # ... and conv2 on what conv1 leaves behind for it: its output is exactly bias1 away from the points, so the
                # activated input minus the constant (delta mode) is exactly zero there
                y1v = fused_ops.conv3d_fused(gz5, conv, None, True, fused_ops.conv3d_occupancy(ones, 32, 64, B)[0])[0]
                t_c2v = ev_time_graph(lambda: fused_ops.conv3d_fused(y1v, conv, (pa, pb), True,
                                                                     fused_ops.conv3d_occupancy(ones, 32, 64, B)[1],
                                                                     prev_conv=conv), 10) - t_occ
                roof_step["conv2_form_voxelized_input"] = {
                    "kernel": "conv3d_split_kernel<..., PRO=true, STATS=true, OCC=2>: Conv3d 3x3x3 64->64 @32^3, B=32, every tile "
                              "computed, input = conv1's output on the voxelised clouds (AdaGN+Swish prologue, constant + delta: what "
                              "conv2 of a PVConv reads)",
                    "bound": "mfma", "achieved": flops / t_c2v / 1e12, "peak": MFMA_F16_PEAK_TF / 3.0,
                    "unit": "TFLOP/s (fp32-equivalent conv FLOPs)", "frac": flops / t_c2v / 1e12 / (MFMA_F16_PEAK_TF / 3.0),
                    "traffic": None, "us_per_launch": t_c2v * 1e6}
                del cov, fv, gridz, gz5, y1v
                roof = dict(roof_step["conv2_form"])
                roof["note"] = ("the in-step instantiation (conv2 of a PVConv) on dense RANDOM operands -- the worst case for the "
                                "board's power management, which holds this launch at 1400 W and ~1.8 GHz; peak = dense fp16 "
                                "MFMA peak (2500 TF) / 3 MFMA products per fp32-equivalent product; roofline_plain_kernel = the "
                                "same layer without prologue / statistics / queue (+ zero_input), roofline_conv1_form = the other "
                                "in-step instantiation, roofline_conv{1,2}_form_voxelized_input = the same launches on the operands the "
                                "model feeds them; mfma_ceiling = what a bare MFMA stream sustains on this board")
                if mfma_ceiling and "random_fp16_operands" in mfma_ceiling:
                    ceil_tf = mfma_ceiling["random_fp16_operands"]["TFLOP/s fp16"]
                    roof["frac_of_power_limited_mfma_rate"] = {
                        "frac": 3.0 * roof["achieved"] / ceil_tf,
                        "note": "3 x achieved (fp16 MFMA TFLOP/s of the launch) / the bare MFMA stream's sustained rate on random "
                                "fp16 operands measured in this run (mfma_ceiling)"}
                del ones
            del xin
            C, N, r = 64, 2048, 32
            co = torch.randn(B, 3, N, device=dev)
            ft = torch.randn(B, C, N, device=dev)
            tv = ev_time_graph(lambda: bk.voxelize_points_forward(ft, co, r, True, 0.0), 20)
            vbytes = 4.0 * B * (3 * N + C * N + C * r ** 3 + N + r ** 3) + 4.0 * B * 3 * N
            # a forward voxelises 4 distinct (cloud, r) pairs 14 times: the step runs lion_voxel_index once per pair and
            # lion_voxel_scatter per feature tensor (round 4).  The scatter kernel is the one the bytes go through; its
            # algorithmic bytes = features in + dense grid out + the plan it reads (pos, 1/count per point, slot lists)
            plan = bk.voxel_index(co, r, True, 0.0)
            ti = ev_time_graph(lambda: bk.voxel_index(co, r, True, 0.0), 20)
            ts = ev_time_graph(lambda: bk.voxel_scatter(ft, plan), 20)
            sbytes = 4.0 * B * (C * N + C * r ** 3 + 4 * N)
            roofv = hbm_roofline("voxel_scatter (K2 from the index plan) C=64 N=2048 r=32: vox_scatter_kernel", sbytes, ts)
            roofv["index_kernel_us (once per (cloud, r) pair, 4 per forward)"] = ti * 1e6
            # the form a PVConv's fused branch launches since round 5: the grid's only reader is the sparse convolution, z-rows
            # outside every occupied tile's halo are not written (lion_voxel_scatter_read; fewer bytes: not a roofline figure)
            occ_r = fused_ops.conv3d_occupancy(plan["cnt"], r, 64, B, consumer_aware=2)[0]
            tsr = ev_time_graph(lambda: bk.voxel_scatter(ft, plan, occ_r), 20)
            roofv["sparse_reader_form_us (lion_voxel_scatter_read, same cloud: rows nobody reads are not written)"] = tsr * 1e6
            del occ_r
            roofv["fused_single_call"] = hbm_roofline("voxelize_points (P1+K1+K2 in one launch, the C-ABI drop-in entry): "
                                                      "vox_fused_kernel", vbytes, tv)
            nc = plan["norm"]
            gridv = torch.randn(B, C, r ** 3, device=dev)
            td = ev_time_graph(lambda: bk.trilinear_devoxelize_forward(r, False, nc, gridv), 20)
            dbytes = 4.0 * B * (3 * N + C * min(r ** 3, 8 * N) + C * N)       # 8 corners per point
            sc_, sh_ = torch.rand(B, C, device=dev) + 0.5, torch.randn(B, C, device=dev)
            gv5 = gridv.view(B, C, r, r, r)
            tda = ev_time_graph(lambda: fused_ops.devoxelize_affine(gv5, nc, r, sc_, sh_), 20)
            # a forward devoxelises the same (cloud, r = 32) four times: the step runs lion_trilinear_devoxelize_plan once per
            # pair and the planned forward per feature tensor (round 4); the plan adds its own bytes to the count
            dplan = fused_ops.devoxelize_plan(nc, r)
            tdp = ev_time_graph(lambda: fused_ops.devoxelize_affine(gv5, nc, r, sc_, sh_, plan=dplan), 20)
            tdi = ev_time_graph(lambda: fused_ops.devoxelize_plan(nc, r), 20)
            roofd = hbm_roofline("trilinear_devoxelize, planned, with the AdaGN x SE affine folded in (what a PVConv runs) C=64 "
                                 "N=2048 r=32: devox_ring_kernel<true, 8, 1>", dbytes + 8.0 * B * C + 16.0 * B * N, tdp)
            roofd["plan_kernel_us (once per (cloud, r = 32) pair and forward)"] = tdi * 1e6
            roofd["one_step_affine"] = hbm_roofline("trilinear_devoxelize_affine (one launch, per-cloud setup inside) C=64 N=2048 "
                                                    "r=32: devox_ring_kernel<true, 8, 0>", dbytes + 8.0 * B * C, tda)
            roofd["plain_eval"] = hbm_roofline("trilinear_devoxelize C=64 N=2048 r=32 (eval, the reference's entry point)",
                                               dbytes, td)
            del plan, gv5, dplan
            # backward scatters of the training path (K5, K8, K12-grad) at the largest shapes of a forward; algorithmic bytes =
            # gradient in + indices / weights + dense gradient out, each once (tools/kbench.py --only bwd uses the same)
            _, inds, wgts = bk.trilinear_devoxelize_forward(r, True, nc, gridv)
            gyp = torch.randn(B, C, N, device=dev)
            tk5 = ev_time_graph(lambda: bk.trilinear_devoxelize_backward(gyp, inds, wgts, r), 10)
            roofb = {"K5": hbm_roofline("trilinear_devoxelize_backward C=64 N=2048 r=32: devox_bwd_lds_kernel",
                                        4.0 * B * (C * N + 16 * N + C * r ** 3), tk5)}
            del gridv, inds, wgts, gyp
            Cg, Ng, Mg = 35, 2048, 1024
            gidx = torch.randint(0, Ng, (B, Mg, 32), device=dev, dtype=torch.int32)
            gyg = torch.randn(B, Cg, Mg, 32, device=dev)
            tk8 = ev_time_graph(lambda: bk.grouping_backward(gyg, gidx, Ng), 10)
            roofb["K8"] = hbm_roofline("grouping_backward C=35 N=2048 M=1024 U=32 (SA-0)", 4.0 * B * (Cg * Mg * 32 + Mg * 32 + Cg * Ng), tk8)
            del gidx, gyg
            Ci, Ni, Mi = 192, 2048, 1024
            pts = torch.randn(B, 3, Ni, device=dev)
            _, ii, iw = bk.three_nearest_neighbors_interpolate_forward(pts, pts[:, :, :Mi].contiguous(), torch.randn(B, Ci, Mi, device=dev))
            gyi = torch.randn(B, Ci, Ni, device=dev)
            tk12 = ev_time_graph(lambda: bk.three_nearest_neighbors_interpolate_backward(gyi, ii, iw, Mi), 10)
            roofb["K12g"] = hbm_roofline("three_nn_interpolate_backward C=192 N=2048 M=1024 (FP-0)", 4.0 * B * (Ci * Ni + 6 * Ni + Ci * Mi), tk12)
            del pts, ii, iw, gyi
            # ---- VALU / latency-bound operators: Chamfer, EMD against the fp32 VALU peak resp. the
            # transcendental issue rate; ball query, FPS, 3-NN as times (latency-bound: no byte or flop roofline applies)
            from lion_amd.chamfer3d import chamfer_3DDist_nograd
            from lion_amd.emd import earth_mover_distance_nograd
            Ne = 2048
            ea, eb = torch.rand(B, Ne, 3, device=dev), torch.rand(B, Ne, 3, device=dev)
            cd = chamfer_3DDist_nograd()
            tcd = ev_time_graph(lambda: cd(ea, eb), 10)
            cd_flops = 2.0 * Ne * Ne * 8 * B
            roof_cd = {"kernel": "chamfer_fwd_kernel: 32 pairs of 2048-point clouds, both directions (csrc/chamfer.hip)",
                       "bound": "valu", "achieved": cd_flops / tcd / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                       "frac": cd_flops / tcd / 1e12 / MFMA_F32_PEAK_TF, "us_per_call": tcd * 1e6,
                       "valu_instructions_per_pair": 11.44,
                       "frac_of_valu_issue_rate": 11.44 * (2.0 * Ne * Ne * B) / tcd / (MFMA_F32_PEAK_TF * 1e12 / 2.0),
                       "note": "2 N M distance evaluations x 8 flop per pair (SURVEY.md 8d) against the fp32 vector peak (157.3 TF, which "
                               "counts an FMA per lane and clock).  The bit-exact expression is 3 sub + 3 mul + 2 add + compare + 2 selects "
                               "= 11.44 VALU instructions per pair in the ISA (no FMA: the oracle's roundings), so the ceiling of THIS "
                               "instruction stream is 78.6 T lane-instructions/s / 11.44; frac_of_valu_issue_rate relates the launch to it"}
            temd = ev_time_graph(lambda: earth_mover_distance_nograd(ea, eb, transpose=False), 5)
            emd_evals = 30.0 * Ne * Ne * B
            exp_peak = MFMA_F32_PEAK_TF * 1e12 / 2.0 / 4.0        # lane-instructions/s (157.3 TF / 2 flop per FMA), quarter rate
            roof_emd = {"kernel": "emd level kernels: approxmatch cost of 32 pairs of 2048-point clouds, no match matrix "
                                  "(csrc/emd.hip, lion_emd_cost)", "bound": "valu (v_exp_f32 issue rate)",
                        "achieved": emd_evals / temd / 1e12, "peak": exp_peak / 1e12, "unit": "T exp-distance evaluations/s",
                        "frac": emd_evals / temd / exp_peak, "us_per_call": temd * 1e6,
                        "note": "30 N M exp-distance evaluations per pair (SURVEY.md 8d); peak = fp32 vector lane rate "
                                "(157.3e12 / 2) / 4: transcendentals issue at quarter rate.  Round 6: exp2 on pre-scaled coordinates, "
                                "pass 3 fused with the next level's pass 1 (21 launches instead of 30, 8.4 full-rate VALU instructions "
                                "per evaluation where there were 13.4): csrc/emd.hip emd_fast_*"}
            p2, p1 = torch.randn(B, 3, 2048, device=dev), None
            tfps = ev_time_graph(lambda: bk.furthest_point_sampling(p2, 1024), 3)
            cen = p2[:, :, :1024].contiguous()
            tbq = ev_time_graph(lambda: bk.ball_query(cen, p2, 0.1, 32), 10)
            cf = torch.randn(B, 192, 1024, device=dev)
            tnn = ev_time_graph(lambda: bk.three_nearest_neighbors_interpolate_forward(p2, cen, cf), 10)
            latency_ops = {"K9 furthest_point_sampling 2048->1024": {"us": tfps * 1e6, "us_per_round": tfps * 1e6 / 1024},
                           "K6 ball_query M=1024 N=2048 r=0.1 U=32": {"us": tbq * 1e6},
                           "K11+K12 three_nn_interpolate C=192 M=1024 N=2048": {"us": tnn * 1e6},
                           "note": "latency / VALU bound (SURVEY.md 8d): reported as times at B=32, graph replay"}
            del ea, eb, p2, cen, cf
            roofv_total, roofd_total = vox_devox_totals(bk, fused_ops, B, dev)
            # HBM bytes per launch of the dominant kernels: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes cannot run inside
            # this process; tools/prof_traffic.sh writes them to profiles/ and the line quotes the file it read
            def traffic_of(name):
                import glob
                for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_traffic.json" % name)), reverse=True):
                    try:
                        t_ = json.load(open(f))
                        e_ = max(t_["kernels"].values(), key=lambda e: e.get("launches", 0))   # the kernel the pass targeted
                        return {"bytes_per_launch": e_.get("hbm_bytes_per_launch"), "avg_us_in_that_pass": e_.get("avg_us"),
                                "source": os.path.relpath(f, ROOT), "profile_commit": t_.get("profile_commit")}
                    except Exception:
                        continue
                return None
            for rf, nm in ((roof, "conv_instep"), (roofv, "vox_scatter_64_2048_32"), (roofd, "devox_affine_64_2048_32")):
                tr = traffic_of(nm)
                if tr is not None:
                    rf["traffic"] = tr["bytes_per_launch"]
                    rf["traffic_source"] = tr
        out = {
            "metric": "shapes/sec @1000-step DDIM, Bx2048pts", "value": value, "unit": "shapes/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            # the real 1000-step chain: whole job / rank 0's wall time of one ddim_step = 1000 call (ranks are independent and
            # equally loaded); with --steps 1000 it is `value` itself
            "value_full_chain_1000": (value if K == 1000 else (None if full_chain is None else
                                                               total_shapes / full_chain["seconds"])),
            "ms_per_step_full_chain": (ms_per_step if K == 1000 else (full_chain or {}).get("ms_per_step")),
            # the same call on the clouds a TRAINED model's chain visits (forced forward-process states), see ForcedClouds
            "ms_per_step_forced_clouds": (forced or {}).get("ms_per_step"),
            "value_forced_clouds": (None if forced is None else total_shapes / B * forced["shapes_per_s"]),
            "voxelize_frac": roofv["frac"], "devoxelize_frac": roofd["frac"],
            "ms_per_step_B4": (small.get(4) or {}).get("ms_per_step"),
            "dtype": ("f32 (operands of the 3x3x3 voxel convolutions" +
                      (" and of the long 1x1 convolutions" if fused_ops.PW_SPLIT else "") +
                      " cut into fp16 hi/lo pairs on the 16-bit MFMA pipe with f32 accumulation -- fp32-accurate, "
                      "tests/test_conv_split_gpu.py, tests/test_pwconv_split_gpu.py; everything else f32)")
                     if conv_ops.SPLIT else "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: unconditional airplane prior sampling, 1000-step DDIM chain "
                                   "(global PriorSEDrop + local PVCNN2Prior) + VAE decode, through the product "
                                   "sampler lion_amd.sampling.generate_samples_vada_2prior",
                       "shapes_per_gpu": B, "shapes_total": total_shapes, "points": 2048, "chain_steps": 1000,
                       "timed_steps_per_prior": K, "extrapolated": K != 1000, "decode_seconds": decode_s,
                       "timed_region_seconds": elapsed,
                       "timed_region_seconds_all_runs": runs,
                       "ms_per_step_all_runs": [max(r_ - decode_s, 1e-9) / K * 1e3 for r_ in runs],
                       "streams": {"geometry_split_graph": geometry.SPLIT_GRAPH, "geometry_prefetch_branch": geometry.ENABLED,
                                   "point_branch_stream": pvcnn2_ada.OVERLAP_POINT_BRANCH,
                                   "captured_chains [global prior, local prior]": streams,
                                   "note": "split graph: the FPS / ball-query chain of a step replays as its own graphs on a "
                                           "second stream beside the first PVConv (lion_amd/chain.py)"},
                       "voxel_index_plans": pvcnn2_ada.VOX_PLAN,
                       # the metric's REAL chain, as scalars the driver's record keeps: one call of the product
                       # sampler with ddim_step = 1000 on rank 0 (with --steps 1000 it IS the timed region)
                       "full_chain_1000_shapes_per_s": (value if K == 1000 else (full_chain or {}).get("shapes_per_s")),
                       "full_chain_1000_ms_per_step": (ms_per_step if K == 1000 else (full_chain or {}).get("ms_per_step")),
                       "full_chain_1000_seconds": (elapsed if K == 1000 else (full_chain or {}).get("seconds")),
                       "forced_clouds_steps": (forced or {}).get("steps"),
                       "forced_clouds_ms_per_step": (forced or {}).get("ms_per_step"),
                       "forced_clouds_shapes_per_s": (forced or {}).get("shapes_per_s"),
                       "forced_clouds_conv1_empty_tile_frac": ((forced or {}).get("tiles") or {}).get("conv1_empty_mean"),
                       "forced_clouds_conv2_empty_tile_frac": ((forced or {}).get("tiles") or {}).get("conv2_empty_mean"),
                       "chain_conv1_empty_tile_frac": (chain_tiles or {}).get("conv1_empty_mean"),
                       "chain_conv2_empty_tile_frac": (chain_tiles or {}).get("conv2_empty_mean"),
                       "forced_clouds": forced, "chain_tiles": chain_tiles,
                       "ms_per_step_B4": (small.get(4) or {}).get("ms_per_step"),
                       "ms_per_step_B8": (small.get(8) or {}).get("ms_per_step"),
                       "ms_per_step_B16": (small.get(16) or {}).get("ms_per_step"),
                       # a 32-shape job over 8 GPUs = 4 shapes per rank: T(32 on 1) / (8 x T(4 on 1)), T = 1000 steps + decode
                       "predicted_strong_scaling_efficiency_8gpu_32shapes": (
                           None if 4 not in small or B != 32 else
                           (1000.0 * ms_per_step / 1e3 + decode_s) /
                           (8.0 * (1000.0 * small[4]["ms_per_step"] / 1e3 + small[4]["decode_seconds"]))),
                       "small_batches": small, "bench_sections_seconds": sections,
                       "voxelize_frac_of_hbm": roofv["frac"], "devoxelize_frac_of_hbm": roofd["frac"],
                       "voxelize_forward_total_frac_of_hbm": roofv_total["frac"],
                       "devoxelize_forward_total_frac_of_hbm": roofd_total["frac"],
                       "vendor_library_fallbacks_in_step": sum(_fallback.counts().values()),
                       "launches_per_step": (census or {}).get("launches"),
                       "aten_kernels_in_step": (census or {}).get("aten"),
                       "launch_census_source": (census or {}).get("source"),
                       "kernel_census": census,
                       "full_chain_1000": full_chain,
                       "parallelism": f"{world} independent rank(s), no data-path collective",
                       "launch": "hipGraph replay of [step prologue, denoiser forward, update + Philox noise] (the local "
                                 "prior's step as single-branch graphs on two streams, see streams)" if graph else "eager",
                       "sparse_voxel_convs": not args.no_sparse,
                       "voxel_conv_kernel": "fp16x2 split operands (LION_CONV_SPLIT=0 selects exact-fp32 MFMA)"
                                            if conv_ops.SPLIT else "exact-fp32 MFMA",
                       "pointwise_conv_kernel": "fp16x2 split operands for B*L >= 8192 columns (LION_PW_SPLIT=0: fp32 MFMA)"
                                                if fused_ops.PW_SPLIT else "fp32 MFMA",
                       "ms_per_step_dense_convs": ms_dense,
                       "note": "step = one DDIM step of BOTH priors; with random-init weights the latents drift and "
                               "the exact empty-tile skip saves a trajectory-dependent share of the conv work: "
                               "ms_per_step_dense_convs is the same call with every tile computed (short chain)"},
            "roofline": roof, "roofline_plain_kernel": roof_plain, "roofline_conv1_form": roof_step.get("conv1_form"),
            "roofline_conv1_form_voxelized_input": roof_step.get("conv1_form_voxelized_input"),
            "roofline_conv2_form_voxelized_input": roof_step.get("conv2_form_voxelized_input"), "mfma_ceiling": mfma_ceiling,
            "whole_step_mfma_frac": (None if ms_dense is None or B != 32 else
                                     {"frac": 1909.0 / ms_dense / (MFMA_F16_PEAK_TF / 3.0),
                                      "note": "1909 GFLOP of a B=32 step (SURVEY.md 8d) / ms_per_step_dense_convs / (2500/3 TF)"}),
            "roofline_fp32_kernel": roof32, "roofline_voxelize": roofv, "roofline_devoxelize": roofd,
            "roofline_voxelize_forward_total": roofv_total, "roofline_devoxelize_forward_total": roofd_total,
            "roofline_backward_operators": roofb, "roofline_chamfer": roof_cd, "roofline_emd": roof_emd,
            "latency_bound_operators": latency_ops,
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
        if world == 1 and args.side_lines:
            # configs[2..4] beside the metric's line (opt-in): short side runs of --mode train_* (own process each); the full
            # lines go to the detail file under a key the driver's parser never sees, three scalars each to config
            torch.cuda.empty_cache()
            side = run_side_lines(["train_vae", "train_prior", "train_prior_clip"], args.side_steps, 240)
            out["training_side_runs"] = side
            for mode_, line_ in side.items():
                out["config"][f"{mode_}_samples_per_s"] = line_.get("value")
                out["config"][f"{mode_}_ms_per_step"] = line_.get("ms_per_step")
                out["config"][f"{mode_}_wgrad_roofline_frac"] = (line_.get("roofline") or {}).get("frac")
        emit(out, args.detail_file)
    if world > 1:
        dist.barrier(device_ids=[torch.cuda.current_device()]) if backend == "nccl" else dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
