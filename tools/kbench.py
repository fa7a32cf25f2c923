"""Per-kernel micro-benchmark of the hot-path operators on the (C,N,r) tuples of one PVCNN2Prior
forward, B=32.  Prints time and ALGORITHMIC GB/s (SURVEY.md 8d byte formulas)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd.functional.backend import _backend as bk  # noqa: E402

VOX = [(4, 2048, 32), (32, 2048, 32), (128, 1024, 16), (192, 256, 8), (128, 64, 8), (128, 256, 8),
       (128, 1024, 16), (64, 2048, 32)]
DEVOX = [(32, 2048, 32), (64, 1024, 16), (128, 256, 8), (128, 64, 8), (128, 1024, 16), (64, 2048, 32)]
FPS = [(2048, 1024), (1024, 256), (256, 64), (64, 16)]
BQ = [(1024, 2048, 0.1), (256, 1024, 0.2), (64, 256, 0.4), (16, 64, 0.8)]
GRP = [(35, 2048, 1024), (67, 1024, 256), (131, 256, 64), (195, 64, 16)]
NN = [(192, 64, 16), (192, 256, 64), (192, 1024, 256), (192, 2048, 1024)]


def timeit(fn, iters=20, warm=3):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    B = args.B
    g = torch.Generator(device="cuda").manual_seed(0)
    rows = []

    def rec(name, shape, t, nbytes):
        rows.append({"op": name, "shape": shape, "us": round(t * 1e6, 2),
                     "GB/s": round(nbytes / t / 1e9, 1) if nbytes else None})
        print(f"{name:28s} {str(shape):22s} {t*1e6:9.1f} us  {nbytes/t/1e9 if nbytes else 0:8.1f} GB/s", flush=True)

    want = lambda k: (not args.only) or (k in args.only.split(","))
    if want("vox"):
        for C, N, r in VOX:
            co = torch.randn(B, 3, N, device="cuda", generator=g)
            feat = torch.randn(B, C, N, device="cuda", generator=g)
            nbytes = 4 * B * (3 * N + C * N + C * r ** 3 + N + r ** 3)
            rec("voxelize_points(P1+K1+K2)", (C, N, r), timeit(lambda: bk.voxelize_points_forward(feat, co, r, True, 0.0), args.iters), nbytes + 4 * B * 3 * N)
            _, nc, _, _ = bk.voxelize_points_forward(feat, co, r, True, 0.0)
            vc = torch.round(nc).int()
            rec("avg_voxelize(K1+K2)", (C, N, r), timeit(lambda: bk.avg_voxelize_forward(feat, vc, r), args.iters), nbytes)
    if want("devox"):
        for C, N, r in DEVOX:
            co = torch.randn(B, 3, N, device="cuda", generator=g)
            _, nc, _, _ = bk.voxelize_points_forward(None, co, r, True, 0.0)
            grid = torch.randn(B, C, r ** 3, device="cuda", generator=g)
            nbytes = 4 * B * (3 * N + C * min(r ** 3, 8 * N) + C * N)
            rec("trilinear_devoxelize(K4)", (C, N, r), timeit(lambda: bk.trilinear_devoxelize_forward(r, False, nc, grid), args.iters), nbytes)
    if want("fps"):
        for N, M in FPS:
            co = torch.randn(B, 3, N, device="cuda", generator=g)
            rec("furthest_point_sampling(K9)", (N, M), timeit(lambda: bk.furthest_point_sampling(co, M), args.iters), 0)
    if want("bq"):
        for M, N, rad in BQ:
            co = torch.randn(B, 3, N, device="cuda", generator=g) * 0.3
            ctr = co[:, :, :M].contiguous()
            rec("ball_query(K6)", (M, N, rad), timeit(lambda: bk.ball_query(ctr, co, rad, 32), args.iters), 0)
    if want("grp"):
        for C, N, M in GRP:
            feat = torch.randn(B, C, N, device="cuda", generator=g)
            idx = torch.randint(0, N, (B, M, 32), device="cuda", dtype=torch.int32)
            nbytes = 4 * B * (C * N + M * 32 + C * M * 32)
            rec("grouping(K7)", (C, N, M), timeit(lambda: bk.grouping_forward(feat, idx), args.iters), nbytes)
    if want("nn"):
        for C, N, M in NN:
            pts = torch.randn(B, 3, N, device="cuda", generator=g)
            ctr = pts[:, :, :M].contiguous()
            cf = torch.randn(B, C, M, device="cuda", generator=g)
            nbytes = 4 * B * (C * M + 6 * N + C * N)
            rec("three_nn_interpolate(K11+12)", (C, N, M), timeit(lambda: bk.three_nearest_neighbors_interpolate_forward(pts, ctr, cf), args.iters), nbytes)
    if want("bwd"):
        # the backward operators of the training path (K3, K5, K8, K12-grad) at the largest shapes of a forward;
        # algorithmic bytes: gradient in + indices / weights + dense gradient out, each once
        C, N, r = 64, 2048, 32
        co = torch.randn(B, 3, N, device="cuda", generator=g)
        feat = torch.randn(B, C, N, device="cuda", generator=g)
        _, nc, ind, cnt = bk.voxelize_points_forward(feat, co, r, True, 0.0)
        gy = torch.randn(B, C, r ** 3, device="cuda", generator=g)
        rec("avg_voxelize_backward(K3)", (C, N, r), timeit(lambda: bk.avg_voxelize_backward(gy, ind, cnt), args.iters),
            4 * B * (C * min(r ** 3, N) + N + r ** 3 + C * N))
        grid = torch.randn(B, C, r ** 3, device="cuda", generator=g)
        _, inds, wgts = bk.trilinear_devoxelize_forward(r, True, nc, grid)
        gyp = torch.randn(B, C, N, device="cuda", generator=g)
        rec("trilinear_devoxelize_backward(K5)", (C, N, r), timeit(lambda: bk.trilinear_devoxelize_backward(gyp, inds, wgts, r), args.iters),
            4 * B * (C * N + 16 * N + C * r ** 3))
        Cg, Ng, Mg = 35, 2048, 1024
        idx = torch.randint(0, Ng, (B, Mg, 32), device="cuda", dtype=torch.int32)
        gyg = torch.randn(B, Cg, Mg, 32, device="cuda", generator=g)
        rec("grouping_backward(K8)", (Cg, Ng, Mg), timeit(lambda: bk.grouping_backward(gyg, idx, Ng), args.iters),
            4 * B * (Cg * Mg * 32 + Mg * 32 + Cg * Ng))
        Ci, Ni, Mi = 192, 2048, 1024
        pts = torch.randn(B, 3, Ni, device="cuda", generator=g)
        ctr = pts[:, :, :Mi].contiguous()
        cf = torch.randn(B, Ci, Mi, device="cuda", generator=g)
        _, ii, iw = bk.three_nearest_neighbors_interpolate_forward(pts, ctr, cf)
        gyi = torch.randn(B, Ci, Ni, device="cuda", generator=g)
        rec("three_nn_interpolate_backward(K12g)", (Ci, Ni, Mi), timeit(lambda: bk.three_nearest_neighbors_interpolate_backward(gyi, ii, iw, Mi), args.iters),
            4 * B * (Ci * Ni + 6 * Ni + Ci * Mi))
    if want("cd"):
        from lion_amd.chamfer3d import chamfer_3DDist_nograd
        from lion_amd.emd import earth_mover_distance_nograd
        x1 = torch.rand(B, 2048, 3, device="cuda", generator=g)
        x2 = torch.rand(B, 2048, 3, device="cuda", generator=g)
        cd = chamfer_3DDist_nograd()
        rec("chamfer(E1)", (B, 2048, 2048), timeit(lambda: cd(x1, x2), args.iters), 0)
        rec("emd approx+cost(E2)", (B, 2048, 2048), timeit(lambda: earth_mover_distance_nograd(x1, x2, transpose=False), 3, 1), 0)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rows, open("gpurun_out/kbench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
