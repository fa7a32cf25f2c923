"""Generic victim harness (round 3): every kernel that overlaps the split-operand convolution inside the captured sampling
step, run on a side stream BESIDE conv3d_split_kernel (6 x Conv3d 64->64 @32^3 at B = 32 on the main stream: all 256
CUs saturated with fp16-MFMA workgroups) inside one hipGraph, replayed R times; per victim the number of replays whose
output differs from the victim run alone, and the number of wrong words.  The victims repeat themselves inside the graph
until they span the convolutions' ~4 ms.
  usage: victims_beside_conv.py [--replays 100] [--B 32] [victim ...]     victims: fps bq nn group grouppts pw gnfold vox devox
  (LION_FPS_SHARE_CU=1 only matters for builds of csrc/sampling.hip from before round 3's fix, selected with LION_HIP_SO:
  those requested a whole CU's LDS unless it was set; the current kernel always shares CUs.)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd.functional.backend import _backend as bk  # noqa: E402
from lion_amd import fused_ops as fo  # noqa: E402
from lion_amd.conv_ops import conv3d_k3  # noqa: E402


def make_victims(B, dev):
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    pts = rn(B, 3, 2048)
    feat = rn(B, 64, 2048)
    ctr_idx = bk.furthest_point_sampling(pts, 1024)
    ctr = bk.gather_features_forward(pts, ctr_idx)
    nb = bk.ball_query(ctr, pts, 0.1 * 3, 32)
    cfeat = rn(B, 128, 1024)
    pw = torch.nn.Conv1d(192, 128, 1).to(dev)
    xp = rn(B, 192, 2048)
    gn = torch.nn.GroupNorm(8, 128).to(dev)
    fac, gb = rn(B, 128).abs() + 0.5, rn(B, 128)
    co32 = torch.rand(B, 3, 2048, device=dev, generator=g) * 31.0
    grid = rn(B, 64, 32 ** 3)
    _, stats0 = fo.pwconv_fused(xp, pw, None, want_stats=True, split=True)

    def v_fps():
        return [bk.furthest_point_sampling(pts, 1024)]

    def v_bq():
        return [bk.ball_query(ctr, pts, 0.3, 32)]

    def v_nn():
        return list(bk.three_nearest_neighbors_interpolate_forward(pts, ctr, cfeat))

    def v_group():
        return [bk.grouping_forward(feat, nb)]

    def v_grouppts():
        return [fo.group_points(pts, ctr, feat, nb)]

    def v_pw():
        return list(fo.pwconv_fused(xp, pw, None, want_stats=True, split=True))

    def v_gnfold():
        return list(fo.groupnorm_fold(stats0, gn, fac, gb, 2048))

    def v_vox():
        return [t for t in bk.voxelize_points_forward(feat, pts, 32, True, 0.0) if t is not None]

    def v_devox():
        return [bk.trilinear_devoxelize_forward(32, False, co32, grid)[0]]

    # (function, repeats inside one graph): repeats sized to ~4 ms of side-stream work
    return {"fps": (v_fps, 6), "bq": (v_bq, 40), "nn": (v_nn, 20), "group": (v_group, 20), "grouppts": (v_grouppts, 20),
            "pw": (v_pw, 40), "gnfold": (v_gnfold, 100), "vox": (v_vox, 30), "devox": (v_devox, 40)}


def run(names, B=32, replays=100, dev="cuda"):
    torch.manual_seed(0)
    conv = torch.nn.Conv3d(64, 64, 3, padding=1).to(dev)
    x = torch.randn(B, 64, 32, 32, 32, device=dev)
    side = torch.cuda.Stream(priority=-1)
    victims = make_victims(B, dev)
    report = {}
    with torch.no_grad():
        yref = x
        for _ in range(6):
            yref = conv3d_k3(yref, conv.weight, conv.bias, split=True)
        yref = yref.clone()
        for name in names:
            fn, reps = victims[name]
            ref = [t.clone() for t in fn()]

            def fwd():
                main = torch.cuda.current_stream()
                side.wait_stream(main)
                outs = []
                with torch.cuda.stream(side):
                    for _ in range(reps):
                        outs.append(fn())
                y = x
                for _ in range(6):
                    y = conv3d_k3(y, conv.weight, conv.bias, split=True)
                main.wait_stream(side)
                return outs, y
            fwd()
            torch.cuda.synchronize()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fwd()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                outs, y = fwd()
            bad_replays = bad_words = bad_conv = 0
            for _ in range(replays):
                gr.replay()
                torch.cuda.synchronize()
                w = 0
                for o in outs:
                    for a, b in zip(o, ref):
                        w += int((a != b).sum().item()) if a.dtype != torch.float32 else \
                            int((a.view(torch.int32) != b.view(torch.int32)).sum().item())
                bad_replays += w > 0
                bad_words += w
                bad_conv += int((y != yref).sum().item()) > 0
            report[name] = {"replays": replays, "victim_launches_per_replay": reps, "replays_with_wrong_victim_output": bad_replays,
                            "wrong_words": bad_words, "replays_with_wrong_conv_output": bad_conv}
            print(name, report[name], flush=True)
            del gr
    return report


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("victims", nargs="*", default=["fps", "bq", "nn", "group", "grouppts", "pw", "gnfold", "vox", "devox"])
    ap.add_argument("--replays", type=int, default=100)
    ap.add_argument("--B", type=int, default=32)
    a = ap.parse_args()
    print("LION_FPS_SHARE_CU =", os.environ.get("LION_FPS_SHARE_CU"), " LION_HIP_SO =", os.environ.get("LION_HIP_SO"))
    run(a.victims, a.B, a.replays)
