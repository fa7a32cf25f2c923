"""Furthest-point sampling on a side stream beside one of the kernels it can overlap, captured in one hipGraph and
replayed: per replay, how many of the B x M sample indices differ from the stand-alone result, and how many conv outputs
differ.  History: the round-2 build of fps_reg_kernel returned 300-1800 wrong indices of 2048 in nearly every replay
beside conv3d_split_kernel (never eagerly, never beside the other kernels); round 3 located the cause in packed fp32
VALU emitted by the SLP vectoriser and the library is built without it -- every replay matches.
tools/victims_beside_conv.py is the B = 32, all-kernels successor of this script.
usage: fps_under_dma.py [C] [R] [conv|fp32|pw|devox|vox]   (conv C->C at R^3, default 32 32 conv)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd.functional import backend as _bk


def run(C=32, R=32, B=2, replays=20, split=True, aggressor="conv"):
    """aggressor: 'conv' (3x3x3 split conv, LDS-DMA weights; split=False: the fp32 kernel), 'pw' (split 1x1 conv, LDS-DMA
    weights, no scratch), 'devox' (row-gather devoxelize, LDS-DMA of the grid rows), 'vox' (voxelize, no LDS-DMA)"""
    torch.manual_seed(0)
    side = torch.cuda.Stream(priority=-1)
    pts = torch.randn(B, 3, 2048, device="cuda")
    conv = torch.nn.Conv3d(C, C, 3, padding=1).cuda()
    x = torch.randn(B, C, R, R, R, device="cuda")
    if aggressor != "conv":
        from lion_amd import fused_ops as fo
        pw = torch.nn.Conv1d(192, 128, 1).cuda()
        xp = torch.randn(32, 192, 2048, device="cuda")
        feat = torch.randn(32, 64, 2048, device="cuda")
        co = torch.rand(32, 3, 2048, device="cuda") * 31.0
        grid = torch.randn(32, 64, 32 ** 3, device="cuda")
        coi = co.floor().int().contiguous()

        def conv3d_k3(y, w, b, split=True):  # noqa: F811 -- stand-in: six launches of the chosen aggressor
            if aggressor == "pw":
                fo.pwconv_fused(xp, pw, None, want_stats=False, split=True)
            elif aggressor == "devox":
                _bk._backend.trilinear_devoxelize_forward(32, False, co, grid)
            elif aggressor == "vox":
                _bk._backend.avg_voxelize_forward(feat, coi, 32)
            return y
    else:
        from lion_amd.conv_ops import conv3d_k3
    with torch.no_grad():
        ref = _bk._backend.furthest_point_sampling(pts, 1024).clone()
        yref = x
        for _ in range(6):
            yref = conv3d_k3(yref, conv.weight, conv.bias, split=split)
        yref = yref.clone()

        def fwd():
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                idx = _bk._backend.furthest_point_sampling(pts, 1024)
            y = x
            for _ in range(6):
                y = conv3d_k3(y, conv.weight, conv.bias, split=split)
            main.wait_stream(side)
            return idx, y
        fwd()
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fwd()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            idx, y = fwd()
        res = []
        for _ in range(replays):
            g.replay()
            torch.cuda.synchronize()
            res.append((int((idx != ref).sum().item()), int((y != yref).sum().item())))
    return res


if __name__ == "__main__":
    C = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32
    R = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 32
    ag = [a for a in sys.argv[1:] if a in ("conv", "fp32", "pw", "devox", "vox")]
    ag = ag[0] if ag else "conv"
    res = run(C, R, split=ag != "fp32", aggressor="conv" if ag == "fp32" else ag)
    print(f"aggressor {ag} (conv {C}->{C} @ {R}^3): (wrong FPS indices, wrong conv outputs) per replay:", res)
