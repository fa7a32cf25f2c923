"""-m gpu: kernels that overlap inside the captured sampling step must not disturb each other.  Round 2 found that the
register FPS kernel returned wrong samples whenever the LDS-DMA convolution workgroups shared its CUs inside a hipGraph
replay (tools/fps_under_dma.py) -- eagerly, or beside kernels without LDS-DMA, never.  These tests replay such graphs and
demand bit-identical results."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("C,R", [(32, 32), (64, 32), (128, 16), (128, 8)])
def test_fps_beside_the_dma_convolution_in_a_graph(C, R):
    import fps_under_dma
    res = fps_under_dma.run(C, R, replays=40)
    assert all(r == (0, 0) for r in res), res


def test_local_prior_graph_replay_equals_eager():
    """the whole local denoiser (geometry prefetch + point branch on side streams) captured once and replayed 40 times:
    every replay == the eager forward, bit for bit"""
    from lion_amd.config import released_prior_cfg
    from lion_amd.models.lion import LION
    torch.manual_seed(3)
    lion = LION(released_prior_cfg())
    lion.priors.eval()
    lion.vae.eval()
    B = 2
    sh = lion.vae.latent_shape()
    prior = lion.priors[1]
    with torch.no_grad():
        style = lion.vae.global2style(torch.randn([B] + sh[0], device="cuda"))
        x = torch.randn([B] + sh[1], device="cuda")
        tt = torch.full((B,), 500.0, device="cuda")

        def f():
            return prior(x=x, t=tt, condition_input=style, clip_feat=None).float()
        ref = f().clone()
        f()
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            f()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = f()
        for i in range(40):
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, ref), (i, ((out - ref).abs().max() / ref.abs().max()).item())
