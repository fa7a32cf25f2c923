"""What does hipGraphDebugDotPrint (torch.cuda.CUDAGraph.debug_dump) write on this ROCm?  Captures a small graph holding one
kernel of this library and two ATen kernels, dumps it, prints the file, then the census of a B = 2 local-prior chain step
(lion_amd.chain.GraphedChain.kernel_census): the source of bench.py's aten_kernels_in_step / launches_per_step."""
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd import chain  # noqa: E402

chain.DEBUG_GRAPHS = True
dev = torch.device("cuda")
x = torch.randn(4, 64, 256, device=dev)
y = torch.empty_like(x)
from lion_amd.diffusion_ops import ddim_update  # noqa: E402
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    z = torch.cat([x, x], 1) * 2.0
    ddim_update(x.view(-1), y.view(-1), None, 0.9, 0.1, 0.0)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = chain._new_graph()
with torch.cuda.graph(g):
    z = torch.cat([x, x], 1) * 2.0
    w = ddim_update(x.view(-1), y.view(-1), None, 0.9, 0.1, 0.0)
path = os.path.join(tempfile.gettempdir(), "probe.dot")
g.debug_dump(path)
txt = open(path, errors="replace").read()
print("---- dot file (%d bytes) ----" % len(txt))
print(txt[:3000])
print("---- names ----")
g2 = chain._new_graph()
with torch.cuda.graph(g2):
    z = torch.cat([x, x], 1) * 2.0
    w = ddim_update(x.view(-1), y.view(-1), None, 0.9, 0.1, 0.0)
print(chain.graph_kernel_names(g2))

if "--chain" in sys.argv:
    import bench
    from lion_amd.config import released_prior_cfg
    from lion_amd.sampling import generate_samples_vada_2prior
    cfg = released_prior_cfg("airplane")
    lion = bench.build_models(cfg, dev)
    with torch.no_grad():
        generate_samples_vada_2prior(lion.vae.latent_shape(), lion.priors, lion.diffusion, lion.vae, 2, ddim_step=2)
    for ch in lion.diffusion._chains._entries.values():
        c = ch.kernel_census()
        print(type(ch.model).__name__, {k: v for k, v in c.items() if k != "aten_names"})
        for k, v in sorted(c["aten_names"].items(), key=lambda kv: -kv[1]):
            print("   %3d  %s" % (v, k))
