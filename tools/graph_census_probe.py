"""Does hipGraphDebugDotPrint (torch.cuda.CUDAGraph.debug_dump) list the kernel nodes of a captured graph on this ROCm?
Round 6 wanted bench.py to count the launches / ATen kernels of a replayed step in-process from it.  Result on ROCm 7.2 /
torch 2.10: debug_dump() warns "DEBUG: calling debug_dump()" and writes NO file -- so the census comes from a rocprofv3 kernel
trace instead (tools/census_run.sh + tools/step_census.py -> profiles/r*_step_census_B32.json, read by bench.py)."""
import os
import sys
import tempfile

import torch

dev = torch.device("cuda")
x = torch.randn(4, 64, 256, device=dev)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    z = torch.cat([x, x], 1) * 2.0
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
g.enable_debug_mode()
with torch.cuda.graph(g):
    z = torch.cat([x, x], 1) * 2.0
path = os.path.join(tempfile.gettempdir(), "probe.dot")
g.debug_dump(path)
print("dot file written:", os.path.exists(path), os.path.getsize(path) if os.path.exists(path) else 0)
