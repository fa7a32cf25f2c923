"""Single-rank RCCL probe (one-GPU box): does the nccl(=RCCL) backend load and create a communicator on this image, does an
all_reduce run, and can all_reduce launches be captured into a hipGraph and replayed (what lion_amd.training.GraphedTrainStep
relies on at world > 1)?  World size 1 exercises library load, communicator init and the capture path of the collective; the
xGMI transport itself needs a multi-GPU node.  Prints one JSON object."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
rep = {}
t0 = time.time()
try:
    dist.init_process_group("nccl", rank=0, world_size=1)
    torch.cuda.set_device(0)
    x = torch.ones(1 << 20, device="cuda")
    dist.all_reduce(x)
    torch.cuda.synchronize()
    rep["eager_all_reduce_ok"] = bool((x == 1).all())
    rep["init_seconds"] = time.time() - t0
    rep["nccl_version"] = str(torch.cuda.nccl.version())
    side = torch.cuda.Stream()
    bufs = [torch.full((8 << 20,), float(i + 1), device="cuda") for i in range(4)]   # 32 MiB buckets
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for b in bufs:
            dist.all_reduce(b)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            for b in bufs:
                b.mul_(2.0)
                dist.all_reduce(b)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        rep["captured_all_reduce_ok"] = bool(all((b == (i + 1) * 8.0).all() for i, b in enumerate(bufs)))
    except Exception as e:   # noqa: BLE001
        rep["captured_all_reduce_ok"] = False
        rep["capture_error"] = repr(e)[:500]
    dist.destroy_process_group()
except Exception as e:   # noqa: BLE001
    rep["error"] = repr(e)[:800]
print(json.dumps(rep))
