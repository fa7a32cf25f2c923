"""torch.profiler table of ONE eager VAE training step (forward + backward + Adam) at B x 2048: ATen / own ops by GPU time,
with input shapes, and the python call sites of the copy / reduction ops (where do the .contiguous() copies and the
bias-gradient sums come from).  usage: train_op_profile.py [--B 32] [--rows 40]"""
import argparse
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd import training  # noqa: E402
from lion_amd.config import released_prior_cfg  # noqa: E402
from lion_amd.dist import BucketedGradAverager  # noqa: E402
from lion_amd.models.vae_adain import Model as VAE  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--rows", type=int, default=40)
a = ap.parse_args()
dev = torch.device("cuda")
cfg = released_prior_cfg("chair")
torch.manual_seed(0)
model = VAE(cfg).to(dev).train()
params = list(model.parameters())
opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.99))
averager = BucketedGradAverager(params)
x = torch.randn(a.B, 2048, 3, device=dev)
for _ in range(3):
    training.vae_train_step(model, opt, x, step=0, averager=averager, distributed=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    training.vae_train_step(model, opt, x, step=0, averager=averager, distributed=False)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=a.rows, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=a.rows, max_name_column_width=50, max_src_column_width=110))
