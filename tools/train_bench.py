"""milliseconds per VAE training step (forward + backward + Adam) at B x 2048 points on one GPU; --lib-backward
routes the Conv3d gradients through ATen/MIOpen (the state before the MFMA dgrad / wgrad kernels)."""
import argparse, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--lib-backward", action="store_true")
a = ap.parse_args()
from lion_amd import conv_ops
from lion_amd.config import released_prior_cfg
from lion_amd.models.vae_adain import Model as VAE
from lion_amd.training import vae_train_step
if a.lib_backward:
    conv_ops.supported_backward = False
    _orig = conv_ops.supported
    class _F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, b):
            ctx.save_for_backward(x, w); ctx.hb = b is not None
            return conv_ops.conv3d_k3(x, w, b)
        @staticmethod
        def backward(ctx, gy):
            x, w = ctx.saved_tensors
            return torch.ops.aten.convolution_backward(gy.contiguous(), x, w, [w.shape[0]] if ctx.hb else None, [1, 1, 1], [1, 1, 1], [1, 1, 1], False, [0, 0, 0], 1, [True, True, ctx.hb])
    conv_ops._Conv3dK3 = _F
torch.manual_seed(0)
cfg = released_prior_cfg()
vae = VAE(cfg).cuda().train()
opt = torch.optim.Adam(vae.parameters(), lr=1e-4)
x = torch.randn(a.batch, 2048, 3, device="cuda")
for _ in range(2):
    vae_train_step(vae, opt, x, step=0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(a.steps):
    vae_train_step(vae, opt, x, step=i)
torch.cuda.synchronize()
print(f"vae_train_step B={a.batch}: {(time.perf_counter() - t0) / a.steps * 1e3:.1f} ms/step ({'library' if a.lib_backward else 'MFMA'} conv backward)")
