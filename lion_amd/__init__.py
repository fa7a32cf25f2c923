"""lion_amd -- MI355X (gfx950) native hot path of nv-tlabs/LION.

Sub-packages:
  csrc/        hand-written HIP kernels + the C ABI (include/lion_hip.h) -> liblion_hip.so
  functional/  operator API, same names as the reference's third_party.pvcnn.functional
  chamfer3d, emd   the reference's chamfer_3D / emd_ext operator modules
  models/      host-side mirror of models.pvcnn2_ada / latent_points_ada / score_sde / vae_adain / lion
  diffusion    DiffusionDiscretized (utils/diffusion_pvd.py) with a fused per-step update
  dist         data-parallel gradient step (utils/utils.py:717-770) over RCCL

Nothing here falls back to a CPU implementation: without the built HIP library every
operator raises RuntimeError.
"""
__version__ = "0.1.0"


def invalidate_weight_caches() -> None:
    """Call after writing model weights through ``.data`` (which hides the write from the version counter):
    packed / mirrored weights, style plans and captured graphs are rebuilt on next use (lion_amd/_wcache.py)."""
    from . import _wcache
    _wcache.invalidate_all()


def _poison_uninitialised():
    """LION_DEBUG_POISON=1: every torch.empty / empty_like on the GPU is filled with NaN (float) / a large negative
    number (int32), so that a kernel reading memory it was supposed to overwrite first shows up as NaN / an index far
    out of range instead of as a rare, allocator-history-dependent mismatch.  Debug aid for the -m gpu suite."""
    import torch
    real_empty, real_like = torch.empty, torch.empty_like

    def fill(t):
        if t.is_cuda and t.numel():
            if t.is_floating_point():
                t.fill_(float("nan"))
            elif t.dtype in (torch.int32, torch.int16):
                t.fill_(-30000)
        return t

    def empty(*a, **k):
        return fill(real_empty(*a, **k))

    def empty_like(*a, **k):
        return fill(real_like(*a, **k))

    torch.empty, torch.empty_like = empty, empty_like


import os as _os
if _os.environ.get("LION_DEBUG_POISON") == "1":
    _poison_uninitialised()
