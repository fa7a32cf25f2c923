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
