"""Install the HIP operators under the module names the reference imports, without editing the
reference tree:

    import lion_amd.shim; lion_amd.shim.install()
    import models.pvcnn2_ada            # the reference's own modules now run on liblion_hip.so

Registers in ``sys.modules``:
  third_party.pvcnn.functional.backend            -> ``_backend`` (12 callables of bindings.cpp:10-37)
  third_party.ChamferDistancePytorch.chamfer3D.dist_chamfer_3D  -> chamfer_3DDist & friends
  third_party.PyTorchEMD.backend                  -> ``emd_cuda_dynamic`` (approxmatch/matchcost API)
The reference's autograd wrappers (voxelization.py, grouping.py, ...) are left untouched: they pull
``_backend`` from the first module."""
import sys
import types


def install():
    from .functional import backend as _bk
    from . import chamfer3d, emd

    m = types.ModuleType("third_party.pvcnn.functional.backend")
    m._backend = _bk._backend
    m.__all__ = ["_backend"]
    sys.modules["third_party.pvcnn.functional.backend"] = m

    c = types.ModuleType("third_party.ChamferDistancePytorch.chamfer3D.dist_chamfer_3D")
    for name in chamfer3d.__all__:
        setattr(c, name, getattr(chamfer3d, name))
    sys.modules["third_party.ChamferDistancePytorch.chamfer3D.dist_chamfer_3D"] = c

    e = types.ModuleType("third_party.PyTorchEMD.backend")
    e.emd_cuda_dynamic = emd.emd_ext
    sys.modules["third_party.PyTorchEMD.backend"] = e
    return m, c, e
