"""lion_amd.shim.install(): the advertised zero-edit drop-in.  After it, the REFERENCE's own autograd wrappers
(third_party/pvcnn/functional/*.py, PyTorchEMD/emd.py, ChamferDistancePytorch) must resolve their native module to
liblion_hip.so's operator objects.  Wiring only (no GPU here): run in a fresh interpreter with /root/reference on the
path, never writing into that tree (python -B)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "third_party", "pvcnn")),
                                reason="/root/reference not present (build container only)")

SCRIPT = r'''
import sys, torch
import lion_amd.shim
m, c, e = lion_amd.shim.install()
from lion_amd.functional.backend import _backend as ours
from lion_amd import chamfer3d, emd

# 1. the reference's package: every wrapper module pulled OUR _backend (bindings.cpp:10-37 names)
import third_party.pvcnn.functional as F
assert F.__file__.startswith("/root/reference/"), F.__file__
# (the package rebinds names like `ball_query` / `grouping` to the functions: fetch the modules from sys.modules)
for short in ("voxelization", "devoxelization", "ball_query", "grouping", "sampling", "interpolatation"):
    mod = sys.modules["third_party.pvcnn.functional." + short]
    assert mod.__file__.startswith("/root/reference/"), mod.__file__
    assert mod._backend is ours, mod.__name__
for name in ("gather_features_forward", "gather_features_backward", "furthest_point_sampling", "ball_query",
             "grouping_forward", "grouping_backward", "three_nearest_neighbors_interpolate_forward",
             "three_nearest_neighbors_interpolate_backward", "trilinear_devoxelize_forward",
             "trilinear_devoxelize_backward", "avg_voxelize_forward", "avg_voxelize_backward"):
    assert callable(getattr(ours, name)), name

# 2. calling the reference's wrapper reaches our operator: CPU tensors -> our "no CPU path" error
try:
    F.avg_voxelize(torch.zeros(1, 2, 4), torch.zeros(1, 3, 4, dtype=torch.int32), 4)
except RuntimeError as ex:
    assert "lion_amd" in str(ex), ex
else:
    raise AssertionError("the reference wrapper did not reach lion_amd's backend")

# 3. EMD: the reference's autograd Function binds emd_ext
import third_party.PyTorchEMD.emd as ref_emd
assert ref_emd.__file__.startswith("/root/reference/")
assert ref_emd.emd_cuda is emd.emd_ext
for name in ("approxmatch_forward", "matchcost_forward", "matchcost_backward"):
    assert callable(getattr(ref_emd.emd_cuda, name))

# 4. Chamfer: importing the reference's module path yields ours (no JIT build, no write into the tree)
from third_party.ChamferDistancePytorch.chamfer3D.dist_chamfer_3D import chamfer_3DDist
assert chamfer_3DDist is chamfer3d.chamfer_3DDist
print("WIRED")
'''


def test_install_rewires_the_reference_wrappers():
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, REF])
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    out = subprocess.run([sys.executable, "-B", "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "WIRED" in out.stdout
