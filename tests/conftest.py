import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; never imported by lion_amd)."""
    import oracle
    return oracle.lib()


def gaussian_cloud(rng, b, n):
    """[B,3,N] i.i.d. N(0,1): latent points at t=T look like this."""
    return rng.standard_normal((b, 3, n)).astype(np.float32)


def surface_cloud(rng, b, n):
    """Unit sphere + 0.01 noise: high voxel collision rate."""
    v = rng.standard_normal((b, 3, n))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return (v + 0.01 * rng.standard_normal((b, 3, n))).astype(np.float32)


def voxel_coords(rng, b, n, r, kind="gauss"):
    """float coords in voxel units [0, r-1] (what Voxelization.forward hands to devoxelize)."""
    c = gaussian_cloud(rng, b, n) if kind == "gauss" else surface_cloud(rng, b, n)
    c = c - c.mean(2, keepdims=True)
    nrm = np.sqrt((c ** 2).sum(1, keepdims=True)).max(2, keepdims=True)
    c = c / (np.maximum(nrm, 1e-12) * 2.0) + 0.5
    return np.clip(c * r, 0, r - 1).astype(np.float32)
