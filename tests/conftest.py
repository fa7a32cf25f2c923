import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def pytest_collection_modifyitems(config, items):
    """No test may hang a run: 600 s per test (the first GPU test pays the 1-2 min first `import torch` of a fresh box),
    enforced by pytest-timeout's thread method -- it dumps every thread's stack and ends the process, which also gets
    out of a wait inside a native call.  (One full GPU run at the end of round 2 sat silent for 1000 s.)"""
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600, method="thread"))


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


def fill_(module, seed=0):
    """Same name-derived deterministic weights as tests/golden/make_golden.py:fill_."""
    import zlib
    import torch
    with torch.no_grad():
        for name, t in sorted(module.state_dict().items()):
            if not t.is_floating_point():
                continue
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + seed) & 0x7fffffff)
            r = torch.randn(t.shape, generator=g)
            if t.dim() >= 2:
                fan_in = t[0].numel()
                t.copy_((r * (0.8 / np.sqrt(fan_in))).to(t.device))
            elif name.endswith("norm.weight") or "normalize" in name and name.endswith("weight"):
                t.copy_((1.0 + 0.1 * r).to(t.device))
            elif name.endswith("emd.bias"):
                v = 0.1 * r
                v[: t.numel() // 2] += 1.0
                t.copy_(v.to(t.device))
            else:
                t.copy_((0.1 * r).to(t.device))


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def oracle_backend(monkeypatch):
    """Runs lion_amd's nn.Modules on the CPU by injecting the oracle as the operator backend.
    TEST-ONLY: the product has no CPU path (lion_amd never imports oracle)."""
    import oracle
    import lion_amd.functional.backend as bk
    monkeypatch.setattr(bk, "_backend", oracle.TorchBackend())
    return bk._backend
