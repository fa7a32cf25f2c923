"""-m gpu: the atomics-free backward scatters (csrc/scatter_csr.hip) behind grouping / 3-NN interpolation / devoxelize
backward: against the sum carried out in float64, against the legacy LDS-atomic entry points of the C ABI (which stay the
fallback), bit-for-bit reproducible, on ragged shapes (bins not a multiple of 8, E not a multiple of the workgroup, empty
bins, one bin holding most entries).  Reference: grouping.cu:58-80, neighbor_interpolate.cu:145-170, trilinear_devox.cu:119-162."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(gy, idx, w, S, bins):
    B, C = gy.shape[:2]
    g = gy.reshape(B, C, -1).double().cpu().numpy()
    ix = idx.reshape(B, -1).cpu().numpy()
    ww = None if w is None else w.reshape(B, -1).double().cpu().numpy()
    out = np.zeros((B, C, bins))
    E = ix.shape[1]
    src = np.arange(E) % S
    for b in range(B):
        contrib = g[b][:, src] * (1.0 if ww is None else ww[b][None, :])
        for c in range(C):
            np.add.at(out[b, c], np.clip(ix[b], 0, bins - 1), contrib[c])
    return out


@pytest.mark.parametrize("B,C,N,M,U", [(2, 35, 2048, 1024, 32), (3, 7, 203, 51, 5), (1, 67, 16, 16, 4)])
def test_grouping_backward_csr(B, C, N, M, U):
    from lion_amd import _lib
    from lion_amd.functional.backend import _backend as bk
    g = torch.Generator(device="cuda").manual_seed(N + M)
    idx = torch.randint(0, N, (B, M, U), device="cuda", dtype=torch.int32, generator=g)
    idx[:, : M // 2] = 3 % N                       # one bin holds half of all entries
    gy = torch.randn(B, C, M, U, device="cuda", generator=g)
    got = bk.grouping_backward(gy, idx, N)
    again = bk.grouping_backward(gy, idx, N)
    assert torch.equal(got, again), "not reproducible"
    ref = _ref(gy, idx, None, M * U, N)
    assert np.abs(got.double().cpu().numpy() - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1.0)
    legacy = torch.empty(B, C, N, device="cuda")
    _lib.check(_lib.load().lion_grouping_backward(_lib.ptr(gy), _lib.ptr(idx), B, C, N, M, U, _lib.ptr(legacy),
                                                  _lib.stream_ptr(gy.device)), "grouping_backward")
    assert (legacy - got).abs().max().item() <= 1e-4 * max(got.abs().max().item(), 1.0)


@pytest.mark.parametrize("B,C,N,M", [(2, 192, 2048, 1024), (3, 5, 333, 77), (1, 16, 64, 16)])
def test_three_nn_interpolate_backward_csr(B, C, N, M):
    from lion_amd import _lib
    from lion_amd.functional.backend import _backend as bk
    g = torch.Generator(device="cuda").manual_seed(N * 7 + M)
    idx = torch.randint(0, M, (B, 3, N), device="cuda", dtype=torch.int32, generator=g)
    w = torch.rand(B, 3, N, device="cuda", generator=g)
    gy = torch.randn(B, C, N, device="cuda", generator=g)
    got = bk.three_nearest_neighbors_interpolate_backward(gy, idx, w, M)
    assert torch.equal(got, bk.three_nearest_neighbors_interpolate_backward(gy, idx, w, M))
    ref = _ref(gy, idx, w, N, M)
    assert np.abs(got.double().cpu().numpy() - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1.0)
    legacy = torch.empty(B, C, M, device="cuda")
    _lib.check(_lib.load().lion_three_nn_interpolate_backward(_lib.ptr(gy), _lib.ptr(idx), _lib.ptr(w), B, C, N, M,
                                                              _lib.ptr(legacy), _lib.stream_ptr(gy.device)), "legacy")
    assert (legacy - got).abs().max().item() <= 1e-4 * max(got.abs().max().item(), 1.0)


@pytest.mark.parametrize("B,C,N,r", [(2, 64, 2048, 32), (2, 16, 500, 16), (1, 8, 64, 8)])
def test_trilinear_devoxelize_backward_csr(B, C, N, r):
    from lion_amd import _lib
    from lion_amd.functional.backend import _backend as bk
    g = torch.Generator(device="cuda").manual_seed(N + r)
    r3 = r ** 3
    inds = torch.randint(0, r3, (B, 8, N), device="cuda", dtype=torch.int32, generator=g)
    wg = torch.rand(B, 8, N, device="cuda", generator=g)
    gy = torch.randn(B, C, N, device="cuda", generator=g)
    # the product op keeps the LDS-atomic kernel here (faster for 32768 mostly empty bins); the atomics-free path itself:
    got = bk._scatter_csr(gy, inds, wg, N, 8 * N, r3, "devoxelize_backward (csr)")
    assert got is not None and torch.equal(got, bk._scatter_csr(gy, inds, wg, N, 8 * N, r3, "devoxelize_backward (csr)"))
    ref = _ref(gy, inds, wg, N, r3)
    assert np.abs(got.double().cpu().numpy() - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1.0)
    legacy = torch.empty(B, C, r3, device="cuda")
    _lib.check(_lib.load().lion_trilinear_devoxelize_backward(_lib.ptr(gy), _lib.ptr(inds), _lib.ptr(wg), B, C, N, r3,
                                                              _lib.ptr(legacy), _lib.stream_ptr(gy.device)), "legacy")
    assert (legacy - got).abs().max().item() <= 1e-4 * max(got.abs().max().item(), 1.0)
