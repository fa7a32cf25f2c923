"""-m gpu: the metric drivers (callers of E1 / E2) on small synthetic sets."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _clouds(n, pts, seed, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, pts, 3, generator=g) + shift


def test_pairwise_cd_matches_bruteforce_and_emd_matches_oracle(orc):
    from lion_amd.metrics import pairwise_distance
    s, r = _clouds(5, 200, 0), _clouds(7, 200, 1)
    cd = pairwise_distance('CD', s, r, pair_batch=16).cpu()
    P = ((s[:, None, :, None, :].double() - r[None, :, None, :, :].double()) ** 2).sum(-1)  # [5,7,200,200]
    ref = P.min(3)[0].mean(2) + P.min(2)[0].mean(2)
    np.testing.assert_allclose(cd.numpy(), ref.numpy(), rtol=1e-5, atol=1e-7)
    emd = pairwise_distance('EMD', s, r, pair_batch=16).cpu().numpy()
    a = s[:, None].expand(-1, 7, -1, -1).reshape(-1, 200, 3).numpy()
    b = r[None].expand(5, -1, -1, -1).reshape(-1, 200, 3).numpy()
    o = orc.matchcost(a, b, orc.approxmatch(a, b)).reshape(5, 7) / 200.0
    np.testing.assert_allclose(emd, o, rtol=2e-4)


def test_compute_all_metrics_separates_distributions():
    from lion_amd.metrics import EMD_CD, compute_all_metrics
    ref = _clouds(12, 256, 10)
    same = _clouds(12, 256, 11)               # same distribution as ref
    far = _clouds(12, 256, 12, shift=0.8)     # shifted clouds: trivially separable
    r_same = compute_all_metrics(same, ref, batch_size=64)
    r_far = compute_all_metrics(far, ref, batch_size=64)
    for k in ('lgan_mmd-CD', 'lgan_cov-CD', '1-NN-CD-acc', 'lgan_mmd-EMD', 'lgan_cov-EMD', '1-NN-EMD-acc'):
        assert k in r_same and np.isfinite(r_same[k])
    assert r_far['1-NN-CD-acc'] == 1.0 and r_far['1-NN-EMD-acc'] == 1.0
    assert r_same['1-NN-CD-acc'] < 0.9
    assert r_far['lgan_mmd-CD'] > 5 * r_same['lgan_mmd-CD']
    paired = EMD_CD(same, same.clone(), batch_size=8)
    assert paired['MMD-CD'].item() == 0.0 and paired['MMD-EMD'].item() < 1e-5


def test_compute_all_metrics_matches_the_reference_golden():
    """the whole driver on the HIP Chamfer / EMD kernels against tests/golden/metrics.npz, the output of the REFERENCE's
    compute_all_metrics on the same clouds (make_golden_metrics.py): pair matrices, MMD / COV / 1-NNA under both
    distances, and the device JSD."""
    import os
    from conftest import GOLDEN
    from lion_amd import metrics
    z = np.load(os.path.join(GOLDEN, "metrics.npz"))
    smp, ref = torch.from_numpy(z["smp"]), torch.from_numpy(z["ref"])
    M = metrics.pairwise_distance("CD", ref, smp, pair_batch=37).cpu().numpy()     # ragged last batch
    np.testing.assert_allclose(M, z["M_rs_CD"], rtol=1e-5, atol=1e-8)
    M = metrics.pairwise_distance("EMD", ref, smp, pair_batch=64).cpu().numpy()
    np.testing.assert_allclose(M, z["M_rs_EMD"], rtol=5e-4)                        # v_exp_f32 vs expf in the auction
    res = metrics.compute_all_metrics(smp, ref, batch_size=64)
    for k in res:
        want = float(z["res/" + k])
        tol = 1e-5 if "CD" in k else 1e-3
        assert abs(res[k] - want) <= tol * max(abs(want), 1e-3), (k, res[k], want)
    assert set(res) == {k[4:] for k in z.files if k.startswith("res/")}
    jsd = metrics.jsd_between_point_cloud_sets(smp.cuda(), ref.cuda())
    assert abs(jsd - float(z["jsd"])) < 1e-6
