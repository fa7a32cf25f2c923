"""tests/golden/metrics.npz: outputs of the REFERENCE's own evaluation driver -- utils/evaluation_metrics_fast.py
``compute_all_metrics`` (:463-560, with its ``_pairwise_EMD_CD_`` / ``lgan_mmd_cov`` / ``knn``) and
``jsd_between_point_cloud_sets`` (:587-601) -- on small synthetic sets, run on the CPU in the build container.
Only the two CUDA operator modules are substituted, at the module boundary the reference imports them through:
  third_party.ChamferDistancePytorch.chamfer3D.dist_chamfer_3D -> the reference's own pure-torch chamfer_python.distChamfer
  third_party.PyTorchEMD.emd / emd_nograd                       -> the C oracle (approxmatch + matchcost, / N), which is pinned
                                                                   bit-exact to the reference's kernel bodies (oracle/_ref)
and ``Tensor.cuda()`` is made the identity.  Everything else (pair batching, MMD / COV / 1-NNA, JSD) is reference code.
Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_metrics.py"""
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "_stubs"), REF, ROOT]

import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self

from third_party.ChamferDistancePytorch import chamfer_python  # noqa: E402  (reference, pure torch)

orc = oracle.lib()


class _Chamfer(torch.nn.Module):
    def forward(self, a, b):
        return chamfer_python.distChamfer(a, b)


cm = types.ModuleType("third_party.ChamferDistancePytorch.chamfer3D.dist_chamfer_3D")
cm.chamfer_3DDist = cm.chamfer_3DDist_nograd = _Chamfer
sys.modules["third_party.ChamferDistancePytorch.chamfer3D.dist_chamfer_3D"] = cm


def _emd(xyz1, xyz2, transpose=True):
    if transpose:
        xyz1, xyz2 = xyz1.transpose(1, 2), xyz2.transpose(1, 2)
    a, b = xyz1.contiguous().numpy().astype(np.float32), xyz2.contiguous().numpy().astype(np.float32)
    cost = orc.matchcost(a, b, orc.approxmatch(a, b))
    return torch.from_numpy(cost / np.float32(a.shape[1]))          # emd_nograd.py:43 divides by N


for name in ("third_party.PyTorchEMD.emd", "third_party.PyTorchEMD.emd_nograd"):
    m = types.ModuleType(name)
    m.earth_mover_distance = m.earth_mover_distance_nograd = _emd
    sys.modules[name] = m
eh = types.ModuleType("utils.exp_helper")


class ExpTimer:
    def __init__(self, *a, **k): pass
    def tic(self): pass
    def toc(self): pass
    def hours_left(self): return 0.0


eh.ExpTimer = ExpTimer
sys.modules["utils.exp_helper"] = eh

from utils import evaluation_metrics_fast as ref  # noqa: E402


def main():
    g = torch.Generator().manual_seed(0)
    n_ref, n_smp, pts = 12, 12, 128
    base = torch.rand(n_ref, pts, 3, generator=g)
    ref_pcs = base * torch.tensor([1.0, 0.4, 0.7]) - 0.5
    smp_pcs = (torch.rand(n_smp, pts, 3, generator=g) * torch.tensor([1.0, 0.5, 0.6]) - 0.5) * 1.1
    smp_pcs[:4] = ref_pcs[:4] + 0.01 * torch.randn(4, pts, 3, generator=g)   # some near-duplicates of references
    out = {"ref": ref_pcs.numpy(), "smp": smp_pcs.numpy()}
    res = ref.compute_all_metrics(smp_pcs, ref_pcs, batch_size=n_ref, verbose=False, accelerated_cd=True)
    for k, v in res.items():
        out["res/" + k] = np.float64(v)
    for metric in ("CD", "EMD"):
        M_rs, _ = ref._pairwise_EMD_CD_(metric, ref_pcs, smp_pcs, n_ref, accelerated_cd=True, require_grad=False, verbose=False)
        M_rr, _ = ref._pairwise_EMD_CD_(metric, ref_pcs, ref_pcs, n_ref, accelerated_cd=True, require_grad=False, verbose=False)
        M_ss, _ = ref._pairwise_EMD_CD_(metric, smp_pcs, smp_pcs, n_ref, accelerated_cd=True, require_grad=False, verbose=False)
        out[f"M_rs_{metric}"], out[f"M_rr_{metric}"], out[f"M_ss_{metric}"] = M_rs.numpy(), M_rr.numpy(), M_ss.numpy()
        for k, v in ref.knn(M_rr, M_rs, M_ss, 1, sqrt=False).items():
            out[f"knn_{metric}/{k}"] = np.float64(v)
        for k, v in ref.lgan_mmd_cov(M_rs.t()).items():
            out[f"mmdcov_{metric}/{k}"] = np.float64(v)
    out["jsd"] = np.float64(ref.jsd_between_point_cloud_sets(smp_pcs.numpy(), ref_pcs.numpy()))
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)
    print({k: float(v) for k, v in out.items() if np.ndim(v) == 0})


if __name__ == "__main__":
    main()
