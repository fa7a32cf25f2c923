"""tests/golden/dataset.npz: what the REFERENCE's ShapeNet15kPointClouds (datasets/pointflow_datasets.py:88-355)
returns for the synthetic tree of synthetic_shapenet.py, for every normalisation mode, train split and a val split
normalised with the training statistics (get_datasets :372-413).  The reference module is imported read-only;
``open3d`` (unused by the class) and ``loguru`` are stubbed.
Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_dataset.py
"""
import os
import sys
import tempfile
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "_stubs"), "/root/reference", HERE]
sys.modules["open3d"] = types.ModuleType("open3d")
# the reference's ``datasets`` directory has no __init__.py and would lose to the installed HuggingFace package
_pkg = types.ModuleType("datasets")
_pkg.__path__ = ["/root/reference/datasets"]
sys.modules["datasets"] = _pkg

import numpy as np  # noqa: E402

from synthetic_shapenet import MODES, write_tree  # noqa: E402


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        write_tree(os.path.join(tmp, "data", "ShapeNetCore.v2.PC15k"))
        os.chdir(tmp)  # the reference looks for ./data/ShapeNetCore.v2.PC15k/ (datasets/data_path.py:13-16)
        from datasets.pointflow_datasets import ShapeNet15kPointClouds as Ref
        for mode, kw in MODES.items():
            tr = Ref(categories=["airplane", "chair"], split="train", tr_sample_size=16, te_sample_size=16, **kw)
            va = Ref(categories=["airplane", "chair"], split="val", tr_sample_size=16, te_sample_size=16,
                     all_points_mean=tr.all_points_mean, all_points_std=tr.all_points_std, **kw)
            for tag, ds in (("train", tr), ("val", va)):
                p = f"{mode}.{tag}."
                out[p + "all_points"] = ds.all_points
                out[p + "mean"] = np.asarray(ds.all_points_mean)
                out[p + "std"] = np.asarray(ds.all_points_std)
                out[p + "cate_idx"] = np.asarray(ds.cate_idx_lst)
                out[p + "mids"] = np.asarray(["|".join(m) for m in ds.all_cate_mids])
                item = ds[2]
                out[p + "item.tr_points"] = item["tr_points"].numpy()
                out[p + "item.mean"], out[p + "item.std"] = np.asarray(item["mean"]), np.asarray(item["std"])
                out[p + "len"] = np.asarray(len(ds))
        one = Ref(categories="chair", split="train", tr_sample_size=20000, te_sample_size=20000, normalize_global=True)
        out["chair_only.mids"] = np.asarray(["|".join(m) for m in one.all_cate_mids])
        out["chair_only.sizes"] = np.asarray([one.tr_sample_size, one.te_sample_size])
    np.savez_compressed(os.path.join(HERE, "dataset.npz"), **out)
    print("wrote dataset.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
