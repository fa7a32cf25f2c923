"""Input-side data format: lion_amd.data vs the reference's ShapeNet15kPointClouds on the same synthetic tree
(tests/golden/dataset.npz, produced by the reference's own class: tests/golden/make_golden_dataset.py)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLDEN)
from synthetic_shapenet import LAYOUT, MODES, POINTS, write_tree  # noqa: E402

from lion_amd import data as D  # noqa: E402


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    return write_tree(str(tmp_path_factory.mktemp("pc15k")))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "dataset.npz"))


def _pair(tree, **kw):
    tr = D.ShapeNet15kPointClouds(categories=["airplane", "chair"], split="train", tr_sample_size=16,
                                  te_sample_size=16, root_dir=tree, **kw)
    va = D.ShapeNet15kPointClouds(categories=["airplane", "chair"], split="val", tr_sample_size=16, te_sample_size=16,
                                  all_points_mean=tr.all_points_mean, all_points_std=tr.all_points_std,
                                  root_dir=tree, **kw)
    return tr, va


@pytest.mark.parametrize("mode", list(MODES))
def test_dataset_equals_reference(tree, gold, mode):
    for tag, ds in zip(("train", "val"), _pair(tree, **MODES[mode])):
        p = f"{mode}.{tag}."
        assert len(ds) == int(gold[p + "len"])
        assert ["|".join(m) for m in ds.all_cate_mids] == list(gold[p + "mids"])  # sort + seeded shuffle
        assert np.array_equal(np.asarray(ds.cate_idx_lst), gold[p + "cate_idx"])
        for name, mine in (("all_points", ds.all_points), ("mean", ds.all_points_mean), ("std", ds.all_points_std)):
            ref = gold[p + name]
            assert np.shape(mine) == ref.shape and np.asarray(mine).dtype == ref.dtype, name
            assert np.array_equal(np.asarray(mine), ref), name  # same numpy expressions: bit-identical
        item = ds[2]
        assert np.array_equal(item["tr_points"].numpy(), gold[p + "item.tr_points"])
        assert np.array_equal(np.asarray(item["mean"]), gold[p + "item.mean"])
        assert np.array_equal(np.asarray(item["std"]), gold[p + "item.std"])
        assert item["input_pts"] is item["tr_points"] and item["display_axis_order"] == [0, 1, 2]


def test_single_category_and_size_caps(tree, gold):
    one = D.ShapeNet15kPointClouds(categories="chair", split="train", tr_sample_size=20000, te_sample_size=20000,
                                   normalize_global=True, root_dir=tree)
    assert ["|".join(m) for m in one.all_cate_mids] == list(gold["chair_only.mids"])
    assert [one.tr_sample_size, one.te_sample_size] == list(gold["chair_only.sizes"]) == [10000, 5000]
    assert len(D.synsetid_to_cate) == 55 and D.cate_to_synsetid["airplane"] == "02691156"
    with pytest.raises(NotImplementedError):
        D.ShapeNet15kPointClouds(categories="chair", root_dir=tree)  # no normalisation mode selected
    with pytest.raises(NotImplementedError):
        D.ShapeNet15kPointClouds(categories="chair", normalize_global=True, clip_forge_enable=1, root_dir=tree)
    with pytest.raises(ValueError):
        D.ShapeNet15kPointClouds(categories="car", normalize_global=True, root_dir=tree)  # category not on disk
    with pytest.raises(FileNotFoundError):
        D.ShapeNet15kPointClouds(categories="chair", normalize_global=True, root_dir=tree + "/nope")


def test_random_subsample_and_renormalize(tree):
    kw = dict(categories="airplane", split="train", tr_sample_size=40, normalize_global=True, root_dir=tree)
    np.random.seed(0)
    item = D.ShapeNet15kPointClouds(random_subsample=True, sample_with_replacement=0, **kw)[1]
    assert len(set(item["select_idx"].tolist())) == 40  # a permutation prefix: no repeats
    ds = D.ShapeNet15kPointClouds(random_subsample=True, sample_with_replacement=1, **kw)
    item = ds[1]
    assert item["select_idx"].shape == (40,) and item["select_idx"].max() < POINTS
    assert np.array_equal(item["tr_points"].numpy(), ds.train_points[1][item["select_idx"]].astype(np.float32))
    raw = ds.all_points * ds.all_points_std + ds.all_points_mean
    ds.renormalize(np.zeros((1, 1, 3), np.float32), np.full((1, 1, 1), 2.0, np.float32))
    np.testing.assert_allclose(ds.all_points, raw / 2.0, rtol=1e-6, atol=1e-6)
    assert ds.train_points.base is ds.all_points or np.shares_memory(ds.train_points, ds.all_points)


def _data_cfg(tree, **over):
    c = dict(cates=["airplane", "chair"], tr_max_sample_points=16, te_max_sample_points=16, dataset_scale=1,
             normalize_shape_box=False, normalize_per_shape=False, normalize_std_per_axis=False,
             normalize_global=True, recenter_per_shape=False, random_subsample=1, sample_with_replacement=1,
             clip_forge_enable=0, clip_model="ViT-B/32", batch_size=4, batch_size_test=3, num_workers=0,
             train_drop_last=1, root_dir=tree)
    c.update(over)
    return types.SimpleNamespace(**c)


def test_loaders(tree, gold):
    args = types.SimpleNamespace(distributed=False, eval_trainnll=False)
    loaders = D.get_data_loaders(_data_cfg(tree), args)
    batches = list(loaders["train_loader"])
    assert len(batches) == 3 and batches[0]["tr_points"].shape == (4, 16, 3)  # 12 shapes, drop_last
    assert batches[0]["tr_points"].dtype == torch.float32
    te = list(loaders["test_loader"])
    assert [b["tr_points"].shape[0] for b in te] == [3, 2]
    # evaluation split: first 16 points, training statistics
    got = torch.cat([b["tr_points"] for b in te]).numpy()
    assert np.array_equal(got, gold["global.val.all_points"][:, :16].astype(np.float32))
    assert list(te[0]["mid"]) == [m.split("|")[1] for m in gold["global.val.mids"][:3]]


def test_resident_set_on_cpu_device(tree):
    tr, _ = _pair(tree, random_subsample=True, **MODES["global"])
    res = D.ResidentPointClouds(tr, "cpu", batch_size=5, seed=3, drop_last=False)
    seen = []
    for b in res.epoch(0):
        assert b["tr_points"].shape[1:] == (16, 3) and b["tr_points"].dtype == torch.float32
        pool = torch.from_numpy(tr.train_points)
        assert torch.equal(b["tr_points"], pool[b["idx"].unsqueeze(1), b["select_idx"]].float())
        assert b["mean"].shape == (b["idx"].numel(), 1, 3) and torch.equal(b["cate_idx"], res.cate_idx[b["idx"]])
        seen += b["idx"].tolist()
    assert sorted(seen) == list(range(12)) and len(res) == 3
    again = [b["idx"].tolist() for b in res.epoch(0)]
    assert sum(again, []) == seen  # same (seed, epoch) -> same order
    assert sum([b["idx"].tolist() for b in res.epoch(1)], []) != seen
    # shards of a 2-rank job: disjoint, together the whole set, same order as torch's DistributedSampler
    shards = [sum([b["idx"].tolist() for b in
                   D.ResidentPointClouds(tr, "cpu", 3, rank=r, world_size=2, seed=3, drop_last=False).epoch(4)], [])
              for r in (0, 1)]
    assert sorted(shards[0] + shards[1]) == list(range(12))
    smp = torch.utils.data.distributed.DistributedSampler(range(12), num_replicas=2, rank=1, shuffle=True, seed=3)
    smp.set_epoch(4)
    assert shards[1] == list(smp)
    # without replacement: every row is a set; fixed prefix when the dataset does not sub-sample
    tr.sample_with_replacement = 0
    pick = D.ResidentPointClouds(tr, "cpu", 4).pick_points(6)
    assert all(len(set(row.tolist())) == 16 for row in pick)
    tr.random_subsample = False
    assert torch.equal(D.ResidentPointClouds(tr, "cpu", 4).pick_points(2), torch.arange(16).expand(2, 16))
    # per-shape statistics travel with the shapes
    ps, _ = _pair(tree, **MODES["per_shape_axis"])
    b = D.ResidentPointClouds(ps, "cpu", 4, shuffle=False).batch([5, 0])
    assert torch.equal(b["mean"], torch.from_numpy(np.asarray(ps.all_points_mean)[[5, 0]]))
    assert b["std"].shape == (2, 1, 3)
