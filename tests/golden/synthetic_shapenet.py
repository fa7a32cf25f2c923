"""A tiny ShapeNetCore.v2.PC15k-shaped tree of .npy clouds, regenerated from a seed by both the golden generator
(make_golden_dataset.py, which feeds it to the REFERENCE's dataset class) and the test (which feeds it to ours)."""
import os

import numpy as np

POINTS = 60
LAYOUT = {"02691156": {"train": 7, "val": 3}, "03001627": {"train": 5, "val": 2}}  # airplane, chair


def write_tree(root, seed=7):
    rs = np.random.RandomState(seed)
    for synset, splits in LAYOUT.items():
        for split, count in splits.items():
            folder = os.path.join(root, synset, split)
            os.makedirs(folder, exist_ok=True)
            for _ in range(count):
                stem = "%032x" % int(rs.randint(0, 2 ** 31 - 1))  # unsorted names: the loader has to sort them
                cloud = rs.randn(POINTS, 3) * rs.uniform(0.2, 1.5, size=(1, 3)) + rs.uniform(-1, 1, size=(1, 3))
                np.save(os.path.join(folder, stem + ".npy"), cloud.astype(np.float32))
            open(os.path.join(folder, "README.txt"), "w").write("not a cloud\n")  # non-.npy entries are skipped
    return root


MODES = {
    "global": dict(normalize_global=True),
    "global_axis": dict(normalize_global=True, normalize_std_per_axis=True),
    "per_shape": dict(normalize_per_shape=True),
    "per_shape_axis": dict(normalize_per_shape=True, normalize_std_per_axis=True),
    "shape_box": dict(normalize_shape_box=True),
    "recenter": dict(recenter_per_shape=True),
}
