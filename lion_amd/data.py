"""ShapeNetCore.v2.PC15k point-cloud sets -- the data format on the input side of the hot path.

Host side: ``ShapeNet15kPointClouds`` / ``get_datasets`` / ``get_data_loaders`` mirror the reference's
``datasets/pointflow_datasets.py`` (class :88-355, builders :363-446): same constructor arguments, attributes
(``all_points``, ``all_points_mean/std``, ``train_points``, ``all_cate_mids``, ...), deterministic shuffle
(``random.Random(38383)``), normalisation modes and item dictionaries, so the trainers and the evaluation code
read identical tensors.  Checked against the reference's own class on a synthetic tree (tests/golden/make_golden_dataset.py
-> dataset.npz).

Device side: ``ResidentPointClouds``.  A category is 0.1-1.3 GB of fp32 points (all 55 categories: 6.3 GB), i.e.
nothing next to 288 GB of HBM, so instead of DataLoader workers + pinned copies every step the normalised set is
uploaded once and a batch is drawn where it is consumed: shape indices in the order of the reference's
(Distributed)Sampler, the ``tr_sample_size`` points per shape picked by an on-device generator, one gather.
Statistics of the draw equal the reference's (uniform with or without replacement); the random stream differs
(numpy per worker there, a device Philox stream seeded per (seed, epoch, rank) here).

The CLIP image branch of the reference (rendered views + ``clip.load`` preprocessing, :106-112, :330-344) needs the
external ``clip`` package and image files and is outside this path: ``clip_forge_enable`` raises here; the
text/image-conditioned priors take their ``clip_feat`` tensor from the caller.
"""
import os
import random

import numpy as np
import torch
from torch.utils import data
from torch.utils.data import Dataset

# ShapeNet synset id -> category (order defines the category index of categories=['all'])
_TAXONOMY = """
02691156:airplane 02773838:bag 02801938:basket 02808440:bathtub 02818832:bed 02828884:bench 02876657:bottle
02880940:bowl 02924116:bus 02933112:cabinet 02747177:can 02942699:camera 02954340:cap 02958343:car 03001627:chair
03046257:clock 03207941:dishwasher 03211117:monitor 04379243:table 04401088:telephone 02946921:tin_can
04460130:tower 04468005:train 03085013:keyboard 03261776:earphone 03325088:faucet 03337140:file 03467517:guitar
03513137:helmet 03593526:jar 03624134:knife 03636649:lamp 03642806:laptop 03691459:speaker 03710193:mailbox
03759954:microphone 03761084:microwave 03790512:motorcycle 03797390:mug 03928116:piano 03938244:pillow
03948459:pistol 03991062:pot 04004475:printer 04074963:remote_control 04090263:rifle 04099429:rocket
04225987:skateboard 04256520:sofa 04330267:stove 04530566:vessel 04554684:washer 02992529:cellphone
02843684:birdhouse 02871439:bookshelf
"""
synsetid_to_cate = dict(item.split(":") for item in _TAXONOMY.split())
cate_to_synsetid = {name: sid for sid, name in synsetid_to_cate.items()}

DEFAULT_ROOT = "./data/ShapeNetCore.v2.PC15k/"  # datasets/data_path.py:13-16
_SHUFFLE_SEED = 38383
_MAX_TRAIN_POINTS = 10000  # the first 10k of the 15k points are the training pool, :300-304
_MAX_TEST_POINTS = 5000


def data_root(root_dir=None):
    root = root_dir or os.environ.get("LION_DATA_ROOT") or DEFAULT_ROOT
    if not os.path.isdir(root):
        raise FileNotFoundError(f"point-cloud root not found: {root} (set LION_DATA_ROOT or pass root_dir)")
    return root


def _box_centre_and_halfspan(pts):
    """per shape: centre of the axis-aligned box [S,1,D] and half of its longest side [S,1,1]"""
    hi, lo = pts.max(axis=1, keepdims=True), pts.min(axis=1, keepdims=True)
    return (hi + lo) / 2, (hi - lo).max(axis=-1, keepdims=True) / 2


class ShapeNet15kPointClouds(Dataset):
    def __init__(self, categories=['airplane'], tr_sample_size=10000, te_sample_size=10000, split='train', scale=1.,
                 normalize_per_shape=False, normalize_shape_box=False, random_subsample=False,
                 sample_with_replacement=1, normalize_std_per_axis=False, normalize_global=False,
                 recenter_per_shape=False, all_points_mean=None, all_points_std=None, input_dim=3,
                 clip_forge_enable=0, clip_model=None, root_dir=None):
        if clip_forge_enable:
            raise NotImplementedError("rendered-image / CLIP preprocessing branch is outside the hot path; "
                                      "pass clip_feat tensors to the priors directly")
        if split not in ('train', 'test', 'val'):
            raise AssertionError(split)
        if scale != 1:
            raise AssertionError("Scale (!= 1) is deprecated")
        self.clip_forge_enable = 0
        self.root_dir = data_root(root_dir)
        self.split = split
        self.cates = [categories] if isinstance(categories, str) else list(categories)
        self.synset_ids = list(cate_to_synsetid.values()) if 'all' in self.cates \
            else [cate_to_synsetid[c] for c in self.cates]
        self.subdirs = self.synset_ids
        self.in_tr_sample_size, self.in_te_sample_size = tr_sample_size, te_sample_size
        self.scale, self.input_dim = scale, input_dim
        self.random_subsample, self.sample_with_replacement = random_subsample, sample_with_replacement
        self.normalize_shape_box, self.normalize_per_shape = normalize_shape_box, normalize_per_shape
        self.normalize_std_per_axis, self.recenter_per_shape = normalize_std_per_axis, recenter_per_shape
        self.gravity_axis = 1

        clouds, cate_idx, mids = [], [], []
        for ci, synset in enumerate(self.synset_ids):
            folder = os.path.join(self.root_dir, synset, split)
            if not os.path.isdir(folder):
                raise ValueError(f"check the data path: directory missing {folder}")
            for stem in sorted(f[:-4] for f in os.listdir(folder) if f.endswith('.npy')):
                clouds.append(np.load(os.path.join(folder, stem + '.npy')))  # (15k, 3)
                cate_idx.append(ci)
                mids.append((synset, os.path.join(split, stem)))  # "<split>/<model id>"
        # deterministic order that depends only on the number of shapes (:211-216)
        self.shuffle_idx = list(range(len(clouds)))
        random.Random(_SHUFFLE_SEED).shuffle(self.shuffle_idx)
        self.cate_idx_lst = [cate_idx[i] for i in self.shuffle_idx]
        self.all_cate_mids = [mids[i] for i in self.shuffle_idx]
        pts = np.stack([clouds[i] for i in self.shuffle_idx])  # [S, 15000, D]

        S, D = pts.shape[0], input_dim
        # precedence of the modes as in the reference (:225-283): box > per-shape > given stats > recentre > global
        if normalize_shape_box:
            mean, std = _box_centre_and_halfspan(pts)
        elif normalize_per_shape:
            mean = pts.mean(axis=1).reshape(S, 1, D)
            std = pts.std(axis=1).reshape(S, 1, D) if normalize_std_per_axis else pts.reshape(S, -1).std(axis=1).reshape(S, 1, 1)
        elif all_points_mean is not None and all_points_std is not None and not recenter_per_shape:
            mean, std = all_points_mean, all_points_std  # evaluation split normalised with the training statistics
        elif recenter_per_shape:
            mean, std = _box_centre_and_halfspan(pts)
        elif normalize_global:
            flat = pts.reshape(-1, D)
            mean = flat.mean(axis=0).reshape(1, 1, D)
            std = flat.std(axis=0).reshape(1, 1, D) if normalize_std_per_axis else pts.reshape(-1).std(axis=0).reshape(1, 1, 1)
        else:
            raise NotImplementedError('No Normalization')
        self.all_points_mean, self.all_points_std = mean, std
        self.all_points = (pts - mean) / std
        self.train_points = self.all_points[:, :min(_MAX_TRAIN_POINTS, self.all_points.shape[1])]
        self.tr_sample_size = min(_MAX_TRAIN_POINTS, tr_sample_size)
        self.te_sample_size = min(_MAX_TEST_POINTS, te_sample_size)
        self.display_axis_order = [0, 1, 2]

    def get_pc_stats(self, idx):
        if self.recenter_per_shape or self.normalize_per_shape or self.normalize_shape_box:
            return self.all_points_mean[idx].reshape(1, self.input_dim), self.all_points_std[idx].reshape(1, -1)
        return self.all_points_mean.reshape(1, -1), self.all_points_std.reshape(1, -1)

    def renormalize(self, mean, std):
        raw = self.all_points * self.all_points_std + self.all_points_mean
        self.all_points_mean, self.all_points_std = mean, std
        self.all_points = (raw - mean) / std
        self.train_points = self.all_points[:, :min(_MAX_TRAIN_POINTS, self.all_points.shape[1])]

    def __len__(self):
        return len(self.train_points)

    def __getitem__(self, idx):
        pool = self.train_points[idx]
        if not self.random_subsample:
            pick = np.arange(self.tr_sample_size)
        elif self.sample_with_replacement:
            pick = np.random.choice(pool.shape[0], self.tr_sample_size)
        else:
            pick = np.random.permutation(np.arange(pool.shape[0]))[:self.tr_sample_size]
        tr = torch.from_numpy(pool[pick, :]).float()
        mean, std = self.get_pc_stats(idx)
        sid, mid = self.all_cate_mids[idx]
        return {'idx': idx, 'select_idx': pick, 'tr_points': tr, 'input_pts': tr, 'mean': mean, 'std': std,
                'cate_idx': self.cate_idx_lst[idx], 'sid': sid, 'mid': mid,
                'display_axis_order': self.display_axis_order}


def init_np_seed(worker_id):
    np.random.seed(torch.initial_seed() % 4294967296)


def _dataset_kwargs(cfg):
    return dict(categories=cfg.cates, tr_sample_size=cfg.tr_max_sample_points, te_sample_size=cfg.te_max_sample_points,
                scale=cfg.dataset_scale, normalize_shape_box=cfg.normalize_shape_box,
                normalize_per_shape=cfg.normalize_per_shape, normalize_std_per_axis=cfg.normalize_std_per_axis,
                normalize_global=cfg.normalize_global, recenter_per_shape=cfg.recenter_per_shape,
                clip_forge_enable=cfg.clip_forge_enable, clip_model=cfg.clip_model,
                root_dir=getattr(cfg, "root_dir", None))


def get_datasets(cfg, args):
    """cfg = the ``data`` sub-config.  The evaluation split never sub-samples randomly and is normalised with the
    training set's statistics (:398-413)."""
    common = _dataset_kwargs(cfg)
    tr = ShapeNet15kPointClouds(split='train', random_subsample=cfg.random_subsample,
                                sample_with_replacement=cfg.sample_with_replacement, **common)
    te = ShapeNet15kPointClouds(split=getattr(args, "eval_split", "val"), all_points_mean=tr.all_points_mean,
                                all_points_std=tr.all_points_std, **common)
    return tr, te


def get_data_loaders(cfg, args):
    tr, te = get_datasets(cfg, args)
    order = {'sampler': data.distributed.DistributedSampler(tr, shuffle=True)} if args.distributed else {'shuffle': True}
    if args.eval_trainnll:
        order['shuffle'] = False
    train_loader = data.DataLoader(dataset=tr, batch_size=cfg.batch_size, num_workers=cfg.num_workers,
                                   drop_last=cfg.train_drop_last == 1, pin_memory=False,
                                   worker_init_fn=init_np_seed if cfg.num_workers else None, **order)
    test_loader = data.DataLoader(dataset=te, batch_size=cfg.batch_size_test, shuffle=False,
                                  num_workers=cfg.num_workers, pin_memory=False, drop_last=False)
    return {"test_loader": test_loader, "train_loader": train_loader}


class ResidentPointClouds:
    """The training pool of a ``ShapeNet15kPointClouds`` held in device memory; batches are cut on the device.

    ``epoch(e)`` yields dictionaries with the keys the trainers read (``tr_points`` [B,n,D] fp32, ``idx``,
    ``cate_idx``, ``mean``, ``std``).  Shape order: a permutation seeded by ``seed + e`` padded to a multiple of the
    world size and strided by rank -- exactly ``torch.utils.data.DistributedSampler`` (which is used to produce it),
    so the ranks see disjoint shards and a one-rank run sees every shape once; ``drop_last`` drops the ragged batch.
    """

    def __init__(self, dataset, device, batch_size, rank=0, world_size=1, seed=0, drop_last=True, shuffle=True):
        self.device = torch.device(device)
        self.pool = torch.as_tensor(np.ascontiguousarray(dataset.train_points), dtype=torch.float32).to(self.device)
        self.n = dataset.tr_sample_size
        self.random_subsample = bool(dataset.random_subsample)
        self.with_replacement = bool(dataset.sample_with_replacement)
        self.cate_idx = torch.as_tensor(dataset.cate_idx_lst, dtype=torch.long, device=self.device)
        per_shape = np.ndim(dataset.all_points_mean) == 3 and np.shape(dataset.all_points_mean)[0] == len(dataset) \
            and (dataset.recenter_per_shape or dataset.normalize_per_shape or dataset.normalize_shape_box)
        S, D = len(dataset), dataset.input_dim
        mean, std = np.asarray(dataset.all_points_mean), np.asarray(dataset.all_points_std)
        rows = S if per_shape else 1
        self.mean = torch.from_numpy(np.array(np.broadcast_to(mean, (rows, 1, D)))).to(self.device)
        self.std = torch.from_numpy(np.array(np.broadcast_to(std, (rows, 1, std.shape[-1])))).to(self.device)
        self.per_shape_stats = per_shape
        self.batch_size, self.drop_last = batch_size, drop_last
        self.rank, self.world_size, self.seed = rank, world_size, seed
        self._order = data.distributed.DistributedSampler(range(S), num_replicas=world_size, rank=rank,
                                                          shuffle=shuffle, seed=seed)
        self._gen = torch.Generator(device=self.device)

    def __len__(self):
        full, rest = divmod(len(self._order), self.batch_size)
        return full if self.drop_last or rest == 0 else full + 1

    def pick_points(self, count):
        """[count, n] indices into the pool's point axis"""
        P = self.pool.shape[1]
        if not self.random_subsample:
            return torch.arange(self.n, device=self.device).expand(count, self.n)
        if self.with_replacement:
            return torch.randint(P, (count, self.n), generator=self._gen, device=self.device)
        # without replacement: the n smallest of P iid keys per shape = a uniformly random n-subset in random order
        keys = torch.rand((count, P), generator=self._gen, device=self.device)
        return keys.topk(self.n, dim=1, largest=False).indices

    def batch(self, shape_idx):
        shape_idx = torch.as_tensor(shape_idx, dtype=torch.long, device=self.device)
        pick = self.pick_points(shape_idx.numel())
        pts = self.pool[shape_idx.unsqueeze(1), pick]  # [B, n, D]: one gather, no host round trip
        sel = shape_idx if self.per_shape_stats else torch.zeros_like(shape_idx)
        return {'idx': shape_idx, 'select_idx': pick, 'tr_points': pts, 'input_pts': pts,
                'mean': self.mean[sel], 'std': self.std[sel], 'cate_idx': self.cate_idx[shape_idx]}

    def epoch(self, epoch=0):
        self._order.set_epoch(epoch)
        self._gen.manual_seed((self.seed * 1000003 + epoch) * 4099 + self.rank)
        order = list(self._order)
        for b in range(len(self)):
            yield self.batch(order[b * self.batch_size:(b + 1) * self.batch_size])
