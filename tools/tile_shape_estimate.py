"""Empty-tile / empty-wave-block fractions of candidate tile shapes for the sparse split convolutions (CPU, numpy).
A tile (or a wave's sub-block) is skipped when no voxel within `margin` of it holds a point: margin 1 for the
conv that reads the voxelised grid, margin 2 for the delta mode of the second conv (models/pvcnn2_ada.py:206-233).
Clouds: the bench's Gaussian latents, the flat (airplane-like) cloud of tools/sparse_conv_bench.py and a clumped cloud
(95 % of the points at a quarter of the scale: what random-weight latents drift into).  Voxelisation as P1
(pvcnn2_ada.py:173-188): centre, / (2 max norm), + 0.5, * r, clamp, round."""
import sys
import numpy as np

rng = np.random.default_rng(0)
B = 32


def voxel_ids(co, r):
    co = co - co.mean(axis=2, keepdims=True)
    nrm = np.sqrt((co ** 2).sum(axis=1)).max(axis=1)[:, None, None]
    nc = np.clip((co / (2 * nrm) + 0.5) * r, 0, r - 1)
    return np.rint(nc).astype(np.int64)


def occupancy(co, r):
    v = voxel_ids(co, r)
    g = np.zeros((B, r, r, r), bool)
    for b in range(B):
        g[b, v[b, 0], v[b, 1], v[b, 2]] = True
    return g


def dilate(g, m):
    out = g.copy()
    for _ in range(m):
        p = np.pad(out, ((0, 0), (1, 1), (1, 1), (1, 1)))
        o = np.zeros_like(out)
        for dz in range(3):
            for dy in range(3):
                for dx in range(3):
                    o |= p[:, dz:dz + out.shape[1], dy:dy + out.shape[2], dx:dx + out.shape[3]]
        out = o
    return out


def empty_frac(gd, r, box):
    td, th, tw = box
    t = gd.reshape(B, r // td, td, r // th, th, r // tw, tw).any(axis=(2, 4, 6))
    return 1.0 - t.mean()


def clouds(n):
    g = rng.standard_normal((B, 3, n))
    flat = rng.standard_normal((B, 3, n)) * np.array([1.0, 0.15, 0.6])[None, :, None]
    cl = rng.standard_normal((B, 3, n))
    cl[:, :, : int(0.95 * n)] *= 0.25
    return (("gauss", g), ("flat", flat), ("clumped", cl))


# (tile, wave block) candidates; 256-voxel tiles = 4 waves x 64 voxels
CANDS = {
    32: [((2, 4, 32), (1, 2, 32)), ((4, 8, 8), (1, 8, 8)), ((4, 4, 16), (1, 4, 16)), ((8, 8, 4), (2, 8, 4)), ((4, 8, 8), (2, 4, 8)),
         ((4, 4, 8), (4, 4, 4)), ((4, 4, 4), (4, 4, 4))],
    16: [((4, 4, 16), (1, 4, 16)), ((4, 8, 8), (1, 8, 8)), ((8, 8, 4), (2, 8, 4)), ((4, 4, 8), (4, 4, 4)), ((4, 4, 4), (4, 4, 4))],
    8: [((4, 8, 8), (1, 8, 8)), ((4, 4, 4), (4, 4, 4)), ((2, 4, 4), (2, 4, 4))],
}
def chain_clouds(path, n):
    """x_t of the real chain (tools/dump_chain_clouds.py): [B, 2048, 3] per dumped step; the r = 16 / r = 8 grids see the
    FPS subsets (2048 -> 1024 -> 256) of the step's cloud, as the set-abstraction modules produce them"""
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    import oracle
    orc = oracle.lib()
    z = np.load(path)
    out = []
    for k in sorted(z.files):
        co = np.ascontiguousarray(z[k].transpose(0, 2, 1)).astype(np.float32)   # [B, 3, 2048]
        m = 2048
        while m > n:
            m //= 2 if m == 2048 else 4
            idx = orc.furthest_point_sampling(co, m)
            co = orc.gather_features_forward(co, idx)
        out.append((k, co.astype(np.float64)))
    return out


src = (lambda n: chain_clouds(sys.argv[2], n)) if len(sys.argv) > 2 and sys.argv[1] == "--clouds" else clouds
for r, n in ((32, 2048), (16, 1024), (8, 256)):
    for name, co in src(n):
        g = occupancy(co, r)
        d1, d2 = dilate(g, 1), dilate(g, 2)
        print(f"r={r} {name:8s} occupied voxels {g.mean():.3f}  dilated m1 {d1.mean():.3f} m2 {d2.mean():.3f}")
        for tile, wb in CANDS[r]:
            print(f"    tile {tile!s:12s} empty m1 {empty_frac(d1, r, tile):.2f} m2 {empty_frac(d2, r, tile):.2f}"
                  f"   wave block {wb!s:11s} empty m1 {empty_frac(d1, r, wb):.2f} m2 {empty_frac(d2, r, wb):.2f}")


def block_histogram(gd, r, tile):
    """voxel compaction inside a tile: active voxels packed into 32-voxel column blocks, block j -> wave j % 4.  Returns
    (fraction of tiles with 0 blocks, histogram of blocks per non-empty tile, MFMA work relative to dense = blocks / 8,
    'rounds' relative to dense = per-wave maximum of blocks / 2)"""
    td, th, tw = tile
    t = gd.reshape(B, r // td, td, r // th, th, r // tw, tw).sum(axis=(2, 4, 6)).reshape(-1)
    nb = (t + 31) // 32
    hist = np.bincount(nb, minlength=9)[:9] / nb.size
    rounds = (nb + 3) // 4
    return hist, nb.mean() / 8.0, rounds.mean() / 2.0


if len(sys.argv) > 1 and sys.argv[-1] == "--blocks":
    print("\nvoxel compaction (active voxels of a tile packed into 32-voxel column blocks):")
    for r, n, tile in ((32, 2048, (2, 4, 32)), (32, 2048, (4, 8, 8)), (16, 1024, (4, 4, 16)), (16, 1024, (4, 8, 8))):
        for name, co in src(n):
            if name.startswith("step_") and name not in ("step_0000", "step_0005", "step_0020", "step_0400"):
                continue
            g = occupancy(co, r)
            for m in (1, 2):
                h, work, rounds = block_histogram(dilate(g, m), r, tile)
                print(f"r={r} tile {tile!s:11s} {name:10s} m{m}: blocks/tile hist {np.array2string(h, precision=2, floatmode='fixed')}"
                      f"  MFMA work {work:.3f} of dense, rounds {rounds:.3f}")
