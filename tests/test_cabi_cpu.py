"""CPU suite: the C-ABI library loads here (no GPU) and exports every symbol include/lion_hip.h
declares; the product fails loudly (no CPU fallback) when handed CPU tensors."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "lion_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lion_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from lion_amd import _lib
    assert os.path.exists(_lib.SO_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.SO_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/lion_hip.h but not exported"
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.load().lion_abi_version() == 1


def test_workspace_queries_are_pure_host_calls():
    from lion_amd import _lib
    l = _lib.load()
    assert l.lion_avg_voxelize_workspace_bytes(32, 64, 2048, 32) >= 32 * 2048 * 4
    assert l.lion_avg_voxelize_workspace_bytes(0, 1, 1, 1) == 0
    assert l.lion_emd_workspace_bytes(2, 2048, 2048) >= 2 * 4 * 2048 * 4


def test_no_cpu_fallback():
    from lion_amd.functional.backend import _backend
    import lion_amd.functional as F
    with pytest.raises(RuntimeError):
        _backend.avg_voxelize_forward(torch.zeros(1, 2, 8), torch.zeros(1, 3, 8, dtype=torch.int32), 4)
    with pytest.raises(RuntimeError):
        F.ball_query(torch.zeros(1, 3, 4), torch.zeros(1, 3, 8), 0.1, 4)
    with pytest.raises(RuntimeError):
        F.furthest_point_sample(torch.zeros(1, 3, 8), 4)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: no file under lion_amd/ may mention it."""
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "lion_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, fn)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "liboracle" in src:
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_weight_cache_never_serves_a_freed_tensors_copy():
    """lion_amd._wcache.WeightCache: entries are keyed by storage address, validated by the version counter and hold a
    strong reference to their source tensor -- a new tensor can therefore never inherit the derived copy of a freed
    one that happened to live at the same address (the failure mode of an id()-keyed cache)."""
    import gc
    import torch
    from lion_amd._wcache import WeightCache
    builds = []

    def build(w):
        builds.append(w.data_ptr())
        return w.detach().clone() * 2

    cache = WeightCache(build, capacity=4)
    a = torch.arange(6.0).reshape(2, 3)
    assert torch.equal(cache.get(a), a * 2) and len(builds) == 1
    assert cache.get(a) is cache.get(a) and len(builds) == 1          # hit
    a.add_(1.0)                                                         # in-place update bumps the version
    assert torch.equal(cache.get(a), a * 2) and len(builds) == 2
    ptr = a.data_ptr()
    del a
    gc.collect()
    b = torch.full((2, 3), 7.0)                                         # may or may not reuse a's block ...
    assert torch.equal(cache.get(b), b * 2)                             # ... the result is right either way
    assert b.data_ptr() != ptr or len(builds) >= 3                      # and a reused address would have rebuilt
    for i in range(8):                                                  # capacity bound (LRU)
        cache.get(torch.zeros(1 + i))
    assert len(cache._entries) <= 4


def test_host_side_plan_functions():
    """size / plan helpers of the C ABI are pure host code: pin the values the Python shim relies on."""
    from lion_amd import _lib
    lib = _lib.load()
    # Conv3d tile plans: dense 4x2 MFMA tiles where >= 512 workgroups remain, 2x2 tiles for sparse launches
    assert lib.lion_conv3d_stat_tiles(32, 64, 32, 0) == 32768 // 512
    assert lib.lion_conv3d_stat_tiles(32, 64, 32, 1) == 32768 // 256
    assert lib.lion_conv3d_stat_tiles(16, 64, 32, 0) == 4096 // 256      # 8 x 1 x 32 = 256 < 512 -> 2x2 tiles
    assert lib.lion_conv3d_stat_tiles(16, 128, 32, 0) == 4096 // 512
    assert lib.lion_conv3d_stat_tiles(8, 128, 32, 0) == 512 // 128        # r = 8: one MFMA column block per wave
    assert lib.lion_conv3d_stat_tiles(7, 64, 32, 0) == 0
    assert lib.lion_conv3d_occupancy_ints(32, 64, 32) == 2 * 32 * 128 + 4    # flags, list, [queue, exit counter, mode, pad]
    assert lib.lion_conv3d_occupancy_ints(8, 64, 32) == 0                  # never sparse at r = 8
    assert lib.lion_conv3d_packed_floats(64, 3) == 4 * 27 * 64             # Cin padded to 4
    assert lib.lion_conv3d_wgrad_workspace_floats(32, 64, 64, 32) == 32 * 1 * 64 * 64 * 27 + 64   # one partial per workgroup (round 3) + the split kernel's maxima / scales (round 4)
    assert lib.lion_conv3d_wgrad_workspace_floats(32, 32, 32, 32) == 32 * 4 * 32 * 32 * 27 + 64  # 4 spatial splits
    assert lib.lion_conv3d_wgrad_workspace_floats(32, 3, 32, 32) == 0      # Cin % 4 != 0: library fallback
    # devoxelize plans (r = 32, N <= 2048): per cloud a 16-byte header, 2304 piece ids, 8 corner offsets and a status per point
    assert lib.lion_devoxelize_plan_bytes(32, 2048, 32) == 32 * (16 + 2304 * 2 + 2048 * 16 + 2048)
    assert lib.lion_devoxelize_plan_bytes(3, 999, 32) == 3 * (16 + 2304 * 2 + 2048 * 16 + 2048)
    assert lib.lion_devoxelize_plan_bytes(32, 2048, 16) == 0 and lib.lion_devoxelize_plan_bytes(32, 4096, 32) == 0
    # pointwise conv: column tiles of 4 waves x VB x 32 columns
    assert lib.lion_pwconv_stat_tiles(64, 35, 32768) == 32768 // 512
    assert lib.lion_pwconv_stat_tiles(128, 64, 8000) == -(-8000 // 256)
    assert lib.lion_pwconv_stat_tiles(48, 16, 10000) == -(-10000 // 512)   # 48 -> two 32-row tiles, one half masked
    assert lib.lion_pwconv_stat_tiles(256, 320, 8192) == 8192 // 512        # 256 / 128 rows x 320 k do not fit LDS: 64-row tiles
    assert lib.lion_pwconv_stat_tiles(4, 128, 2048) == 2048 // 32           # <= 4096 columns: the K-split kernel, 32-column tiles
    assert lib.lion_pwconv_stat_tiles(128, 256, 64) == 2
    # split-operand 1x1 kernel: 4 waves x VB x 32 columns per workgroup, VB = 1 / 2 / 2 for channel tiles of 128 / 64 / 32
    assert lib.lion_pwconv_split_stat_tiles(128, 192, 2048) == 2048 // 128
    assert lib.lion_pwconv_split_stat_tiles(64, 35, 32768) == 32768 // 256
    assert lib.lion_pwconv_split_stat_tiles(96, 16, 1000) == -(-1000 // 256)
    assert lib.lion_pwconv_split_packed_halfs(128, 35) == 3 * 4 * 128 * 8 + 8      # ceil16(35) = 3 chunks + the scale tail
    assert lib.lion_conv3d_split_stat_tiles(8, 128) == 2 and lib.lion_conv3d_split_stat_tiles(32, 64) == 128
    assert lib.lion_pwconv_packed_floats(64, 35) == 36 * 64 and lib.lion_pwconv_packed_floats(4, 128) == 128 * 64
    # skinny GEMM: split K until ~256 workgroups, >= 8 k-steps per wave
    assert lib.lion_skinny_splits(2048, 2048) == 4
    assert lib.lion_skinny_splits(2048, 256) == 8                        # 8 output tiles only: split K further
    assert lib.lion_skinny_splits(256, 2048) == 1
    assert lib.lion_skinny_splits(128, 48) == 0
    assert lib.lion_skinny_packed_floats(2048, 2048) == 2048 * 8 * 256
