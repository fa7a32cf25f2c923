"""Inference-time fusions of PVConv's voxel branch (SURVEY.md 8a P2-P4 folded into C3 / K4):
conv -> AdaGN -> Swish -> conv -> AdaGN -> SE3d -> devoxelize with no stand-alone pass over the grid.
Thin wrappers over lion_conv3d_k3_fused_forward / lion_groupnorm_fold /
lion_trilinear_devoxelize_affine_forward (include/lion_hip.h)."""
import torch

from . import _lib
from ._wcache import WeightCache
from . import conv_ops
from .conv_ops import packed_weight, supported


def _border_sums(weight):
    m = torch.tensor([[0., 1., 1.], [1., 1., 1.], [1., 1., 0.]], dtype=torch.float64, device=weight.device)
    ws = torch.einsum("oidhw,ad,bh,cw->abcio", weight.detach().double(), m, m, m)
    return ws.reshape(27, weight.shape[1], weight.shape[0]).float().contiguous()


_WSUM_CACHE = WeightCache(_border_sums)


def border_weight_sums(weight):
    """[Cout,Cin,3,3,3] -> [27,Cin,Cout]: the weights summed over the taps that stay inside the grid, for each of
    the 27 border configurations (per axis: 0 = voxel on the low face, 1 = interior, 2 = on the high face); fp64
    sums, cached per (storage, version)."""
    return _WSUM_CACHE.get(weight)


def conv3d_occupancy(counts, r, cout, b, consumer_aware=False):
    """(occ for the conv on the voxelised grid, occ for the delta mode of the following conv), one launch.  A sparse
    conv3d_fused call pops its work from the buffer's queue and re-arms it when its last workgroup leaves, so the same
    buffer serves any number of convolutions one after the other on a stream (never two at once).
    counts: int32 [B, r^3] from the voxelisation.
    consumer_aware = 1 (only for the conv -> delta conv -> devoxelize chain of a PVConv, models/pvcnn2_ada.py): the
    convolutions leave the output of empty tiles WITHOUT A READER unwritten (the first conv's output is read where the
    second one stages its halos, the second one's around the points); their GroupNorm sums stay complete.
    consumer_aware = 2: the delta convolution is the split kernel, which stages zeros for the rows of the first
    convolution's empty tiles without loading them -- the first convolution stores its occupied tiles only."""
    lib = _lib.load()
    n = lib.lion_conv3d_occupancy_ints(r, cout, b)
    buf = torch.empty((2, n), device=counts.device, dtype=torch.int32)
    cnt_c = counts.contiguous()
    _lib.check(lib.lion_conv3d_tile_occupancy_aware(_lib.ptr(cnt_c), b, r, cout, _lib.ptr(buf[0]), _lib.ptr(buf[1]),
                                                    int(consumer_aware), _lib.stream_ptr(counts.device)),
               "conv3d_tile_occupancy_aware")
    return buf[0], buf[1]


class FoldSpec:
    """what groupnorm_fold / groupnorm_fold_se need besides the tile sums: the AdaGN's GroupNorm, the style factor / bias
    views, the element count per channel and (optionally) the SE3d module -- handed to a producer (conv3d_fused), which
    returns (A, Bs) instead of the sums.  (Round 6 built the fold into the producing kernel's tail -- the workgroup finishing
    a sample's last tile folds it, agent-scope stores + a relaxed arrival counter, no L2 write-back -- bit-identical, 14-28
    launches per step fewer and NOT faster: the tail costs the convolution what the launch cost the step.  Kept out of the
    product: tools/exp/fold_in_tail/, profiles/r06_fold_in_tail_ab.txt.)"""

    def __init__(self, gn: torch.nn.GroupNorm, fac, gbias, count, se=None):
        self.gn, self.fac, self.gbias, self.count, self.se = gn, fac, gbias, int(count), se

    def apply(self, stats):
        """the separate launch: (A, Bs) from tile sums"""
        if self.se is not None:
            merged = groupnorm_fold_se(stats, self.gn, self.fac, self.gbias, self.count, self.se)
            if merged is not None:
                return merged
            a, b_, m = groupnorm_fold(stats, self.gn, self.fac, self.gbias, self.count)
            return se_gate_(a, b_, m, self.se)
        a, b_, _ = groupnorm_fold(stats, self.gn, self.fac, self.gbias, self.count)
        return a, b_


def conv3d_fused(x, conv, pro=None, want_stats=True, occ=None, prev_conv=None, split=None, fold=None):
    """x [B,Cin,r,r,r] -> (y [B,Cout,r,r,r], stats [B,Cout,T,2] | None).  pro = (A, Bs) applies
    swish(x*A+Bs) to the input on the fly.
    occ (from conv3d_occupancy: the first element for the conv on the voxelised grid, the second for the conv
    after it) enables the sparse evaluation:
      * pro is None (the conv that reads the voxelised grid): tiles whose halo holds no point skip their K loop
        (output = bias exactly), occupied tiles are balanced over the CUs through a work queue;
      * pro given and prev_conv = the convolution that produced x: x is bias1 wherever no point is near, so the
        activated input is a per-channel constant + a sparse delta; the constant's response is added in the
        epilogue (27 border configurations), the MFMA loop runs on the delta and skips tiles with no point within
        2 voxels.
    split (None = conv_ops.SPLIT): run on the split-operand kernel (fp16 x 2 pieces on the 16-bit MFMA pipe, fp32
    accurate, csrc/conv3d_split.hip) where Cin % 16 == 0; same modes, same results within fp32 rounding.
    fold (a FoldSpec): return (y, (A, Bs)) -- the GroupNorm fold of y (+ SE gate) -- instead of (y, stats)."""
    lib = _lib.load()
    b, cin, r = x.shape[0], x.shape[1], x.shape[2]
    cout = conv.out_channels
    use_split = conv_ops.use_split(split, cin, cout, r)
    if cin % 4:
        assert pro is None
        x = torch.cat([x, x.new_zeros(b, 4 - cin % 4, r, r, r)], dim=1)
        cin = x.shape[1]
    x = x.contiguous()
    wp = conv_ops.split_packed_weight(conv.weight) if use_split else packed_weight(conv.weight)
    y = torch.empty((b, cout, r, r, r), device=x.device, dtype=torch.float32)
    st = _lib.stream_ptr(x.device)
    pa = pb = pbias = tconst = None
    if pro is not None:
        pa, pb = pro[0].contiguous(), pro[1].contiguous()
    bias = conv.bias.detach().contiguous() if conv.bias is not None else None
    sparse = occ is not None and r >= 16
    if sparse and pro is not None:
        if prev_conv is None or cin > 256:
            sparse = False
        else:
            ws = border_weight_sums(conv.weight)
            pbias = prev_conv.bias.detach().contiguous() if prev_conv.bias is not None else None
            tconst = torch.empty((b, 27, cout), device=x.device, dtype=torch.float32)
            _lib.check(lib.lion_conv3d_const_response(_lib.ptr(ws), _lib.ptr(bias), _lib.ptr(pbias), _lib.ptr(pa),
                                                      _lib.ptr(pb), b, cin, cout, _lib.ptr(tconst), st),
                       "conv3d_const_response")
    stats = None
    if want_stats:
        tiles = lib.lion_conv3d_split_stat_tiles(r, cout) if use_split else lib.lion_conv3d_stat_tiles(r, cout, b, int(sparse))
        stats = torch.empty((b, cout, tiles, 2), device=x.device, dtype=torch.float32)
    if not sparse:
        occ = None
    fwd = lib.lion_conv3d_k3_split_forward if use_split else lib.lion_conv3d_k3_fused_forward
    _lib.check(fwd(
        _lib.ptr(x), _lib.ptr(wp), _lib.ptr(bias),
        b, cin, cout, r, _lib.ptr(pa), _lib.ptr(pb), _lib.ptr(pbias), _lib.ptr(tconst), _lib.ptr(y), _lib.ptr(stats),
        _lib.ptr(occ), st), "conv3d_k3_split_forward" if use_split else "conv3d_k3_fused_forward")
    if fold is not None:
        return y, fold.apply(stats)
    return y, stats


def groupnorm_fold(stats, gn: torch.nn.GroupNorm, fac, gbias, voxels):
    """per-tile channel sums -> (A, Bs, chmean) with AdaGN(x) == x*A + Bs per (batch, channel)."""
    b, c, t, _ = stats.shape
    A = torch.empty((b, c), device=stats.device, dtype=torch.float32)
    Bs = torch.empty_like(A)
    cm = torch.empty_like(A)
    # fac / gbias are normally the two halves of the [B, 2C] style projection: consume them in place
    # (row stride ld) instead of materialising two contiguous copies per fold.  Every temporary is
    # kept alive in a local until the launch is enqueued: a temporary created inside the argument
    # list is freed before the next argument is evaluated and the caching allocator hands the same
    # block to the next allocation (fac would silently become gbias).
    if not (fac.stride(1) == 1 and gbias.stride(1) == 1 and fac.stride(0) == gbias.stride(0) and fac.stride(0) >= c):
        fac, gbias = fac.contiguous(), gbias.contiguous()
    fac_c, gb_c, ld = fac, gbias, int(fac.stride(0))
    _lib.check(_lib.load().lion_groupnorm_fold(
        _lib.ptr(stats), b, c, t, gn.num_groups, int(voxels), _lib.ptr(gn.weight.detach()),
        _lib.ptr(gn.bias.detach()), _lib.ptr(fac_c), _lib.ptr(gb_c), ld, float(gn.eps),
        _lib.ptr(A), _lib.ptr(Bs), _lib.ptr(cm), _lib.stream_ptr(stats.device)), "groupnorm_fold")
    return A, Bs, cm


def groupnorm_fold_se(stats, gn: torch.nn.GroupNorm, fac, gbias, voxels, se):
    """groupnorm_fold followed by se_gate_ in one launch; None when the shape is outside the merged kernel's range."""
    b, c, t, _ = stats.shape
    w1, w2 = se.fc[0].weight.detach(), se.fc[2].weight.detach()
    h = w1.shape[0]
    if c > 256 or c < 4 or h > 128 or not (w1.is_contiguous() and w2.is_contiguous()):
        return None
    A = torch.empty((b, c), device=stats.device, dtype=torch.float32)
    Bs = torch.empty_like(A)
    if not (fac.stride(1) == 1 and gbias.stride(1) == 1 and fac.stride(0) == gbias.stride(0) and fac.stride(0) >= c):
        fac, gbias = fac.contiguous(), gbias.contiguous()
    fac_c, gb_c, ld = fac, gbias, int(fac.stride(0))   # kept alive until the launch is enqueued (see groupnorm_fold)
    _lib.check(_lib.load().lion_groupnorm_fold_se(
        _lib.ptr(stats), b, c, t, gn.num_groups, int(voxels), _lib.ptr(gn.weight.detach()), _lib.ptr(gn.bias.detach()),
        _lib.ptr(fac_c), _lib.ptr(gb_c), ld, float(gn.eps), _lib.ptr(w1), _lib.ptr(w2), h, _lib.ptr(A), _lib.ptr(Bs),
        _lib.stream_ptr(stats.device)), "groupnorm_fold_se")
    return A, Bs


def se_gate_(A, Bs, chmean, se):
    """in place: (A, Bs) *= SE3d gate computed from the folded scalars (mean of AdaGN(y) = A*mean(y)+Bs)."""
    w1, w2 = se.fc[0].weight.detach(), se.fc[2].weight.detach()
    h, c = w1.shape
    if c > 1024 or h > 128 or not (w1.is_contiguous() and w2.is_contiguous()):
        gate = se.fc(A * chmean + Bs)
        return A * gate, Bs * gate
    _lib.check(_lib.load().lion_se_gate(_lib.ptr(chmean), _lib.ptr(w1), _lib.ptr(w2), A.shape[0], c, h,
                                        _lib.ptr(A), _lib.ptr(Bs), _lib.stream_ptr(A.device)), "se_gate")
    return A, Bs


def devoxelize_plan(coords, r):
    """what the voxel coordinates [B,3,N] alone decide about a devoxelisation at resolution r (needed row pieces, LDS
    slots, corner offsets per point): computed once per (cloud, r), used by every devoxelize_affine(..., plan=) at these
    coordinates.  None when the shape is outside the planned kernel's range (r = 32, N <= 2048)."""
    lib = _lib.load()
    b, _, n = coords.shape
    nbytes = lib.lion_devoxelize_plan_bytes(b, n, int(r))
    if nbytes == 0 or not coords.is_contiguous() or coords.dtype != torch.float32:
        return None
    buf = torch.empty((nbytes,), device=coords.device, dtype=torch.uint8)
    _lib.check(lib.lion_trilinear_devoxelize_plan(_lib.ptr(coords), b, n, int(r), _lib.ptr(buf), nbytes,
                                                  _lib.stream_ptr(coords.device)), "trilinear_devoxelize_plan")
    return {"buf": buf, "coords": coords, "shape": (b, n, int(r))}


def devoxelize_affine(grid, coords, r, scale, shift, plan=None):
    """trilinear_devoxelize(scale[b,c]*grid + shift[b,c]) without materialising the scaled grid.  plan (devoxelize_plan
    of these coordinates): the two-step form, bit-identical."""
    b, c = grid.shape[:2]
    n = coords.shape[2]
    out = torch.empty((b, c, n), device=grid.device, dtype=torch.float32)
    co_c, gr_c, sc_c, sh_c = coords.contiguous(), grid.contiguous(), scale.contiguous(), shift.contiguous()
    if plan is not None and plan["shape"] == (b, n, int(r)) and plan["coords"].data_ptr() == co_c.data_ptr():
        buf = plan["buf"]
        _lib.check(_lib.load().lion_trilinear_devoxelize_planned_forward(
            _lib.ptr(buf), buf.numel(), _lib.ptr(co_c), _lib.ptr(gr_c), _lib.ptr(sc_c), _lib.ptr(sh_c), b, c, n, int(r),
            _lib.ptr(out), _lib.stream_ptr(grid.device)), "trilinear_devoxelize_planned_forward")
        return out
    _lib.check(_lib.load().lion_trilinear_devoxelize_affine_forward(
        _lib.ptr(co_c), _lib.ptr(gr_c), _lib.ptr(sc_c), _lib.ptr(sh_c), b, c, n, int(r), _lib.ptr(out),
        _lib.stream_ptr(grid.device)),
        "trilinear_devoxelize_affine_forward")
    return out


def fusable(conv1, conv2, r, x):
    return (x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()
            and supported(conv1.in_channels, conv1.out_channels, r)
            and supported(conv2.in_channels, conv2.out_channels, r)
            and conv2.in_channels % 4 == 0 and conv2.in_channels <= 256)


def _pw_pack(weight):
    cout, cin = weight.shape[:2]
    lib = _lib.load()
    wp = torch.empty((lib.lion_pwconv_packed_floats(cout, cin),), device=weight.device, dtype=torch.float32)
    w_c = weight.detach().reshape(cout, cin).contiguous()
    _lib.check(lib.lion_pwconv_pack_weights(_lib.ptr(w_c), cout, cin, _lib.ptr(wp),
                                            _lib.stream_ptr(weight.device)), "pwconv_pack_weights")
    return wp


_PW_CACHE = WeightCache(_pw_pack)
PW_ALL = __import__("os").environ.get("LION_PW_ALL", "1") != "0"


def pw_packed_weight(weight):
    """[Cout,Cin,1(,1)] -> k-major [ceil2(Cin),Cout]; cached per (storage, version)."""
    return _PW_CACHE.get(weight)


def _pw_split_pack(weight):
    cout, cin = weight.shape[:2]
    lib = _lib.load()
    wp = torch.empty((lib.lion_pwconv_split_packed_halfs(cout, cin),), device=weight.device, dtype=torch.int16)
    w_c = weight.detach().reshape(cout, cin).contiguous()
    _lib.check(lib.lion_pwconv_split_pack_weights(_lib.ptr(w_c), cout, cin, _lib.ptr(wp),
                                                  _lib.stream_ptr(weight.device)), "pwconv_split_pack_weights")
    return wp


_PW_SPLIT_CACHE = WeightCache(_pw_split_pack)
# Which arithmetic the 1x1 convolutions run in (both fp32-accurate): fp16 x 2 split operands on the 16-bit MFMA pipe
# (csrc/pwconv_split.hip) where it is the faster kernel, the fp32 MFMA kernels elsewhere and everywhere with
# LION_PW_SPLIT=0.  tools/pw_bench.py, B = 32 (graph replay, us fp32 -> split): 192->128 L=2048 78 -> 28, 128->128 (+AdaGN
# prologue) 106 -> 26, 128->256 196 -> 46, 320->256 L=512 63 -> 27; short activations (B x L < 8192 columns: a handful
# of workgroups, latency bound) and the long thin ones (L >= 8192 with Cin x Cout < 8192: the fp32 kernel already moves
# 3.1-3.8 TB/s) stay on the fp32 kernels.
PW_SPLIT = __import__("os").environ.get("LION_PW_SPLIT", "1") != "0"


def pw_use_split(split, b, cin, cout, L):
    """split: None = module policy, True = wherever the kernel can run, False = never."""
    can = cin * L < (1 << 29) and cin <= 4096
    if split is None:
        return PW_SPLIT and can and b * L >= 8192 and (L <= 4096 or cin * cout >= 8192)
    return bool(split) and can


def pw_supported(conv, x):
    """every kernel-size-1 Conv1d / Conv2d whose weight tile fits LDS (all of the released models'): large
    activations (set-abstraction MLPs) for the bytes, short ones (L = 16..256 points, classifier, attention
    projections, time embedding) so that no library GEMM / layout-transposing convolution is left in the step."""
    if not PW_ALL:  # A/B switch (LION_PW_ALL=0): only the large activations, the rest on the library GEMM
        L = x[0, 0].numel()
        if not (conv.out_channels in (32, 64, 128, 256) and L >= 1024 and x.shape[0] * L * conv.out_channels >= (1 << 22)):
            return False
    return (conv.groups == 1 and all(k == 1 for k in conv.kernel_size) and all(s == 1 for s in conv.stride)
            and all(p == 0 for p in conv.padding) and x.is_cuda and x.dtype == torch.float32 and x[0, 0].numel() > 0
            and _lib.load().lion_pwconv_stat_tiles(conv.out_channels, conv.in_channels, x[0, 0].numel()) > 0)


def pwconv_fused(x, conv, pro=None, want_stats=True, split=None):
    """1x1 conv of a [B,Cin,*] activation on the MFMA kernel: y [B,Cout,*] and its GroupNorm tile sums
    [B,Cout,T,2]; pro = (A, Bs) applies swish(x*A+Bs) (the previous layer's AdaGN + Swish) in flight.
    split: see pw_use_split (None = the split-operand kernel for long activations, the fp32 MFMA kernels for short)."""
    lib = _lib.load()
    x = x.contiguous()
    b, cin = x.shape[:2]
    L = x[0, 0].numel()
    cout = conv.out_channels
    use_split = pw_use_split(split, b, cin, cout, L)
    wp = _PW_SPLIT_CACHE.get(conv.weight) if use_split else pw_packed_weight(conv.weight)
    y = torch.empty((b, cout) + tuple(x.shape[2:]), device=x.device, dtype=torch.float32)
    tiles = (lib.lion_pwconv_split_stat_tiles if use_split else lib.lion_pwconv_stat_tiles)(cout, cin, L)
    stats = torch.empty((b, cout, tiles, 2), device=x.device, dtype=torch.float32) if want_stats else None
    pa = pb = None
    if pro is not None:
        pa, pb = pro[0].contiguous(), pro[1].contiguous()
    bias = conv.bias.detach().contiguous() if conv.bias is not None else None
    fwd = lib.lion_pwconv_split_forward if use_split else lib.lion_pwconv_forward
    _lib.check(fwd(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(bias), b, cin, cout, L,
                   _lib.ptr(pa), _lib.ptr(pb), _lib.ptr(y), _lib.ptr(stats),
                   _lib.stream_ptr(x.device)), "pwconv_split_forward" if use_split else "pwconv_forward")
    return y, stats


# The last layer of a set-abstraction MLP evaluated twice (sums, then activated maximum) instead of stored (round 4) --
# built, bit-identical (tests/test_fold_se_gpu.py), and NOT faster: SA-0 layer 2 (32 -> 64 over 1 M columns) takes 100 us per
# pass without storing a byte against 121 us with its 268 MB output + 42 us for the max pass (208 vs 163 us; the sampling
# step is unchanged within noise).  The layer is not bound by its bytes: one wave = 64 loads, one wait, 128 MFMAs, with two
# workgroups per CU to overlap those phases.  Off by default; LION_PW_MAX_RECOMPUTE=1 selects it (saves the 268 MB tensor).
MAX_RECOMPUTE = __import__("os").environ.get("LION_PW_MAX_RECOMPUTE", "0") != "0"


def pwconv_max_recompute(x, conv, gn, style, pro):
    """max over the 32 neighbours of swish(AdaGN(conv(act(x)))) for x [B, Cin, M, 32] -> [B, Cout, M] without storing the
    conv's output: pass 1 = GroupNorm sums only, fold, pass 2 = the same GEMM with this layer's AdaGN + Swish and the max in
    the epilogue (lion_pwconv_forward_max).  None when the shape is outside that kernel's range."""
    lib = _lib.load()
    x = x.contiguous()
    b, cin, m, u = x.shape
    L = m * u
    cout = conv.out_channels
    wp = pw_packed_weight(conv.weight)
    tiles = lib.lion_pwconv_stat_tiles(cout, cin, L)
    if tiles <= 0:
        return None
    stats = torch.empty((b, cout, tiles, 2), device=x.device, dtype=torch.float32)
    pa = pb = None
    if pro is not None:
        pa, pb = pro[0].contiguous(), pro[1].contiguous()
    bias = conv.bias.detach().contiguous() if conv.bias is not None else None
    st = _lib.stream_ptr(x.device)
    rc = lib.lion_pwconv_forward_max(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(bias), b, cin, cout, L, _lib.ptr(pa), _lib.ptr(pb),
                                     None, None, _lib.ptr(stats), None, st)
    if rc == -2:   # LION_EUNSUPPORTED: the caller takes the stored-output path
        return None
    _lib.check(rc, "pwconv_forward_max (sums)")
    f, g = gn.affine(style)
    A, Bs, _ = groupnorm_fold(stats, gn.norm, f, g, L)
    y = torch.empty((b, cout, m), device=x.device, dtype=torch.float32)
    _lib.check(lib.lion_pwconv_forward_max(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(bias), b, cin, cout, L, _lib.ptr(pa),
                                           _lib.ptr(pb), _lib.ptr(A), _lib.ptr(Bs), None, _lib.ptr(y), st),
               "pwconv_forward_max (max)")
    return y


def pwconv_raw(x, w2d, bias=None, cached_param=None, transposed=False):
    """y[b] = w2d @ x[b] (+ bias) on the library's 1x1-convolution kernels for a plain [Cout, Cin] matrix; None when
    the shape is not supported.  cached_param: the nn.Parameter w2d is a view of (its packed form is cached per
    version).  transposed: y[b] = w2d^T @ x[b] -- the data gradient of a layer with weight w2d; on the split kernel the
    packed W^T comes straight from W and the scale of W's cached packed form (lion_pwconv_split_pack_weights_t)."""
    lib = _lib.load()
    x = x.contiguous()
    b, cin = x.shape[:2]
    cout = w2d.shape[1] if transposed else w2d.shape[0]
    L = x[0, 0].numel()
    if not (x.is_cuda and x.dtype == torch.float32 and L > 0 and lib.lion_pwconv_stat_tiles(cout, cin, L) > 0):
        return None
    use_split = pw_use_split(None, b, cin, cout, L)
    if transposed and use_split and cached_param is not None:
        wf = _PW_SPLIT_CACHE.get(cached_param)
        wp = torch.empty((lib.lion_pwconv_split_packed_halfs(cout, cin),), device=x.device, dtype=torch.int16)
        w_c = w2d.detach().contiguous()
        _lib.check(lib.lion_pwconv_split_pack_weights_t(_lib.ptr(w_c), cin, cout, _lib.ptr(wf), _lib.ptr(wp),
                                                        _lib.stream_ptr(x.device)), "pwconv_split_pack_weights_t")
    elif transposed:
        wp = (_pw_split_pack if use_split else _pw_pack)(w2d.detach().t().contiguous())
    elif cached_param is not None:
        wp = _PW_SPLIT_CACHE.get(cached_param) if use_split else _PW_CACHE.get(cached_param)
    else:
        wp = (_pw_split_pack if use_split else _pw_pack)(w2d)
    y = torch.empty((b, cout) + tuple(x.shape[2:]), device=x.device, dtype=torch.float32)
    bias_c = bias.detach().contiguous() if bias is not None else None
    fwd = lib.lion_pwconv_split_forward if use_split else lib.lion_pwconv_forward
    _lib.check(fwd(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(bias_c), b, cin, cout, L, None, None, _lib.ptr(y), None,
                   _lib.stream_ptr(x.device)), "pwconv_split_forward" if use_split else "pwconv_forward")
    return y


def pwconv_wgrad(x, gy, want_bias=False):
    """gw [Cout, Cin] = sum_b gy[b] @ x[b]^T for x [B, Cin, *], gy [B, Cout, *] (csrc/pwconv_wgrad.hip); want_bias: returns
    (gw, gb) with gb [Cout] = gy summed over batch and positions, from the same launch"""
    lib = _lib.load()
    x, gy = x.contiguous(), gy.contiguous()
    b, cin = x.shape[:2]
    cout = gy.shape[1]
    L = x[0, 0].numel()
    wsb = lib.lion_pwconv_wgrad_workspace_bytes(b, cin, cout, L)
    ws = torch.empty((wsb,), device=x.device, dtype=torch.uint8)
    gw = torch.empty((cout, cin), device=x.device, dtype=torch.float32)
    gb = torch.empty((cout,), device=x.device, dtype=torch.float32) if want_bias else None
    _lib.check(lib.lion_pwconv_wgrad(_lib.ptr(x), _lib.ptr(gy), b, cin, cout, L, _lib.ptr(ws), wsb, _lib.ptr(gw), _lib.ptr(gb),
                                     _lib.stream_ptr(x.device)), "pwconv_wgrad")
    return (gw, gb) if want_bias else gw


def group_points(coords, centers, feat, idx):
    """[B, 3 + C, M, U]: neighbour coordinates relative to their centre and (feat not None) the gathered features, in
    one tensor (BallQuery.forward, pvcnn2_ada.py:98-114) -- no subtraction pass, no torch.cat."""
    b, _, n = coords.shape
    m, u = idx.shape[1], idx.shape[2]
    c = 0 if feat is None else feat.shape[1]
    out = torch.empty((b, 3 + c, m, u), device=coords.device, dtype=torch.float32)
    co, ce, ix = coords.contiguous(), centers.contiguous(), idx.contiguous()
    fe = feat.contiguous() if feat is not None else None
    _lib.check(_lib.load().lion_group_points_forward(_lib.ptr(co), _lib.ptr(ce), _lib.ptr(fe), _lib.ptr(ix), b, c, n,
                                                     m, u, _lib.ptr(out), _lib.stream_ptr(coords.device)),
               "group_points_forward")
    return out


def linear_rows(x, weight, bias=None, act=0, slope=0.0):
    """nn.Linear on [B, K]: act(x W^T + b) in ONE launch on the fp32 MFMA kernel (lion_linear_forward);
    act 0 none / 1 relu / 2 leaky-relu(slope)."""
    lib = _lib.load()
    x = x.contiguous()
    b, k = x.shape
    o = weight.shape[0]
    wp = pw_packed_weight(weight)
    y = torch.empty((b, o), device=x.device, dtype=torch.float32)
    bias_c = bias.detach().contiguous() if bias is not None else None
    _lib.check(lib.lion_linear_forward(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(bias_c), b, k, o, int(act), float(slope),
                                       _lib.ptr(y), _lib.stream_ptr(x.device)), "linear_forward")
    return y


def linear_attention_core(qkv, heads, dim_head):
    """[B, 3*heads*dim_head, N] -> [B, heads*dim_head, N]: softmax over N of k, ctx = k v^T, out = ctx^T q."""
    qkv = qkv.contiguous()
    b, n = qkv.shape[0], qkv[0, 0].numel()
    out = torch.empty((b, heads * dim_head) + tuple(qkv.shape[2:]), device=qkv.device, dtype=torch.float32)
    _lib.check(_lib.load().lion_linear_attention_core(_lib.ptr(qkv), b, heads, dim_head, n, _lib.ptr(out),
                                                      _lib.stream_ptr(qkv.device)), "linear_attention_core")
    return out


def row_stats(x):
    """x f32[B, C, ...] -> f32[B*C, 2]: per (batch, channel) row the sum and the sum of squares (one streaming pass)"""
    lib = _lib.load()
    x = x.contiguous()
    rows = x.shape[0] * x.shape[1]
    stats = torch.empty((rows, 2), device=x.device, dtype=torch.float32)
    _lib.check(lib.lion_row_stats(_lib.ptr(x), rows, x[0, 0].numel(), _lib.ptr(stats), _lib.stream_ptr(x.device)), "row_stats")
    return stats


def affine_swish(x, A, Bs, reduce_max=False, add=None):
    """swish(x*A+Bs) per (batch, channel) row; reduce_max: max over the last (neighbour) dimension too; add: a tensor of
    the output's shape summed onto the result in the same pass."""
    lib = _lib.load()
    b, c = x.shape[:2]
    st = _lib.stream_ptr(x.device)
    if add is not None and not reduce_max:
        add = add.contiguous()
        assert add.shape == x.shape and add.dtype == torch.float32
        y = torch.empty_like(x)
        _lib.check(lib.lion_affine_swish_add(_lib.ptr(x), _lib.ptr(A), _lib.ptr(Bs), _lib.ptr(add), b * c, x[0, 0].numel(),
                                             _lib.ptr(y), st), "affine_swish_add")
        return y
    if reduce_max:
        m, u = x.shape[2], x.shape[3]
        y = torch.empty((b, c, m), device=x.device, dtype=torch.float32)
        _lib.check(lib.lion_affine_swish_max(_lib.ptr(x), _lib.ptr(A), _lib.ptr(Bs), b * c, m, u, _lib.ptr(y), st),
                   "affine_swish_max")
        return y
    y = torch.empty_like(x)
    _lib.check(lib.lion_affine_swish(_lib.ptr(x), _lib.ptr(A), _lib.ptr(Bs), b * c, x[0, 0].numel(), _lib.ptr(y), st),
               "affine_swish")
    return y


def shared_mlp(x, convs, adagns, style, reduce_max=False, add=None):
    """[1x1 conv -> AdaGN -> Swish] x n (inference).  Large activations: the conv runs on the MFMA kernel,
    reads the RAW output of the previous conv and applies that layer's AdaGN + Swish in flight, and
    emits its own GroupNorm sums -- one read and one write per layer instead of five passes.  Short
    activations keep the library GEMM (+ one row-sum pass).  Only the last layer needs a stand-alone
    apply pass (optionally with the max over the neighbourhood)."""
    lib = _lib.load()
    pro = None
    for li, (conv, gn) in enumerate(zip(convs, adagns)):
        if (MAX_RECOMPUTE and reduce_max and add is None and li == len(convs) - 1 and x.dim() == 4 and x.shape[3] == 32
                and pw_supported(conv, x)
                and not pw_use_split(None, x.shape[0], x.shape[1], conv.out_channels, x[0, 0].numel())):
            out = pwconv_max_recompute(x, conv, gn, style, pro)
            if out is not None:
                return out
        if pw_supported(conv, x):
            x, st = pwconv_fused(x, conv, pro)
        else:
            if pro is not None:
                x = affine_swish(x, pro[0], pro[1])
            x = conv(x).contiguous()
            b, c = x.shape[:2]
            st = torch.empty((b, c, 1, 2), device=x.device, dtype=torch.float32)
            _lib.check(lib.lion_row_stats(_lib.ptr(x), b * c, x[0, 0].numel(), _lib.ptr(st),
                                          _lib.stream_ptr(x.device)), "row_stats")
        f, g = gn.affine(style)
        A, Bs, _ = groupnorm_fold(st, gn.norm, f, g, x[0, 0].numel())
        pro = (A, Bs)
    out = affine_swish(x, pro[0], pro[1], reduce_max, add=None if reduce_max else add)
    return out + add if (add is not None and reduce_max) else out


def adagn_swish(x, adagn, style, reduce_max=False):
    """swish(AdaGN(x)) for a 1-D [B,C,N] / 2-D [B,C,M,U] activation in 3 launches (row sums, fold,
    apply); reduce_max=True additionally takes the max over the last (neighbour) dimension."""
    lib = _lib.load()
    x = x.contiguous()
    b, c = x.shape[:2]
    L = x[0, 0].numel()
    stats = torch.empty((b, c, 1, 2), device=x.device, dtype=torch.float32)
    st = _lib.stream_ptr(x.device)
    _lib.check(lib.lion_row_stats(_lib.ptr(x), b * c, L, _lib.ptr(stats), st), "row_stats")
    f, g = adagn.affine(style)
    A, Bs, _ = groupnorm_fold(stats, adagn.norm, f, g, L)
    if reduce_max:
        m, u = x.shape[2], x.shape[3]
        y = torch.empty((b, c, m), device=x.device, dtype=torch.float32)
        _lib.check(lib.lion_affine_swish_max(_lib.ptr(x), _lib.ptr(A), _lib.ptr(Bs), b * c, m, u, _lib.ptr(y), st),
                   "affine_swish_max")
        return y
    y = torch.empty_like(x)
    _lib.check(lib.lion_affine_swish(_lib.ptr(x), _lib.ptr(A), _lib.ptr(Bs), b * c, L, _lib.ptr(y), st), "affine_swish")
    return y


# ---- round 6: layout / concatenation passes of a denoiser forward on this library's kernels (no ATen copy in a captured step) --

def broadcast_rows(temb):
    """a [B, C, N] time embedding that is a per-sample row expanded along N (stride 0) -> ([B-strided, C] view, row stride in
    floats), else None.  The row stride is 0 when the batch dimension is an expand of one row as well."""
    if temb is None or temb.dim() != 3 or temb.stride(2) != 0 or temb.stride(1) != 1 or temb.dtype != torch.float32:
        return None
    rows = temb[:, :, 0]
    return rows, int(rows.stride(0))


def latent_unpack(x, n_points, d, want_all=True, want_coords=True, want_rest=True):
    """x [B, N*D(,1,1)] point-major latent -> (all [B,D,N], coords [B,3,N], rest [B,D-3,N]) channel-major, one launch
    (was: view + permute + contiguous, slice + contiguous twice -- three ATen copies per step)."""
    b = x.shape[0]
    xc = x.contiguous()
    mk = lambda c, want: torch.empty((b, c, n_points), device=x.device, dtype=torch.float32) if want and c > 0 else None
    al, co, re = mk(d, want_all), mk(3, want_coords), mk(d - 3, want_rest)
    _lib.check(_lib.load().lion_latent_unpack(_lib.ptr(xc), b, n_points, d, _lib.ptr(al), _lib.ptr(co), _lib.ptr(re),
                                              _lib.stream_ptr(x.device)), "latent_unpack")
    return al, co, re


def concat_broadcast(a, temb):
    """torch.cat([a, temb], dim=1) for a [B,Ca,N] and a broadcast time embedding (broadcast_rows), one launch of this library;
    None when the operands do not qualify (the caller then uses torch.cat)."""
    br = broadcast_rows(temb)
    if br is None or a.dim() != 3 or a.dtype != torch.float32 or a.shape[2] % 4 or not a.is_contiguous() \
            or temb.shape[0] != a.shape[0] or temb.shape[2] != a.shape[2]:
        return None
    rows, ld = br
    b, ca, n = a.shape
    ct = rows.shape[1]
    out = torch.empty((b, ca + ct, n), device=a.device, dtype=torch.float32)
    rc = _lib.load().lion_concat_broadcast(_lib.ptr(a), _lib.ptr(rows), b, ca, ct, n, ld, _lib.ptr(out),
                                           _lib.stream_ptr(a.device))
    if rc == -2:      # LION_EUNSUPPORTED (more than 65535 (sample, channel) rows): the caller's torch.cat
        return None
    _lib.check(rc, "concat_broadcast")
    return out


def three_nn_interpolate_cat(points, centers, cfeat, temb, skip):
    """PointNetFPModule's [interpolate(cat(cfeat, temb)) ; skip] in one pass (lion_three_nn_interpolate_cat_forward);
    temb (a broadcast time embedding over the M centres) and skip may be None.  None when the operands do not qualify."""
    rows, ld, c2 = None, 0, 0
    if temb is not None:
        br = broadcast_rows(temb)
        if br is None or temb.shape[0] != cfeat.shape[0]:
            return None
        rows, ld = br
        c2 = rows.shape[1]
    if cfeat.dtype != torch.float32 or (skip is not None and skip.dtype != torch.float32):
        return None
    from .functional import backend as _bk
    fn = getattr(_bk._backend, "three_nearest_neighbors_interpolate_cat_forward", None)
    if fn is None:          # an operator backend without the fused entry point (tests run the oracle's): the composition
        return None
    points, centers, cfeat = points[:, :3].contiguous(), centers[:, :3].contiguous(), cfeat.contiguous()
    skip = None if skip is None else skip.contiguous()
    return fn(points, centers, cfeat, rows, ld, skip)[0]


# ---- D2: global denoiser on channel-major activations (csrc/skinny.hip) -----------------------------------

def _rows2d(x):
    """[B, C(, 1, 1)] as (2-D view, row stride in floats) for lion_to_channel_major, or None (not a float32 GPU tensor with
    unit channel stride)"""
    b, c = x.shape[0], x.shape[1]
    v = x.reshape(b, c) if x.dim() != 2 else x
    if not (v.is_cuda and v.dtype == torch.float32 and (c == 1 or v.stride(1) == 1)):
        return None
    return v, (0 if b == 1 else int(v.stride(0)))


def to_channel_major_pair(x, t):
    """(x, t) [B, C, 1, 1] -> their channel-major forms in ONE launch of this library (t may be an expanded single row); the
    torch formulation when either operand does not qualify"""
    rx, rt = _rows2d(x), _rows2d(t)
    if rx is None or rt is None or torch.is_grad_enabled():
        if t.shape[0] == 1 and x.shape[0] > 1:
            t = t.expand(x.shape[0], *t.shape[1:])
        return to_channel_major(x), to_channel_major(t)
    (vx, ldx), (vt, ldt) = rx, rt
    b = x.shape[0]
    nb = (b + 31) // 32
    ox = torch.empty((nb, vx.shape[1], 32), device=x.device, dtype=torch.float32)
    ot = torch.empty((nb, vt.shape[1], 32), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().lion_to_channel_major(_lib.ptr(vx), ldx, vx.shape[1], _lib.ptr(ox), _lib.ptr(vt), ldt, vt.shape[1],
                                                 _lib.ptr(ot), b, _lib.stream_ptr(x.device)), "to_channel_major")
    return ox, ot


def to_channel_major(x):
    """[B, C, 1, 1] (or [B, C]) -> [nb, C, 32] with the batch zero-padded to a multiple of 32."""
    b, c = x.shape[0], x.shape[1]
    nb = (b + 31) // 32
    xt = x.reshape(b, c)
    if b != nb * 32:
        xt = torch.cat([xt, xt.new_zeros(nb * 32 - b, c)], 0)
    return xt.reshape(nb, 32, c).transpose(1, 2).contiguous()


def from_channel_major(xt, b):
    nb, c, _ = xt.shape
    if xt.is_cuda and xt.dtype == torch.float32 and xt.is_contiguous() and not torch.is_grad_enabled():
        y = torch.empty((b, c, 1, 1), device=xt.device, dtype=torch.float32)
        _lib.check(_lib.load().lion_from_channel_major(_lib.ptr(xt), b, c, _lib.ptr(y), _lib.stream_ptr(xt.device)),
                   "from_channel_major")
        return y
    return xt.transpose(1, 2).reshape(nb * 32, c)[:b].reshape(b, c, 1, 1).contiguous()


def _sk_pack(weight):
    cout, cin = weight.shape[:2]
    wp = torch.empty((_lib.load().lion_skinny_packed_floats(cout, cin),), device=weight.device, dtype=torch.float32)
    w_c = weight.detach().reshape(cout, cin).contiguous()
    _lib.check(_lib.load().lion_skinny_pack_weights(_lib.ptr(w_c), cout, cin, _lib.ptr(wp),
                                                    _lib.stream_ptr(weight.device)), "skinny_pack_weights")
    return wp


_SK_CACHE = WeightCache(_sk_pack)


def skinny_packed_weight(weight):
    """[Cout,Cin,1,1] -> tile-major packed copy (lion_skinny_pack_weights); cached per (storage, version)."""
    return _SK_CACHE.get(weight)


def skinny_conv(pin, conv, bias_in=None, act_in=0, add=None):
    """one 1x1 conv of the global denoiser: pin [ks_in, nb, Cin, 32] partials (or [nb, Cin, 32]) -> raw partial
    sums [ks, nb, Cout, 32] of conv.weight @ (act_in(sum pin + bias_in) + add); conv.bias is applied by the
    consumer (see lion_skinny_gemm)."""
    lib = _lib.load()
    if pin.dim() == 3:
        pin = pin.unsqueeze(0)
    ks_in, nb, cin, _ = pin.shape
    cout = conv.out_channels
    wp = skinny_packed_weight(conv.weight)
    out = torch.empty((lib.lion_skinny_splits(cin, cout), nb, cout, 32), device=pin.device, dtype=torch.float32)
    _lib.check(lib.lion_skinny_gemm(_lib.ptr(pin), ks_in, _lib.ptr(bias_in), int(act_in), _lib.ptr(add), _lib.ptr(wp),
                                    nb, cin, cout, _lib.ptr(out), _lib.stream_ptr(pin.device)), "skinny_gemm")
    return out


def skinny_conv_se_finish(pin, conv, A, bias_a, resid, act_in=1):
    """the block's last GEMM (SE fc2) with the block's tail in its epilogue: resid + relu(sum A + bias_a) * sigmoid(conv.weight @
    act_in(sum pin)) -- skinny_conv + skinny_finish in one launch (lion_skinny_gemm_se_finish); None when the layer needs k-splits."""
    lib = _lib.load()
    if pin.dim() == 3:
        pin = pin.unsqueeze(0)
    ks_in, nb, cin, _ = pin.shape
    cout = conv.out_channels
    if lib.lion_skinny_splits(cin, cout) != 1:
        return None
    wp = skinny_packed_weight(conv.weight)
    y = torch.empty((nb, cout, 32), device=pin.device, dtype=torch.float32)
    _lib.check(lib.lion_skinny_gemm_se_finish(_lib.ptr(pin), ks_in, None, int(act_in), _lib.ptr(wp), nb, cin, cout, _lib.ptr(A),
                                              A.shape[0], _lib.ptr(bias_a), _lib.ptr(resid), _lib.ptr(y),
                                              _lib.stream_ptr(pin.device)), "skinny_gemm_se_finish")
    return y


def skinny_finish(A, bias_a, Bp=None, resid=None):
    """[nb, C, 32] from partials: sum A + bias_a (Bp None) or resid + relu(sum A + bias_a) * sigmoid(sum Bp)."""
    ks_a, nb, c, _ = A.shape
    y = torch.empty((nb, c, 32), device=A.device, dtype=torch.float32)
    _lib.check(_lib.load().lion_skinny_finish(_lib.ptr(A), ks_a, _lib.ptr(bias_a), _lib.ptr(Bp),
                                              0 if Bp is None else Bp.shape[0], _lib.ptr(resid), nb, c,
                                              0 if Bp is None else 1, _lib.ptr(y), _lib.stream_ptr(A.device)),
               "skinny_finish")
    return y
