"""Conv3d (3x3x3, pad 1) on the fp32 matrix cores: lion_conv3d_k3_forward over the C ABI."""
import os

import torch

from . import _lib
from ._wcache import WeightCache

# Which arithmetic the voxel convolutions run in (both are fp32-accurate, csrc/conv3d_split.hip):
#   True  -> fp16 x 2 split operands on the 16-bit MFMA pipe where the shape allows (Cin % 16 == 0), fp32 MFMA elsewhere
#   False -> the exact-fp32 MFMA kernel everywhere (csrc/conv3d.hip)
SPLIT = os.environ.get("LION_CONV_SPLIT", "1") != "0"
SPLIT_MIN_R = int(os.environ.get("LION_CONV_SPLIT_MIN_R", "8"))   # A/B switch: 16 keeps the r = 8 layers on the fp32 kernel


def supported(cin, cout, r):
    return r in (8, 16, 32) and cout % 32 == 0 and cin >= 1


def split_supported(cin, cout, r):
    """what csrc/conv3d_split.hip can run"""
    return r in (8, 16, 32) and cout % 32 == 0 and cin >= 16 and cin % 16 == 0


def split_preferred(cin, cout, r):
    """where it is the faster kernel (tools/conv_split_bench.py, B=32): 2.2x at 64->64 r=32, 2.5x at 128->128 r=16,
    1.5x at 32->32 r=32, 2.1x at r=8 (the pipelined half-sample kernel) -- everywhere it can run."""
    return split_supported(cin, cout, r) and r >= SPLIT_MIN_R


def use_split(split, cin, cout, r):
    """split: None = module policy (SPLIT and split_preferred), True = wherever supported, False = never."""
    if split is None:
        return SPLIT and split_preferred(cin, cout, r)
    return bool(split) and split_supported(cin, cout, r)


def _split_pack(weight):
    cout, cin = weight.shape[:2]
    lib = _lib.load()
    wp = torch.empty((lib.lion_conv3d_split_packed_halfs(cout, cin),), device=weight.device, dtype=torch.int16)
    w_c = weight.detach().contiguous()
    _lib.check(lib.lion_conv3d_split_pack_weights(_lib.ptr(w_c), cout, cin, _lib.ptr(wp),
                                                  _lib.stream_ptr(weight.device)), "conv3d_split_pack_weights")
    return wp


_SPLIT_CACHE = WeightCache(_split_pack)


def split_packed_weight(weight):
    """[Cout,Cin,3,3,3] -> fp16 hi / lo pieces in MFMA operand order + the tensor's power-of-two scale; cached per
    (storage, version, generation), see _wcache.py."""
    return _SPLIT_CACHE.get(weight)


def _pack(weight):
    cout, cin = weight.shape[:2]
    lib = _lib.load()
    wp = torch.empty((lib.lion_conv3d_packed_floats(cout, cin),), device=weight.device, dtype=torch.float32)
    w_c = weight.detach().contiguous()  # local reference: see fused_ops.groupnorm_fold
    _lib.check(lib.lion_conv3d_pack_weights(_lib.ptr(w_c), cout, cin, _lib.ptr(wp),
                                            _lib.stream_ptr(weight.device)), "conv3d_pack_weights")
    return wp


_PACK_CACHE = WeightCache(_pack)


def packed_weight(weight):
    """[Cout,Cin,3,3,3] -> packed [ceil4(Cin),27,Cout]; cached per (storage, version), see _wcache.py."""
    return _PACK_CACHE.get(weight)


def conv3d_k3(x, weight, bias=None, split=None, packed=None, occ=None):
    """x [B,Cin,r,r,r] fp32 -> [B,Cout,r,r,r]; Cin is zero-padded to a multiple of 4 if needed.
    split: None = the module default (SPLIT), False = the exact-fp32 MFMA kernel, True = split operands if supported.
    packed: callable(kind) -> the packed copy of `weight` for kind in ("split", "f32"), for callers whose `weight` is a
    derived temporary and who cache its packed forms under the ORIGINAL parameter (the data-gradient path).
    occ: the tile occupancy of a freshly voxelised `x` (fused_ops.conv3d_occupancy(counts)[0]): tiles without a point within
    one voxel skip their K loop (output = bias exactly); split kernel only, ignored otherwise."""
    _lib.require_cuda(x)
    b, cin, r = x.shape[0], x.shape[1], x.shape[2]
    cout = weight.shape[0]
    if use_split(split, cin, cout, r) and weight.shape[1] == cin:
        x = x.contiguous()
        y = torch.empty((b, cout, r, r, r), device=x.device, dtype=torch.float32)
        bias_c = bias.detach().contiguous() if bias is not None else None
        wp = packed("split") if packed is not None else split_packed_weight(weight)
        _lib.check(_lib.load().lion_conv3d_k3_split_forward(
            _lib.ptr(x), _lib.ptr(wp), _lib.ptr(bias_c), b, cin, cout, r, None, None, None, None, _lib.ptr(y), None,
            _lib.ptr(occ) if (occ is not None and r >= 16) else None, _lib.stream_ptr(x.device)), "conv3d_k3_split_forward")
        return y
    if cin % 4:
        pad = 4 - cin % 4
        x = torch.cat([x, x.new_zeros(b, pad, r, r, r)], dim=1)
        cin_p = cin + pad
    else:
        cin_p = cin
    x = x.contiguous()
    wp = packed("f32") if packed is not None else packed_weight(weight)
    y = torch.empty((b, cout, r, r, r), device=x.device, dtype=torch.float32)
    bias_c = bias.detach().contiguous() if bias is not None else None
    _lib.check(_lib.load().lion_conv3d_k3_forward(
        _lib.ptr(x), _lib.ptr(wp), _lib.ptr(bias_c),
        b, cin_p, cout, r, _lib.ptr(y), _lib.stream_ptr(x.device)), "conv3d_k3_forward")
    return y


def _mirror(weight):
    wt = weight.detach().flip(2, 3, 4).transpose(0, 1)
    pad = (-wt.shape[0]) % 32
    if pad:
        wt = torch.cat([wt, wt.new_zeros((pad,) + tuple(wt.shape[1:]))], 0)
    return wt.contiguous()


_DGRAD_CACHE = WeightCache(_mirror)


def dgrad_weight(weight):
    """weights of the data-gradient convolution: grad_x = conv3d(grad_y, W'), W'[ci, co, kd, kh, kw] =
    W[co, ci, 2-kd, 2-kh, 2-kw] (stride 1, padding 1: the transposed convolution is again a 3x3x3 / pad 1
    convolution with the channels swapped and the taps mirrored); its output channels (= Cin) are zero-padded to a
    multiple of 32, the kernel's channel tile (callers slice); cached per (storage, version)."""
    return _DGRAD_CACHE.get(weight)


# Packed forms of the MIRROR, cached under the original parameter: keyed on the mirror tensor itself (a temporary that is
# rebuilt at a new address after every optimizer step) each step would leave one dead mirror + packed entry per layer in
# the LRU, pinned in HBM until 256 newer entries pushed it out.
_DGRAD_SPLIT_CACHE = WeightCache(lambda w: _split_pack(dgrad_weight(w)))
_DGRAD_PACK_CACHE = WeightCache(lambda w: _pack(dgrad_weight(w)))


def conv3d_k3_dgrad(gy, weight, counts=None):
    """grad_x (with Cin padded to a multiple of 32, callers slice) = conv3d_k3(gy, mirror(weight)); every derived tensor
    is cached under `weight` (storage, version, generation) and replaced in place when the parameter changes.
    counts: the point counts of the voxelised grid x was (int32 [B, r^3]): its gradient is read by the voxelisation's backward
    at voxels that hold a point only, so tiles without a point within one voxel are not computed (they hold zeros)."""
    wt = dgrad_weight(weight)
    occ = None
    r = gy.shape[2]
    if counts is not None and TRAIN_SPARSE and r in (16, 32) and use_split(None, gy.shape[1], wt.shape[0], r):
        from . import fused_ops
        occ = fused_ops.conv3d_occupancy(counts, r, wt.shape[0], gy.shape[0])[0]
    return conv3d_k3(gy, wt, None, occ=occ,
                     packed=lambda kind: (_DGRAD_SPLIT_CACHE if kind == "split" else _DGRAD_PACK_CACHE).get(weight))


# weight gradient on the 16-bit matrix pipe at fp32 accuracy where Cin % 8 == 0 (csrc/conv3d_wgrad.hip, round 4);
# training: the convolution that reads a freshly voxelised grid skips the tiles without a point within one voxel (forward; the
# weight gradient skips all-zero input windows by itself).  LION_TRAIN_SPARSE=0: dense.
TRAIN_SPARSE = os.environ.get("LION_TRAIN_SPARSE", "1") != "0"
# LION_WGRAD_SPLIT=0: the exact-fp32 MFMA kernel everywhere
WGRAD_SPLIT = os.environ.get("LION_WGRAD_SPLIT", "1") != "0"


def conv3d_k3_wgrad(x, gy, weight_shape, split=None, counts=None):
    """weight gradient [Cout,Cin,3,3,3] of the 3x3x3 / pad 1 conv on the MFMA kernels (x [B,Cin,r,r,r], Cin % 4 == 0).
    split: None = WGRAD_SPLIT, True / False = the split-operand kernel where it applies / the exact-fp32 kernel.
    counts: x is a freshly voxelised grid with these point counts (int32 [B, r^3]): its empty tiles are not loaded."""
    lib = _lib.load()
    b, cin, r = x.shape[0], x.shape[1], x.shape[2]
    cout = gy.shape[1]
    gw = torch.empty(tuple(weight_shape), device=x.device, dtype=torch.float32)
    sparse = counts is not None and TRAIN_SPARSE and (WGRAD_SPLIT if split is None else split) and cin % 8 == 0
    n = (lib.lion_conv3d_wgrad_sparse_workspace_floats if sparse else lib.lion_conv3d_wgrad_workspace_floats)(b, cin, cout, r)
    ws = torch.empty((n,), device=x.device, dtype=torch.float32)
    x_c, gy_c = x.contiguous(), gy.contiguous()
    st = _lib.stream_ptr(x.device)
    if sparse:
        cnt_c = counts.contiguous()
        rc = lib.lion_conv3d_k3_wgrad_split_sparse(_lib.ptr(x_c), _lib.ptr(gy_c), _lib.ptr(cnt_c), b, cin, cout, r, _lib.ptr(gw),
                                                   _lib.ptr(ws), n, st)
        if rc != -2:
            _lib.check(rc, "conv3d_k3_wgrad_split_sparse")
            return gw
    if (WGRAD_SPLIT if split is None else split) and cin % 8 == 0:
        rc = lib.lion_conv3d_k3_wgrad_split(_lib.ptr(x_c), _lib.ptr(gy_c), b, cin, cout, r, _lib.ptr(gw), _lib.ptr(ws), n, st)
        if rc != -2:   # LION_EUNSUPPORTED (alignment): the fp32 kernel below
            _lib.check(rc, "conv3d_k3_wgrad_split")
            return gw
    _lib.check(lib.lion_conv3d_k3_wgrad(_lib.ptr(x_c), _lib.ptr(gy_c), b, cin, cout, r, _lib.ptr(gw), _lib.ptr(ws), n, st),
               "conv3d_k3_wgrad")
    return gw


class _Conv3dK3(torch.autograd.Function):
    """forward, data gradient (the convolution with the mirrored, channel-swapped weights) and weight gradient
    (voxels on the MFMA k axis) on the fp32 MFMA kernels; shapes they do not cover fall back to ATen's
    convolution_backward (MIOpen)."""

    @staticmethod
    def forward(ctx, x, weight, bias, counts=None):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.counts = counts      # (not a differentiable input; kept for the data gradient's empty tiles)
        occ = None
        r, cout = x.shape[2], weight.shape[0]
        if counts is not None and TRAIN_SPARSE and r in (16, 32) and use_split(None, x.shape[1], cout, r):
            from . import fused_ops
            occ = fused_ops.conv3d_occupancy(counts, r, cout, x.shape[0])[0]
        return conv3d_k3(x, weight, bias, occ=occ)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        from . import train_ops
        gb_tagged = train_ops.tagged_channel_sum(gy, weight.shape[0])   # from the AdaGN behind this convolution, if it made gy
        gy = gy.contiguous()
        cout, cin = weight.shape[:2]
        r = x.shape[2]
        own_dgrad = ctx.needs_input_grad[0] and r in (8, 16, 32) and cout % 4 == 0
        own_wgrad = ctx.needs_input_grad[1] and supported(cin, cout, r)
        want_gb = ctx.has_bias and ctx.needs_input_grad[2]
        gx = gw = gb = None
        lib_mask = [ctx.needs_input_grad[0] and not own_dgrad, ctx.needs_input_grad[1] and not own_wgrad,
                    want_gb and not own_wgrad]
        if any(lib_mask):
            from . import _fallback
            _fallback.note("conv_ops._Conv3dK3.backward", f"convolution_backward mask {lib_mask} for {cin}->{cout} at r={r}")
            gx, gw, gb = torch.ops.aten.convolution_backward(
                gy, x, weight, [weight.shape[0]] if ctx.has_bias else None,
                [1, 1, 1], [1, 1, 1], [1, 1, 1], False, [0, 0, 0], 1, lib_mask)
        if own_dgrad:
            gx = conv3d_k3_dgrad(gy, weight, ctx.counts)
            if gx.shape[1] != cin:
                gx = gx[:, :cin].contiguous()
        if own_wgrad:
            if cin % 4:  # the kernel wants Cin % 4 == 0: zero input channels, their gradient rows sliced away
                pad = 4 - cin % 4
                xp = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, 0, 0, pad))
                gw = conv3d_k3_wgrad(xp, gy, (cout, cin + pad, 3, 3, 3))[:, :cin].contiguous()
            else:
                gw = conv3d_k3_wgrad(x, gy, weight.shape, counts=ctx.counts)
            if want_gb and gb_tagged is not None:
                gb = gb_tagged   # sum of dx over batch and voxels, a by-product of the AdaGN backward (train_ops.tag_channel_sum)
            elif want_gb:
                # one streaming pass (row sums per (b, c), then a [B, C] -> [C] sum) instead of ATen's strided
                # reduction over dims (0, 2, 3, 4): 240-277 us -> ~60 us at [32, 64, 32^3]
                from . import fused_ops
                gb = fused_ops.row_stats(gy)[:, 0].reshape(gy.shape[0], gy.shape[1]).sum(0)
        return gx, gw, gb, None


def conv3d_module(conv: torch.nn.Conv3d, x):
    """Run an nn.Conv3d(k=3, s=1, p=1) module through the MFMA kernel when the shape is supported
    (r in {8,16,32}, Cout % 32 == 0, fp32, HIP tensor); otherwise the library convolution."""
    ok = (x.is_cuda and x.dtype == torch.float32 and conv.kernel_size == (3, 3, 3)
          and conv.stride == (1, 1, 1) and conv.padding == (1, 1, 1) and conv.dilation == (1, 1, 1)
          and conv.groups == 1 and x.shape[2] == x.shape[3] == x.shape[4]
          and supported(conv.in_channels, conv.out_channels, x.shape[2])
          and not torch.is_autocast_enabled())
    if not ok:
        from . import _fallback
        _fallback.note("conv_ops.conv3d_module", f"Conv3d {conv.in_channels}->{conv.out_channels} k={conv.kernel_size} on "
                       f"{tuple(x.shape)} {x.dtype} {x.device.type}, autocast={torch.is_autocast_enabled()}")
        return conv(x)
    if torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad):
        # always the module's own weight tensor: the packed / mirrored copies are cached per (storage, version),
        # a padded temporary would miss every step and pin dead copies in HBM (odd Cin is padded inside
        # conv3d_k3 / backward instead)
        # a grid straight from the voxelisation carries its point counts (functional/voxelization.py): the empty tiles of THIS
        # convolution follow from them (the version check voids the tag after an in-place write to the grid)
        tag = getattr(x, "_lion_voxel_counts", None)
        counts = tag[0] if (tag is not None and tag[1] == x._version and tag[0].shape[0] == x.shape[0]) else None
        return _Conv3dK3.apply(x, conv.weight, conv.bias, counts)
    return conv3d_k3(x, conv.weight, conv.bias)
