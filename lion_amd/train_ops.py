"""Training forms of GroupNorm / AdaGN (+ Swish) on the library's own kernels (csrc/norm_train.hip).

``adagn_act(x, norm, factor, bias, act)`` == ``act(norm(x) * factor[:, :, None..] + bias[:, :, None..])`` with
``norm = nn.GroupNorm(G, C)`` (reference models/adagn.py:45-65 followed by models/pvcnn2_ada.py:78-84), differentiable in
x, norm.weight, norm.bias, factor, bias.  ATen evaluates that expression in five passes over the activation forward and
about ten backward; here the forward is two passes (row sums; one fused apply) and the backward three (row sums of the
activation gradient; one fused apply) plus one small kernel per direction for the [B, C] scalar algebra (double inside).
"""
import os

import torch
from torch.autograd.function import once_differentiable

from . import _lib

ENABLED = os.environ.get("LION_TRAIN_FUSE", "1") != "0"
PWCONV = os.environ.get("LION_TRAIN_PWCONV", "1") != "0"   # 1x1 convolutions of the training path on own kernels
# ... from this many columns (B x L) on.  Round 4 had measured the library GEMM faster below 2^18 columns (prior step 82 vs 88 ms);
# with round 5's kernels it is the other way round (round 6, one box, profiles/r06_train_pwconv_ab.txt): VAE step 110.1 ms at
# 2^18 -> 101.8 at 2^16 -> 96.9 at 2^13 -> 96.7 at 2^11 -> 97.1 with every layer; prior step 61.9 -> 58.4 -> 56.1 -> 56.1 -> 72.9.
# The last number is the global prior's 2048-wide layers on [32, C, 1, 1] activations (32 columns: a weight-streaming skinny
# GEMM, not what these kernels are tiled for): they stay on the rocBLAS matrix product, everything from 512 columns on is ours.
PWCONV_MIN_COLS = int(os.environ.get("LION_TRAIN_PWCONV_MIN_COLS", "512"))
DEVOX_FUSED = os.environ.get("LION_TRAIN_DEVOX_FUSED", "1") != "0"       # AdaGN -> SE3d -> devoxelize as one op (PVConv tail)
DROPOUT_FUSED = os.environ.get("LION_TRAIN_DROPOUT_FUSED", "1") != "0"   # nn.Dropout behind AdaGN + Swish inside the activation pass


def usable(x, need_grad=True) -> bool:
    """need_grad=False: also without autograd (the frozen VAE's style encoder inside a prior training step, an encode at
    inference) -- for blocks that have no fused inference path of their own (models/pvcnn2.py); the forward kernels are the same"""
    return (ENABLED and x.is_cuda and x.dtype == torch.float32 and (torch.is_grad_enabled() or not need_grad)
            and not torch.is_autocast_enabled() and x.dim() >= 3 and x[0, 0].numel() >= 1)


def _rows(t):
    return t.reshape(-1).contiguous()


def _bc_view(t, B, C):
    """the [B, C] a factor / bias stands for, by broadcasting RULES (not by element count: at B == C a [B, 1] and a
    [1, C] factor have the same number of elements): [B, C], [1, C], [B, 1], [C], scalars, and the same with trailing
    singleton dimensions ([B, C, 1, 1], [B, 1, 1] ...)"""
    while t.dim() > 2 and t.shape[-1] == 1:
        t = t.squeeze(-1)
    if t.dim() > 2:
        raise ValueError(f"AdaGN factor / bias of shape {tuple(t.shape)} does not broadcast to [B, C] = [{B}, {C}]")
    return t.expand(B, C)


def _rowview(t, B, C):
    """(pointer holder, row stride) of a [B, C] float32 view whose channel stride is 1 (a chunk of the [B, 2C] AdaGN
    projection is such a view) -- anything else is made contiguous"""
    if tuple(t.shape) != (B, C):      # a broadcastable factor: materialise the [B, C] it stands for
        t = _bc_view(t, B, C)
    if t.dtype != torch.float32 or t.stride(1) != 1 or (B > 1 and t.stride(0) == 0):
        t = t.float().contiguous()
    return t, int(t.stride(0))


def _param_grads(pw, want_w, want_b, want_dxs=False):
    """pw [B, C, 3] (per-sample terms of the GroupNorm weight / bias gradients and of the row sums of dx) -> (d weight [C],
    d bias [C], sum over batch and positions of dx [C] or None) in one launch"""
    if not (want_w or want_b or want_dxs):
        return None, None, None
    B, C = pw.shape[:2]
    dgw, dgb = torch.empty(C, device=pw.device, dtype=torch.float32), torch.empty(C, device=pw.device, dtype=torch.float32)
    dxs = torch.empty(C, device=pw.device, dtype=torch.float32) if want_dxs else None
    _lib.check(_lib.load().lion_gn_train_param_grads(_lib.ptr(pw), B, C, _lib.ptr(dgw), _lib.ptr(dgb), _lib.ptr(dxs),
                                                     _lib.stream_ptr(pw.device)), "gn_train_param_grads")
    return (dgw if want_w else None), (dgb if want_b else None), dxs


def _affine_grad_buffers(B, C, has_f, has_b, dev):
    """(d factor, d bias, row stride): with both wanted, the two halves of ONE [B, 2C] buffer -- `halves` below hands that
    buffer on as the gradient of the [B, 2C] AdaGN projection without a cat"""
    if has_f and has_b:
        d = torch.empty(B, 2 * C, device=dev, dtype=torch.float32)
        return d[:, :C], d[:, C:], 2 * C
    return (torch.empty(B, C, device=dev, dtype=torch.float32) if has_f else None,
            torch.empty(B, C, device=dev, dtype=torch.float32) if has_b else None, C)


class _Halves(torch.autograd.Function):
    """e [B, 2C] -> (e[:, :C], e[:, C:]), as `chunk(2, 1)`; backward: when the two gradients are the halves of one buffer (what
    the AdaGN ops above produce) that buffer IS the gradient -- chunk's backward cats them (122 launches per VAE step)."""

    @staticmethod
    def forward(ctx, e):
        c = e.shape[1] // 2
        ctx.shape = tuple(e.shape)
        return e.narrow(1, 0, c), e.narrow(1, c, c)

    @staticmethod
    def backward(ctx, ga, gb):
        B, C2 = ctx.shape
        C = C2 // 2
        if (ga is not None and gb is not None and ga.dtype == gb.dtype == torch.float32 and tuple(ga.shape) == tuple(gb.shape) == (B, C)
                and ga.stride() == gb.stride() == (C2, 1) and gb.data_ptr() == ga.data_ptr() + 4 * C
                and ga.untyped_storage().data_ptr() == gb.untyped_storage().data_ptr()):
            return ga.as_strided((B, C2), (C2, 1))
        if ga is None and gb is None:
            return None
        ga = torch.zeros_like(gb) if ga is None else ga
        gb = torch.zeros_like(ga) if gb is None else gb
        return torch.cat([ga, gb], 1)


def halves(e):
    """the (factor, bias) halves of an AdaGN projection [B, 2C] (models/adagn.py) for the training ops"""
    if e.dim() != 2 or e.shape[1] % 2:
        return tuple(e.chunk(2, 1))
    return _Halves.apply(e)


CHANNEL_SUM_TAG = "_lion_channel_sum"


def tag_channel_sum(dx, dxs):
    """hand the consumer of `dx` (the backward of the convolution that produced the normalised tensor) the sum of dx over batch and
    positions -- its bias gradient -- as an attribute of the gradient tensor, with the tensor's version: an in-place accumulation of
    another gradient into dx (fan-out of the convolution's output) bumps the version and voids the tag."""
    setattr(dx, CHANNEL_SUM_TAG, (dxs, dx._version))


def tagged_channel_sum(g, channels):
    tag = getattr(g, CHANNEL_SUM_TAG, None)
    if tag is None or tag[1] != g._version or tuple(tag[0].shape) != (channels,) or g.shape[1] != channels:
        return None
    return tag[0]


class _AdaGNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gw, gb, factor, bias, groups, eps, act, drop_p=0.0):
        lib = _lib.load()
        x = x.contiguous()
        B, C = x.shape[:2]
        L = x[0, 0].numel()
        st = _lib.stream_ptr(x.device)
        dev = x.device
        stats = torch.empty(B * C, 2, device=dev, dtype=torch.float64)   # shifted sums, double hand-over (csrc/norm_train.hip)
        _lib.check(lib.lion_row_stats64(_lib.ptr(x), B * C, L, _lib.ptr(stats), st), "row_stats64")
        A, Bs, mean, rstd = (torch.empty(B, C, device=dev, dtype=torch.float32) for _ in range(4))
        gwc, gbc = gw.detach().float().contiguous(), gb.detach().float().contiguous()
        f, fs = _rowview(factor.detach(), B, C) if factor is not None else (None, 0)
        bb, bs = _rowview(bias.detach(), B, C) if bias is not None else (None, 0)
        _lib.check(lib.lion_gn_train_fold64(_lib.ptr(stats), _lib.ptr(gwc), _lib.ptr(gbc), _lib.ptr(f), fs, _lib.ptr(bb), bs,
                                          B, C, groups, L, eps, _lib.ptr(A), _lib.ptr(Bs), _lib.ptr(mean), _lib.ptr(rstd),
                                            st), "gn_train_fold64")
        y = torch.empty_like(x)
        keep = 1.0 - float(drop_p)
        if drop_p > 0.0:
            # nn.Dropout behind the activation, in the same pass: the mask comes from a seed torch's generator draws on the device
            # (a replayed graph draws a new one) and is regenerated by the backward passes -- it is never stored
            seed = torch.empty(1, device=dev, dtype=torch.int64).random_()
            _lib.check(lib.lion_affine_act_dropout(_lib.ptr(x), _lib.ptr(A), _lib.ptr(Bs), B * C, L, int(act), _lib.ptr(seed), keep,
                                                   _lib.ptr(y), st), "affine_act_dropout")
        else:
            seed = x.new_empty(0)
            _lib.check(lib.lion_affine_act(_lib.ptr(x), _lib.ptr(A), _lib.ptr(Bs), B * C, L, int(act), _lib.ptr(y), st), "affine_act")
        ctx.save_for_backward(x, A, Bs, mean, rstd, gwc, gbc, f if f is not None else x.new_empty(0), seed, stats)
        ctx.meta = (groups, int(act), factor is not None, bias is not None, fs,
                    None if factor is None else factor.shape, None if bias is None else bias.shape, keep)
        return y

    @staticmethod
    @once_differentiable   # raw kernels: a double backward must raise, not silently treat these gradients as constants
    def backward(ctx, gy):
        lib = _lib.load()
        x, A, Bs, mean, rstd, gwc, gbc, f, seed, stats = ctx.saved_tensors
        groups, act, has_f, has_b, fs, f_shape, b_shape, keep = ctx.meta
        drop = seed.numel() > 0
        gy = gy.contiguous()
        B, C = x.shape[:2]
        L = x[0, 0].numel()
        st = _lib.stream_ptr(x.device)
        dev = x.device
        S = torch.empty(B * C, 2, device=dev, dtype=torch.float32)
        if drop:
            _lib.check(lib.lion_affine_act_dropout_bwd_stats(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(A), _lib.ptr(Bs), B * C, L, act,
                                                             _lib.ptr(seed), keep, _lib.ptr(S), st), "affine_act_dropout_bwd_stats")
        else:
            _lib.check(lib.lion_affine_act_bwd_stats(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(A), _lib.ptr(Bs), B * C, L, act,
                                                     _lib.ptr(S), st), "affine_act_bwd_stats")
        Q, R = torch.empty(B, C, device=dev, dtype=torch.float32), torch.empty(B, C, device=dev, dtype=torch.float32)
        dfac, dbias, dstride = _affine_grad_buffers(B, C, has_f, has_b, dev)
        pw = torch.empty(B, C, 3, device=dev, dtype=torch.float32)
        _lib.check(lib.lion_gn_train_bwd_fold(_lib.ptr(S), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gwc), _lib.ptr(gbc),
                                              _lib.ptr(f) if has_f else None, fs, B, C, groups, L, _lib.ptr(Q), _lib.ptr(R),
                                              _lib.ptr(dfac), _lib.ptr(dbias), dstride, _lib.ptr(pw), _lib.ptr(A), _lib.ptr(stats),
                                              None, None, None, st), "gn_train_bwd_fold")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if drop:
                _lib.check(lib.lion_affine_act_dropout_bwd_apply(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(A), _lib.ptr(Bs), _lib.ptr(Q),
                                                                 _lib.ptr(R), B * C, L, act, _lib.ptr(seed), keep, _lib.ptr(dx), st),
                           "affine_act_dropout_bwd_apply")
            else:
                _lib.check(lib.lion_affine_act_bwd_apply(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(A), _lib.ptr(Bs), _lib.ptr(Q),
                                                         _lib.ptr(R), B * C, L, act, _lib.ptr(dx), st), "affine_act_bwd_apply")
        # d norm.weight, d norm.bias; for a voxel grid also the sum of dx per channel = the bias gradient of the Conv3d in front
        dgw, dgb, dxs = _param_grads(pw, ctx.needs_input_grad[1], ctx.needs_input_grad[2], dx is not None and x.dim() == 5)
        if dxs is not None:
            tag_channel_sum(dx, dxs)
        def back(g, shape):   # the gradient of a broadcast factor: summed over exactly the dimensions it was spread over
            shape = tuple(int(d) for d in shape)
            core = shape
            while len(core) > 2 and core[-1] == 1:
                core = core[:-1]
            return g.sum_to_size(core if core else (1,)).reshape(shape)
        dfac = back(dfac, f_shape) if has_f and ctx.needs_input_grad[3] else None
        dbias = back(dbias, b_shape) if has_b and ctx.needs_input_grad[4] else None
        return dx, dgw, dgb, dfac, dbias, None, None, None, None


class _AdaGNActMax(torch.autograd.Function):
    """max over the last (neighbour) axis of act(GroupNorm(x) * factor + bias) for x [B, C, M, U] -> [B, C, M]: _AdaGNAct with
    the set-abstraction pooling (reference pvcnn2_ada.py:375-377) folded in.  Forward: row sums + one pass that writes [B, C, M]
    only; backward: one reduction + one pass writing dx -- the activated [B, C, M, U] tensor and its (one-hot per group) gradient
    are never materialised (csrc/norm_train.hip, round 6)."""

    @staticmethod
    def forward(ctx, x, gw, gb, factor, bias, groups, eps, act):
        lib = _lib.load()
        x = x.contiguous()
        B, C, M, U = x.shape
        L = M * U
        st = _lib.stream_ptr(x.device)
        dev = x.device
        stats = torch.empty(B * C, 2, device=dev, dtype=torch.float64)
        _lib.check(lib.lion_row_stats64(_lib.ptr(x), B * C, L, _lib.ptr(stats), st), "row_stats64")
        A, Bs, mean, rstd = (torch.empty(B, C, device=dev, dtype=torch.float32) for _ in range(4))
        gwc, gbc = gw.detach().float().contiguous(), gb.detach().float().contiguous()
        f, fs = _rowview(factor.detach(), B, C) if factor is not None else (None, 0)
        bb, bs = _rowview(bias.detach(), B, C) if bias is not None else (None, 0)
        _lib.check(lib.lion_gn_train_fold64(_lib.ptr(stats), _lib.ptr(gwc), _lib.ptr(gbc), _lib.ptr(f), fs, _lib.ptr(bb), bs,
                                            B, C, groups, L, eps, _lib.ptr(A), _lib.ptr(Bs), _lib.ptr(mean), _lib.ptr(rstd),
                                            st), "gn_train_fold64")
        y = torch.empty(B, C, M, device=dev, dtype=torch.float32)
        _lib.check(lib.lion_affine_act_max(_lib.ptr(x), _lib.ptr(A), _lib.ptr(Bs), B * C, M, U, int(act), _lib.ptr(y), st),
                   "affine_act_max")
        ctx.save_for_backward(x, A, Bs, mean, rstd, gwc, gbc, f if f is not None else x.new_empty(0))
        ctx.meta = (groups, int(act), factor is not None, bias is not None, fs,
                    None if factor is None else factor.shape, None if bias is None else bias.shape)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        lib = _lib.load()
        x, A, Bs, mean, rstd, gwc, gbc, f = ctx.saved_tensors
        groups, act, has_f, has_b, fs, f_shape, b_shape = ctx.meta
        gy = gy.contiguous()
        B, C, M, U = x.shape
        L = M * U
        st = _lib.stream_ptr(x.device)
        dev = x.device
        S = torch.empty(B * C, 2, device=dev, dtype=torch.float32)
        _lib.check(lib.lion_affine_act_max_bwd_stats(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(A), _lib.ptr(Bs), B * C, M, U, act,
                                                     _lib.ptr(S), st), "affine_act_max_bwd_stats")
        Q, R = torch.empty(B, C, device=dev, dtype=torch.float32), torch.empty(B, C, device=dev, dtype=torch.float32)
        dfac, dbias, dstride = _affine_grad_buffers(B, C, has_f, has_b, dev)
        pw = torch.empty(B, C, 3, device=dev, dtype=torch.float32)
        _lib.check(lib.lion_gn_train_bwd_fold(_lib.ptr(S), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gwc), _lib.ptr(gbc),
                                              _lib.ptr(f) if has_f else None, fs, B, C, groups, L, _lib.ptr(Q), _lib.ptr(R),
                                              _lib.ptr(dfac), _lib.ptr(dbias), dstride, _lib.ptr(pw), None, None, None, None, None,
                                              st), "gn_train_bwd_fold")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(lib.lion_affine_act_max_bwd_apply(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(A), _lib.ptr(Bs), _lib.ptr(Q),
                                                         _lib.ptr(R), B * C, M, U, act, _lib.ptr(dx), st),
                       "affine_act_max_bwd_apply")
        dgw, dgb, _ = _param_grads(pw, ctx.needs_input_grad[1], ctx.needs_input_grad[2])   # d norm.weight, d norm.bias

        def back(g, shape):
            shape = tuple(int(d) for d in shape)
            core = shape
            while len(core) > 2 and core[-1] == 1:
                core = core[:-1]
            return g.sum_to_size(core if core else (1,)).reshape(shape)
        dfac = back(dfac, f_shape) if has_f and ctx.needs_input_grad[3] else None
        dbias = back(dbias, b_shape) if has_b and ctx.needs_input_grad[4] else None
        return dx, dgw, dgb, dfac, dbias, None, None, None


def adagn_act_max_usable(x, need_grad=True) -> bool:
    return usable(x, need_grad) and x.dim() == 4 and x.shape[3] in (8, 16, 32, 64) and x.shape[0] * x.shape[1] <= 65535


def adagn_act_max(x, norm, factor=None, bias=None, act=True):
    """max_u act(GroupNorm(x) * factor + bias) over the last axis of x [B, C, M, U] (see _AdaGNActMax)."""
    return _AdaGNActMax.apply(x, norm.weight, norm.bias, factor, bias, int(norm.num_groups), float(norm.eps), bool(act))


def adagn_act(x, norm, factor=None, bias=None, act=True, dropout_p=0.0):
    """dropout(act(GroupNorm(x) * factor + bias), dropout_p); factor / bias [B, C] (or broadcastable views of it) or None."""
    return _AdaGNAct.apply(x, norm.weight, norm.bias, factor, bias, int(norm.num_groups), float(norm.eps), bool(act),
                           float(dropout_p))


def fusable_dropout(layers, i):
    """the keep-side probability p of an nn.Dropout at layers[i] that the activation pass in front of it can take over (0.0: there
    is none / it is the identity now / p = 1, which stays with the module), and how many layers that consumes"""
    if i < len(layers) and type(layers[i]) is torch.nn.Dropout:
        d = layers[i]
        if not d.training or d.p <= 0.0:
            return 0.0, 1     # identity: skipped
        if d.p < 1.0 and DROPOUT_FUSED:
            return float(d.p), 1
    return 0.0, 0


class _AdaGNSE(torch.autograd.Function):
    """SE3d(AdaGN(x)) with no activation between the two -- the tail of every PVConv's voxel branch (reference pvcnn2_ada.py:211-226:
    Conv3d -> AdaGN -> SE3d) -- as ONE differentiable op.  u = A x + Bs (the AdaGN) and y = g u (the gate, g from the channel means
    of u) are both affine per (sample, channel): forward = row sums of x, the [B, C] algebra (GroupNorm fold, mean(u) = A mean(x) + Bs,
    the gate), ONE pass y = (g A) x + g Bs; backward = ONE reduction {sum gy, sum gy x}, the [B, C] algebra (the gate's gradient is
    A S2 + Bs S1; du = g gy + Qse and its row sums follow in closed form; GroupNorm backward), ONE pass dx = (A g) gy + Q + R x.
    The two separate ops (adagn_act(act=False) -> se3d) make 6 passes over the grid forward and 10 backward."""

    @staticmethod
    def forward(ctx, x, gw, gb, factor, bias, w1, w2, groups, eps):
        lib = _lib.load()
        x = x.contiguous()
        B, C = x.shape[:2]
        Cr = w1.shape[0]
        L = x[0, 0].numel()
        st = _lib.stream_ptr(x.device)
        dev = x.device
        stats = torch.empty(B * C, 2, device=dev, dtype=torch.float64)
        _lib.check(lib.lion_row_stats64(_lib.ptr(x), B * C, L, _lib.ptr(stats), st), "row_stats64")
        A, Bs, mean, rstd = (torch.empty(B, C, device=dev, dtype=torch.float32) for _ in range(4))
        gwc, gbc = gw.detach().float().contiguous(), gb.detach().float().contiguous()
        f, fs = _rowview(factor.detach(), B, C) if factor is not None else (None, 0)
        bb, bs = _rowview(bias.detach(), B, C) if bias is not None else (None, 0)
        _lib.check(lib.lion_gn_train_fold64(_lib.ptr(stats), _lib.ptr(gwc), _lib.ptr(gbc), _lib.ptr(f), fs, _lib.ptr(bb), bs,
                                            B, C, groups, L, eps, _lib.ptr(A), _lib.ptr(Bs), _lib.ptr(mean), _lib.ptr(rstd),
                                            st), "gn_train_fold64")
        w1c, w2c = w1.detach().float().contiguous(), w2.detach().float().contiguous()
        um, g, A2, B2 = (torch.empty(B, C, device=dev, dtype=torch.float32) for _ in range(4))
        h = torch.empty(B, Cr, device=dev, dtype=torch.float32)
        _lib.check(lib.lion_gn_se_gate_fwd(_lib.ptr(stats), _lib.ptr(A), _lib.ptr(Bs), _lib.ptr(w1c), _lib.ptr(w2c), B, C, Cr, L,
                                           _lib.ptr(um), _lib.ptr(h), _lib.ptr(g), _lib.ptr(A2), _lib.ptr(B2), st), "gn_se_gate_fwd")
        y = torch.empty_like(x)
        _lib.check(lib.lion_affine_act(_lib.ptr(x), _lib.ptr(A2), _lib.ptr(B2), B * C, L, 0, _lib.ptr(y), st), "affine_act")
        ctx.save_for_backward(x, A, Bs, mean, rstd, gwc, gbc, f if f is not None else x.new_empty(0), stats, um, h, g, w1c, w2c)
        ctx.meta = (groups, factor is not None, bias is not None, fs,
                    None if factor is None else factor.shape, None if bias is None else bias.shape)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        lib = _lib.load()
        x, A, Bs, mean, rstd, gwc, gbc, f, stats, um, h, g, w1, w2 = ctx.saved_tensors
        groups, has_f, has_b, fs, f_shape, b_shape = ctx.meta
        gy = gy.contiguous()
        B, C = x.shape[:2]
        Cr = w1.shape[0]
        L = x[0, 0].numel()
        st = _lib.stream_ptr(x.device)
        dev = x.device
        S = torch.empty(B * C, 2, device=dev, dtype=torch.float32)
        _lib.check(lib.lion_affine_act_bwd_stats(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(A), _lib.ptr(Bs), B * C, L, 0, _lib.ptr(S), st),
                   "affine_act_bwd_stats")                      # act 0: S = {sum gy, sum gy x}
        Sp = torch.empty(B * C, 2, device=dev, dtype=torch.float32)
        dpre2, Qse, Q, R, Aout = (torch.empty(B, C, device=dev, dtype=torch.float32) for _ in range(5))
        dpre1 = torch.empty(B, Cr, device=dev, dtype=torch.float32)
        dw1, dw2 = torch.empty_like(w1), torch.empty_like(w2)
        _lib.check(lib.lion_gn_se_gate_bwd(_lib.ptr(S), _lib.ptr(stats), _lib.ptr(A), _lib.ptr(Bs), _lib.ptr(g), _lib.ptr(h),
                                           _lib.ptr(um), _lib.ptr(w1), _lib.ptr(w2), B, C, Cr, L, _lib.ptr(Sp), _lib.ptr(dpre2),
                                           _lib.ptr(dpre1), _lib.ptr(Qse), _lib.ptr(dw1), _lib.ptr(dw2), st), "gn_se_gate_bwd")
        dfac, dbias, dstride = _affine_grad_buffers(B, C, has_f, has_b, dev)
        pw = torch.empty(B, C, 3, device=dev, dtype=torch.float32)
        _lib.check(lib.lion_gn_train_bwd_fold(_lib.ptr(Sp), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gwc), _lib.ptr(gbc),
                                              _lib.ptr(f) if has_f else None, fs, B, C, groups, L, _lib.ptr(Q), _lib.ptr(R),
                                              _lib.ptr(dfac), _lib.ptr(dbias), dstride, _lib.ptr(pw), _lib.ptr(A), _lib.ptr(stats),
                                              _lib.ptr(g), _lib.ptr(Qse), _lib.ptr(Aout), st), "gn_train_bwd_fold")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            # (act 0: the kernel's Bs argument does not enter)
            _lib.check(lib.lion_affine_act_bwd_apply(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(Aout), _lib.ptr(Bs), _lib.ptr(Q),
                                                     _lib.ptr(R), B * C, L, 0, _lib.ptr(dx), st), "affine_act_bwd_apply")
        dgw, dgb, dxs = _param_grads(pw, ctx.needs_input_grad[1], ctx.needs_input_grad[2], dx is not None and x.dim() == 5)
        if dxs is not None:
            tag_channel_sum(dx, dxs)

        def back(g_, shape):
            shape = tuple(int(d) for d in shape)
            core = shape
            while len(core) > 2 and core[-1] == 1:
                core = core[:-1]
            return g_.sum_to_size(core if core else (1,)).reshape(shape)
        dfac = back(dfac, f_shape) if has_f and ctx.needs_input_grad[3] else None
        dbias = back(dbias, b_shape) if has_b and ctx.needs_input_grad[4] else None
        return (dx, dgw, dgb, dfac, dbias, (dw1 if ctx.needs_input_grad[5] else None), (dw2 if ctx.needs_input_grad[6] else None),
                None, None)


class _AdaGNSEDevox(torch.autograd.Function):
    """trilinear_devoxelize(SE3d(AdaGN(x)), coords) -- the whole tail of a PVConv's voxel branch behind its second convolution
    (reference pvcnn2_ada.py:211-233) -- as ONE differentiable op.  All three are linear in x given the [B, C] scalars, and the
    devoxelisation's weights sum to one per point, so devox(g (A x + Bs)) = (g A) devox(x) + g Bs: the forward devoxelises x itself
    and never writes a gated grid; the backward gets its two row sums from the POINTS (sum_p gpt ws, sum_p gpt devox(x): no pass over
    the grid) and writes dx = A' scatter(gpt) + Q + R x inside the scatter's own pass.  Dense passes over [B, C, r^3]: 1 forward
    (the GroupNorm row sums), 2 backward (read x, write dx); _AdaGNSE followed by the devoxelisation makes 3 and 6."""

    @staticmethod
    def forward(ctx, x, gw, gb, factor, bias, w1, w2, coords, groups, eps, r):
        from .functional import backend as _bk
        lib = _lib.load()
        x = x.contiguous()
        B, C = x.shape[:2]
        Cr = w1.shape[0]
        L = x[0, 0].numel()
        st = _lib.stream_ptr(x.device)
        dev = x.device
        stats = torch.empty(B * C, 2, device=dev, dtype=torch.float64)
        _lib.check(lib.lion_row_stats64(_lib.ptr(x), B * C, L, _lib.ptr(stats), st), "row_stats64")
        A, Bs, mean, rstd = (torch.empty(B, C, device=dev, dtype=torch.float32) for _ in range(4))
        gwc, gbc = gw.detach().float().contiguous(), gb.detach().float().contiguous()
        f, fs = _rowview(factor.detach(), B, C) if factor is not None else (None, 0)
        bb, bs = _rowview(bias.detach(), B, C) if bias is not None else (None, 0)
        _lib.check(lib.lion_gn_train_fold64(_lib.ptr(stats), _lib.ptr(gwc), _lib.ptr(gbc), _lib.ptr(f), fs, _lib.ptr(bb), bs,
                                            B, C, groups, L, eps, _lib.ptr(A), _lib.ptr(Bs), _lib.ptr(mean), _lib.ptr(rstd),
                                            st), "gn_train_fold64")
        w1c, w2c = w1.detach().float().contiguous(), w2.detach().float().contiguous()
        um, g, A2, B2 = (torch.empty(B, C, device=dev, dtype=torch.float32) for _ in range(4))
        h = torch.empty(B, Cr, device=dev, dtype=torch.float32)
        _lib.check(lib.lion_gn_se_gate_fwd(_lib.ptr(stats), _lib.ptr(A), _lib.ptr(Bs), _lib.ptr(w1c), _lib.ptr(w2c), B, C, Cr, L,
                                           _lib.ptr(um), _lib.ptr(h), _lib.ptr(g), _lib.ptr(A2), _lib.ptr(B2), st), "gn_se_gate_fwd")
        dv, inds, wgts = _bk._backend.trilinear_devoxelize_forward(int(r), True, coords[:, :3].contiguous(), x.flatten(2))
        out = torch.addcmul(B2.unsqueeze(-1), dv, A2.unsqueeze(-1))          # [B, C, N]: (g A) devox(x) + g Bs
        wsum = wgts.sum(1)                                                     # [B, N]: the 8 corner weights of a point
        ctx.save_for_backward(x, A, Bs, mean, rstd, gwc, gbc, f if f is not None else x.new_empty(0), stats, um, h, g, w1c, w2c,
                              dv, inds, wgts, wsum)
        ctx.meta = (groups, factor is not None, bias is not None, fs,
                    None if factor is None else factor.shape, None if bias is None else bias.shape)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gpt):
        lib = _lib.load()
        x, A, Bs, mean, rstd, gwc, gbc, f, stats, um, h, g, w1, w2, dv, inds, wgts, wsum = ctx.saved_tensors
        groups, has_f, has_b, fs, f_shape, b_shape = ctx.meta
        gpt = gpt.contiguous()
        B, C = x.shape[:2]
        Cr = w1.shape[0]
        L = x[0, 0].numel()
        N = gpt.shape[2]
        st = _lib.stream_ptr(x.device)
        dev = x.device
        S = torch.empty(B * C, 2, device=dev, dtype=torch.float32)
        _lib.check(lib.lion_rows_dot2(_lib.ptr(gpt), _lib.ptr(dv), _lib.ptr(wsum), B, C, N, _lib.ptr(S), st), "rows_dot2")
        Sp = torch.empty(B * C, 2, device=dev, dtype=torch.float32)
        dpre2, Qse, Q, R, Aout = (torch.empty(B, C, device=dev, dtype=torch.float32) for _ in range(5))
        dpre1 = torch.empty(B, Cr, device=dev, dtype=torch.float32)
        dw1, dw2 = torch.empty_like(w1), torch.empty_like(w2)
        _lib.check(lib.lion_gn_se_gate_bwd(_lib.ptr(S), _lib.ptr(stats), _lib.ptr(A), _lib.ptr(Bs), _lib.ptr(g), _lib.ptr(h),
                                           _lib.ptr(um), _lib.ptr(w1), _lib.ptr(w2), B, C, Cr, L, _lib.ptr(Sp), _lib.ptr(dpre2),
                                           _lib.ptr(dpre1), _lib.ptr(Qse), _lib.ptr(dw1), _lib.ptr(dw2), st), "gn_se_gate_bwd")
        dfac, dbias, dstride = _affine_grad_buffers(B, C, has_f, has_b, dev)
        pw = torch.empty(B, C, 3, device=dev, dtype=torch.float32)
        _lib.check(lib.lion_gn_train_bwd_fold(_lib.ptr(Sp), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gwc), _lib.ptr(gbc),
                                              _lib.ptr(f) if has_f else None, fs, B, C, groups, L, _lib.ptr(Q), _lib.ptr(R),
                                              _lib.ptr(dfac), _lib.ptr(dbias), dstride, _lib.ptr(pw), _lib.ptr(A), _lib.ptr(stats),
                                              _lib.ptr(g), _lib.ptr(Qse), _lib.ptr(Aout), st), "gn_train_bwd_fold")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(lib.lion_trilinear_devoxelize_backward_affine(_lib.ptr(gpt), _lib.ptr(inds), _lib.ptr(wgts), _lib.ptr(x),
                                                                     _lib.ptr(Aout), _lib.ptr(Q), _lib.ptr(R), B, C, N, L,
                                                                     _lib.ptr(dx), st), "trilinear_devoxelize_backward_affine")
        dgw, dgb, dxs = _param_grads(pw, ctx.needs_input_grad[1], ctx.needs_input_grad[2], dx is not None and x.dim() == 5)
        if dxs is not None:
            tag_channel_sum(dx, dxs)

        def back(g_, shape):
            shape = tuple(int(d) for d in shape)
            core = shape
            while len(core) > 2 and core[-1] == 1:
                core = core[:-1]
            return g_.sum_to_size(core if core else (1,)).reshape(shape)
        dfac = back(dfac, f_shape) if has_f and ctx.needs_input_grad[3] else None
        dbias = back(dbias, b_shape) if has_b and ctx.needs_input_grad[4] else None
        return (dx, dgw, dgb, dfac, dbias, (dw1 if ctx.needs_input_grad[5] else None), (dw2 if ctx.needs_input_grad[6] else None),
                None, None, None, None)


def adagn_se_devox_usable(x, se) -> bool:
    r3 = x[0, 0].numel() if x.dim() == 5 else 0
    return (DEVOX_FUSED and x.dim() == 5 and se3d_trainable(se, x) and torch.is_grad_enabled() and 0 < r3 * 4 <= 128 * 1024
            and r3 % 8 == 0)


def adagn_se_devox(x, norm, factor, bias, se, coords, r):
    """trilinear_devoxelize(SE3d(GroupNorm(x) * factor + bias), coords, r) as one op (see _AdaGNSEDevox)"""
    return _AdaGNSEDevox.apply(x, norm.weight, norm.bias, factor, bias, se.fc[0].weight, se.fc[2].weight, coords.detach(),
                               int(norm.num_groups), float(norm.eps), int(r))


def adagn_se(x, norm, factor, bias, se):
    """SE3d(GroupNorm(x) * factor + bias) as one op (see _AdaGNSE); `se` passes se3d_trainable."""
    return _AdaGNSE.apply(x, norm.weight, norm.bias, factor, bias, se.fc[0].weight, se.fc[2].weight, int(norm.num_groups),
                          float(norm.eps))


class _SE3d(torch.autograd.Function):
    """y = x * sigmoid(W2 relu(W1 mean_voxels(x))) -- SE3d (reference models/pvcnn2_ada.py:27-41) as ONE differentiable op:
    forward = row sums + one scaling pass; backward = one pass for sum_n gy x (the gate's gradient) + one fused apply
    dx = gate gy + d mean / L; the [B, C] algebra of the two bias-free Linear layers is csrc/norm_train.hip's se_gate kernels
    (one launch forward, two backward).  ATen runs the reference expression as three chained mean() reductions, a broadcast multiply, and in backward
    two more broadcast multiplies, an expand and three reductions over the 268-MB grid."""

    @staticmethod
    def forward(ctx, x, w1, w2):
        lib = _lib.load()
        x = x.contiguous()
        B, C = x.shape[:2]
        Cr = w1.shape[0]
        L = x[0, 0].numel()
        st = _lib.stream_ptr(x.device)
        dev = x.device
        stats = torch.empty(B * C, 2, device=dev, dtype=torch.float32)
        _lib.check(lib.lion_row_stats(_lib.ptr(x), B * C, L, _lib.ptr(stats), st), "row_stats")
        w1c, w2c = w1.detach().float().contiguous(), w2.detach().float().contiguous()
        mean, g, zero = (torch.empty(B, C, device=dev, dtype=torch.float32) for _ in range(3))
        h = torch.empty(B, Cr, device=dev, dtype=torch.float32)
        _lib.check(lib.lion_se_gate_fwd(_lib.ptr(stats), _lib.ptr(w1c), _lib.ptr(w2c), B, C, Cr, L, _lib.ptr(mean), _lib.ptr(h),
                                        _lib.ptr(g), _lib.ptr(zero), st), "se_gate_fwd")
        y = torch.empty_like(x)
        _lib.check(lib.lion_affine_act(_lib.ptr(x), _lib.ptr(g), _lib.ptr(zero), B * C, L, 0, _lib.ptr(y), st), "affine_act")
        ctx.save_for_backward(x, w1c, w2c, mean, h, g, zero)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        lib = _lib.load()
        x, w1, w2, mean, h, g, zero = ctx.saved_tensors
        gy = gy.contiguous()
        B, C = x.shape[:2]
        Cr = w1.shape[0]
        L = x[0, 0].numel()
        st = _lib.stream_ptr(x.device)
        dev = x.device
        S = torch.empty(B * C, 2, device=dev, dtype=torch.float32)
        _lib.check(lib.lion_affine_act_bwd_stats(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(g), _lib.ptr(zero), B * C, L, 0,
                                                 _lib.ptr(S), st), "affine_act_bwd_stats")   # S[:, 1] = sum over the voxels of gy * x
        dpre2, Q = torch.empty(B, C, device=dev, dtype=torch.float32), torch.empty(B, C, device=dev, dtype=torch.float32)
        dpre1 = torch.empty(B, Cr, device=dev, dtype=torch.float32)
        dw1, dw2 = torch.empty_like(w1), torch.empty_like(w2)
        _lib.check(lib.lion_se_gate_bwd(_lib.ptr(S), _lib.ptr(g), _lib.ptr(h), _lib.ptr(mean), _lib.ptr(w1), _lib.ptr(w2), B, C, Cr,
                                        L, _lib.ptr(dpre2), _lib.ptr(dpre1), _lib.ptr(Q), _lib.ptr(dw1), _lib.ptr(dw2), st),
                   "se_gate_bwd")
        dx = None
        if ctx.needs_input_grad[0]:   # Q: d loss / d x through the mean, the same for every voxel of a channel
            dx = torch.empty_like(x)
            _lib.check(lib.lion_affine_act_bwd_apply(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(g), _lib.ptr(zero), _lib.ptr(Q),
                                                     _lib.ptr(zero), B * C, L, 0, _lib.ptr(dx), st), "affine_act_bwd_apply")
        return dx, (dw1 if ctx.needs_input_grad[1] else None), (dw2 if ctx.needs_input_grad[2] else None)


def se3d_trainable(se, x, need_grad=True) -> bool:
    fc = getattr(se, "fc", None)
    return (usable(x, need_grad) and fc is not None and len(fc) == 4 and isinstance(fc[0], torch.nn.Linear) and fc[0].bias is None
            and isinstance(fc[2], torch.nn.Linear) and fc[2].bias is None
            and fc[0].in_features <= 1024 and fc[0].out_features <= 128 and fc[2].weight.dtype == torch.float32)


def se3d(se, x):
    return _SE3d.apply(x, se.fc[0].weight, se.fc[2].weight)


class _PwConv(torch.autograd.Function):
    """kernel-size-1 convolution, every direction on the library's kernels: forward and the data gradient on the 1x1
    MFMA kernels (the data gradient = the same kernel on the transposed matrix), the weight gradient on
    csrc/pwconv_wgrad.hip, the bias gradient as one streaming row-sum pass."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from . import fused_ops
        w2d = weight.reshape(weight.shape[0], -1)
        y = fused_ops.pwconv_raw(x, w2d, bias, cached_param=weight)
        if y is None:
            raise RuntimeError("pwconv: unsupported shape (call sites check fused_ops.pwconv_raw support first)")
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        from . import fused_ops
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        w2d = weight.detach().reshape(weight.shape[0], -1)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = fused_ops.pwconv_raw(gy, w2d, cached_param=weight, transposed=True)
            if gx is None:
                gx = torch.matmul(w2d.t(), gy.flatten(2)).reshape(x.shape)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if want_b:   # the bias gradient rides on the weight gradient's pass over gy
                gw, gb = fused_ops.pwconv_wgrad(x, gy, want_bias=True)
            else:
                gw = fused_ops.pwconv_wgrad(x, gy)
            gw = gw.reshape(weight.shape)
        elif want_b:
            gb = fused_ops.row_stats(gy)[:, 0].reshape(gy.shape[0], gy.shape[1]).sum(0)
        return gx, gw, gb


def pwconv_trainable(conv, x) -> bool:
    from . import fused_ops
    if not (PWCONV and usable(x) and conv.groups == 1 and all(k == 1 for k in conv.kernel_size) and all(s == 1 for s in conv.stride)
            and all(p == 0 for p in conv.padding)):
        return False
    lib = _lib.load()
    L = x[0, 0].numel()
    return (x.shape[0] * L >= PWCONV_MIN_COLS and lib.lion_pwconv_stat_tiles(conv.out_channels, conv.in_channels, L) > 0
            and (x.data_ptr() & 15) == 0)


def pwconv(conv, x):
    return _PwConv.apply(x, conv.weight, conv.bias)


ATTENTION = os.environ.get("LION_TRAIN_ATTENTION", "1") != "0"   # LinearAttention core (forward + backward) on own kernels


class _LinAttnCore(torch.autograd.Function):
    """softmax_N(k), ctx = k v^T, out = ctx^T q of LinearAttention (reference models/pvcnn2_ada.py:62-68) between its two
    1x1 convolutions: csrc/attention.hip forward and backward (one workgroup per (batch, head) each) instead of a
    rearrange copy, a softmax, two einsums and their five autograd kernels."""

    @staticmethod
    def forward(ctx, qkv, heads):
        lib = _lib.load()
        qkv = qkv.contiguous()
        b, n = qkv.shape[0], qkv.shape[2]
        out = torch.empty((b, heads * 32, n), device=qkv.device, dtype=torch.float32)
        _lib.check(lib.lion_linear_attention_core(_lib.ptr(qkv), b, heads, 32, n, _lib.ptr(out),
                                                  _lib.stream_ptr(qkv.device)), "linear_attention_core")
        ctx.save_for_backward(qkv)
        ctx.heads = heads
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        (qkv,) = ctx.saved_tensors
        lib = _lib.load()
        gout = gout.contiguous()
        b, n = qkv.shape[0], qkv.shape[2]
        gqkv = torch.empty_like(qkv)
        _lib.check(lib.lion_linear_attention_core_backward(_lib.ptr(qkv), _lib.ptr(gout), b, ctx.heads, 32, n, _lib.ptr(gqkv),
                                                           _lib.stream_ptr(qkv.device)), "linear_attention_core_backward")
        return gqkv, None


def linear_attention_core(qkv, heads):
    """[B, 3*heads*32, N] -> [B, heads*32, N], differentiable in qkv"""
    return _LinAttnCore.apply(qkv, heads)


def attention_trainable(qkv, heads, dim_head) -> bool:
    return (ENABLED and ATTENTION and dim_head == 32 and qkv.is_cuda and qkv.dtype == torch.float32 and qkv.dim() == 3
            and torch.is_grad_enabled() and not torch.is_autocast_enabled())
