"""PVCNN building blocks with adaptive GroupNorm -- mirror of the reference's
``models/pvcnn2_ada.py`` (SE3d :27, LinearAttention :43, Swish :78, BallQuery :86, SharedMLP :120,
Voxelization :166, PVConv :195, PointNetAModule :282, PointNetSAModule :321, PointNetFPModule :388,
builders :416/:448/:520).  Parameter names / shapes are identical, so reference checkpoints load.

Differences that matter on MI355X (all numerically equivalent to fp32 rounding):
  * every point-voxel operator is a HIP kernel from ``lion_amd.functional``;
  * ``Voxelization.forward`` is two launches (fused normalise + index + mean-pool) instead of ~12;
  * the time embedding that the reference broadcasts to [B,64,N] and concatenates before every
    block is kept as given (shape contract of the callers), nothing is re-laid-out on the host.
"""
import contextlib
import functools
import os

import torch
import torch.nn as nn
from einops import rearrange

from .. import functional as F
from ..conv_ops import conv3d_module
from .. import fused_ops

FUSE_INFERENCE = True  # module-level switch (tests compare the fused and the layer-by-layer paths)
# conv1 of a PVConv reads the voxelised grid: skip (exactly) the tiles whose halo holds no point
SPARSE_CONV1 = True
# run the point branch of a PVConv on a second stream, concurrently with its voxel branch (inference).  Off by default:
# as a parallel branch of the captured step it costs 0.5 ms per step (lion_amd/geometry.py has the measurements)
OVERLAP_POINT_BRANCH = os.environ.get("LION_OVERLAP_POINT_BRANCH", "0") != "0"
_POINT_STREAMS = {}


# Index plans of the voxelisations of one forward pass (inference): every PVConv of a stage voxelises the SAME coordinates
# at the SAME resolution (reference :235-243), and the feature-propagation stages come back to the set-abstraction stages'
# clouds -- 4 distinct (cloud, r) pairs for 14 voxelisations in the released denoiser.  While a PVCNN2Unet forward is in
# flight this holds {(data_ptr, shape, r, normalize, eps): (coords tensor, plan)}; the coordinates tensor is kept alive by
# the entry, so its address cannot be handed to another tensor while the entry exists.
_VOX_PLANS = None
VOX_PLAN = os.environ.get("LION_VOX_PLAN", "1") != "0"   # A/B switch: 0 = every voxelisation recomputes its indices
OCC_PLAN = os.environ.get("LION_OCC_PLAN", "1") != "0"       # A/B switch: 0 = every PVConv recomputes its tile occupancy
# the fused voxel branch is the ONLY reader of its two convolutions' outputs (conv1 -> conv2's halos, conv2 -> the 8 voxels
# around each point): empty tiles without a reader are not written (round 5; 0 = every voxel of both outputs is written)
SKIP_UNREAD = os.environ.get("LION_CONV_SKIP_UNREAD", "1") != "0"
SKIP_UNREAD_LEVEL2 = os.environ.get("LION_CONV_SKIP_UNREAD_LEVEL2", "1") != "0"   # A/B: 0 = conv1 still stores the empty tiles conv2's halos touch
OCC_CLONE = os.environ.get("LION_OCC_CLONE", "0") != "0"     # A/B switch: 1 = a copy of the occupancy buffers per convolution (libraries before round 5)
DEVOX_PLAN = os.environ.get("LION_DEVOX_PLAN", "1") != "0"   # A/B switch: 0 = every r = 32 devoxelisation redoes its per-cloud setup


@contextlib.contextmanager
def voxel_plans():
    global _VOX_PLANS
    prev, _VOX_PLANS = _VOX_PLANS, ({} if VOX_PLAN else None)
    try:
        yield
    finally:
        _VOX_PLANS = prev


def _occupancy(counts, r, cout, b, level=None):
    """tile occupancy + work lists (fused_ops.conv3d_occupancy) of the count grid of a (cloud, r) pair: the same for every
    PVConv that voxelises this pair (the sparse tile plan depends on r only), so while a forward's plans are alive it is
    computed once and every convolution of the pair works from the SAME buffers: a sparse convolution re-arms its work
    queue when its last workgroup leaves (round 5; until then every use took a copy -- 7 copy launches per step).
    level: the consumer-aware level of the buffers (0 every voxel written, 1 unread empty tiles skipped, 2 the delta
    convolution is the split kernel: conv1 stores occupied tiles only); None = 1 if SKIP_UNREAD else 0."""
    if level is None:
        level = 1 if SKIP_UNREAD else 0
    if _VOX_PLANS is None or not OCC_PLAN:
        return fused_ops.conv3d_occupancy(counts, r, cout, b, consumer_aware=level)
    key = ("occ", counts.data_ptr(), tuple(counts.shape), int(r), int(b), int(level))
    hit = _VOX_PLANS.get(key)
    if hit is None:
        o1, o2 = fused_ops.conv3d_occupancy(counts, r, cout, b, consumer_aware=level)
        hit = (counts, (o1, o2))   # counts stays alive with the entry
        _VOX_PLANS[key] = hit
    if OCC_CLONE:   # A/B against a library from before round 5, whose convolutions leave the queue counter consumed
        return hit[1][0].clone(), hit[1][1].clone()
    return hit[1]


def _devox_plan(voxel_coords, r):
    """the devoxelisation plan of (voxel coordinates, r) of the forward in flight (kept beside the voxel index plans: the
    same four (cloud, r) pairs, the r = 32 pair devoxelised four times), or None."""
    if _VOX_PLANS is None or not DEVOX_PLAN or r != 32 or voxel_coords.shape[2] > 2048 or not voxel_coords.is_contiguous():
        return None
    key = ("devox", voxel_coords.data_ptr(), tuple(voxel_coords.shape), int(r))
    hit = _VOX_PLANS.get(key)
    if hit is None:
        plan = fused_ops.devoxelize_plan(voxel_coords, r)
        hit = (voxel_coords, plan)      # the coordinates tensor stays alive with the entry
        _VOX_PLANS[key] = hit
    return hit[1]


def _point_stream(device):
    key = (device.type, device.index)
    if key not in _POINT_STREAMS:
        _POINT_STREAMS[key] = torch.cuda.Stream(device=device)
    return _POINT_STREAMS[key]
from .adagn import AdaGN


class SE3d(nn.Module):
    """Channel gate from the global mean over the voxel grid (reference :27-41)."""

    def __init__(self, channel, reduction=8):
        super().__init__()
        self.fc = nn.Sequential(
            nn.Linear(channel, channel // reduction, bias=False),
            nn.ReLU(inplace=True),
            nn.Linear(channel // reduction, channel, bias=False),
            nn.Sigmoid())
        self.channel = channel

    def __repr__(self):
        return f"SE({self.channel}, {self.channel})"

    def gate(self, inputs):
        """[B, C] gate; the reference's three chained mean(-1) (:41) are one mean over the grid."""
        return self.fc(inputs.mean(-1).mean(-1).mean(-1))

    def forward(self, inputs):
        return inputs * self.gate(inputs).view(inputs.shape[0], inputs.shape[1], 1, 1, 1)


class LinearAttention(nn.Module):
    """O(N) attention over points (reference :43-71): softmax over N on k, context = k v^T."""

    def __init__(self, dim, heads=4, dim_head=32, verbose=True):
        super().__init__()
        self.heads = heads
        hidden_dim = dim_head * heads
        self.to_qkv = nn.Conv2d(dim, hidden_dim * 3, 1, bias=False)
        self.to_out = nn.Conv2d(hidden_dim, dim, 1)

    def forward(self, x):
        if own_kernels(x) and self.to_qkv.out_channels == 3 * self.heads * 32 and fused_ops.pw_supported(self.to_qkv, x):
            # inference: 3 launches -- to_qkv (MFMA 1x1 conv), the attention core of one workgroup per (batch, head)
            # (csrc/attention.hip), to_out -- instead of conv, rearrange copy, softmax, two bmm, rearrange, conv
            qkv = fused_ops.pwconv_fused(x, self.to_qkv, None, want_stats=False)[0]
            core = fused_ops.linear_attention_core(qkv, self.heads, 32)
            return fused_ops.pwconv_fused(core, self.to_out, None, want_stats=False)[0]
        x = x.unsqueeze(-1)  # [B, C, N, 1]
        b, c, h, w = x.shape
        qkv = self.to_qkv(x)
        from .. import train_ops
        if w == 1 and self.to_qkv.out_channels == 3 * self.heads * 32 and train_ops.attention_trainable(qkv.squeeze(-1), self.heads, 32):
            # training: the core (softmax over N, ctx, out) and its gradient on csrc/attention.hip
            out = train_ops.linear_attention_core(qkv.squeeze(-1), self.heads)
            return self.to_out(out.unsqueeze(-1)).squeeze(-1)
        q, k, v = rearrange(qkv, 'b (qkv heads c) h w -> qkv b heads c (h w)', heads=self.heads, qkv=3)
        k = k.softmax(dim=-1)
        context = torch.einsum('bhdn,bhen->bhde', k, v)
        out = torch.einsum('bhde,bhdn->bhen', context, q)
        out = rearrange(out, 'b heads c (h w) -> b (heads c) h w', heads=self.heads, h=h, w=w)
        return self.to_out(out).squeeze(-1)


def swish(input):
    return input * torch.sigmoid(input)


class Swish(nn.Module):
    def forward(self, input):
        return swish(input)


class BallQuery(nn.Module):
    """ball query + grouping of coordinates (relative to the centre) and features (reference :86-118)."""

    def __init__(self, radius, num_neighbors, include_coordinates=True):
        super().__init__()
        self.radius = radius
        self.num_neighbors = num_neighbors
        self.include_coordinates = include_coordinates

    def forward(self, points_coords, centers_coords, points_features=None):
        with torch.autocast("cuda", enabled=False):
            points_coords = points_coords.float().contiguous()
            centers_coords = centers_coords.float().contiguous()
            neighbor_indices = F.ball_query(centers_coords, points_coords, self.radius, self.num_neighbors)
            if own_kernels(points_coords) and self.include_coordinates and points_coords.shape[1] == 3 \
                    and (points_features is None or points_features.dtype == torch.float32):
                # inference: relative coordinates and features written straight into one [B, 3 + C, M, U] tensor
                return fused_ops.group_points(points_coords, centers_coords, points_features, neighbor_indices)
            neighbor_coordinates = F.grouping(points_coords, neighbor_indices)
            neighbor_coordinates = neighbor_coordinates - centers_coords.unsqueeze(-1)
            if points_features is None:
                assert self.include_coordinates, 'No Features For Grouping'
                return neighbor_coordinates
            neighbor_features = F.grouping(points_features.float(), neighbor_indices)
            if self.include_coordinates:
                neighbor_features = torch.cat([neighbor_coordinates, neighbor_features], dim=1)
            return neighbor_features

    def extra_repr(self):
        return 'radius={}, num_neighbors={}{}'.format(
            self.radius, self.num_neighbors, ', include coordinates' if self.include_coordinates else '')


def own_kernels(x):
    """inference on the GPU in fp32: every dense layer has a kernel of liblion_hip.so"""
    return (FUSE_INFERENCE and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
            and not torch.is_autocast_enabled())


def linear(lin, x, act=0, slope=0.0):
    """nn.Linear (+ ReLU / LeakyReLU) on [B, K]: one MFMA launch in inference, torch otherwise."""
    if own_kernels(x) and x.dim() == 2:
        return fused_ops.linear_rows(x, lin.weight, lin.bias, act, slope)
    y = lin(x)
    return y if act == 0 else (torch.relu(y) if act == 1 else nn.functional.leaky_relu(y, slope))


def conv1x1(conv, x):
    """kernel-size-1 Conv1d / Conv2d as a (broadcast) matrix product.  Same arithmetic; the point is the backward:
    autograd then differentiates a GEMM (rocBLAS) instead of asking the convolution library for backward-data /
    backward-filter kernels of a 1x1 conv, for which it falls back to naive kernels on an untuned box (5 ms per
    layer, 60 of the 158 ms of a VAE training step).  Inference keeps the library's forward kernel (measured 0.6 ms per
    denoiser step faster than the batched matmul for these shapes)."""
    if own_kernels(x) and fused_ops.pw_supported(conv, x):
        if x[0, 0].numel() == 1:  # [B, C, 1(, 1)]: a Linear over the batch (one column per sample would waste the tile)
            y = fused_ops.linear_rows(x.reshape(x.shape[0], x.shape[1]), conv.weight.flatten(1), conv.bias)
            return y.reshape(x.shape[0], conv.out_channels, *x.shape[2:])
        return fused_ops.pwconv_fused(x, conv, None, want_stats=False)[0]  # fp32 MFMA GEMM, no layout transposes
    from .. import train_ops
    if train_ops.pwconv_trainable(conv, x):  # training: forward, data / weight / bias gradients on own kernels
        return train_ops.pwconv(conv, x)
    from .. import _fallback
    if (x.is_cuda and torch.is_grad_enabled() and all(k == 1 for k in conv.kernel_size) and all(s == 1 for s in conv.stride)
            and all(p == 0 for p in conv.padding) and conv.groups == 1):
        _fallback.note("pvcnn2_ada.conv1x1 (training matmul)", f"{tuple(conv.weight.shape)} on {tuple(x.shape)}")
        y = torch.matmul(conv.weight.flatten(1), x.flatten(2))
        if conv.bias is not None:
            y = y + conv.bias[:, None]
        return y.reshape(x.shape[0], conv.out_channels, *x.shape[2:])
    _fallback.note("pvcnn2_ada.conv1x1 (library conv)", f"{tuple(conv.weight.shape)} on {tuple(x.shape)} {x.dtype} {x.device.type}")
    return type(conv).forward(conv, x)  # the class's own forward (conv.forward may be routed here)


def route_1x1_convs(module):
    """give every kernel-size-1 Conv1d / Conv2d under `module` the matrix-product forward above (an instance
    attribute: parameters, state_dict keys and module types are untouched)."""
    for m in module.modules():
        if (isinstance(m, (nn.Conv1d, nn.Conv2d)) and all(k == 1 for k in m.kernel_size) and m.groups == 1
                and all(s == 1 for s in m.stride) and all(p == 0 for p in m.padding)
                and "forward" not in m.__dict__):
            m.forward = functools.partial(conv1x1, m)
    return module


def run_layers(layers, x, style, conv, reduce_max=False):
    """the generic (training / non-fused) walk over [conv, AdaGN, Swish, ...] layer lists.  With gradients enabled on the
    GPU an AdaGN (and the Swish behind it, when there is one) is ONE differentiable op on the library's kernels
    (lion_amd/train_ops.py: two passes forward, three backward, instead of ATen's five and ten).
    reduce_max: additionally the max over the last axis of the result (the SA modules' pooling); when the list ends in
    AdaGN + Swish on a [B, C, M, U] activation that pooling is part of the same op (train_ops.adagn_act_max, round 6)."""
    from .. import train_ops
    i, n = 0, len(layers)
    while i < n:
        layer = layers[i]
        if isinstance(layer, AdaGN):
            if train_ops.usable(x) and layer.n_channel <= 1024:
                fused_act = i + 1 < n and isinstance(layers[i + 1], Swish)
                factor, bias = layer.affine(style)
                if (not fused_act and i + 1 < n and isinstance(layers[i + 1], SE3d) and x.dim() == 5
                        and train_ops.se3d_trainable(layers[i + 1], x)):
                    x = train_ops.adagn_se(x, layer.norm, factor, bias, layers[i + 1])   # AdaGN -> SE3d, nothing between: one op
                    i += 2
                    continue
                if reduce_max and i + (2 if fused_act else 1) == n and train_ops.adagn_act_max_usable(x):
                    return train_ops.adagn_act_max(x, layer.norm, factor, bias, act=fused_act)   # pooled: [B, C, M]
                i += 2 if fused_act else 1
                drop_p, used = train_ops.fusable_dropout(layers, i) if fused_act else (0.0, 0)   # Swish -> Dropout (PVConv :211-222)
                x = train_ops.adagn_act(x, layer.norm, factor, bias, act=fused_act, dropout_p=drop_p)
                i += used
                continue
            x = layer(x, style)
        elif isinstance(layer, (nn.Conv1d, nn.Conv2d, nn.Conv3d)):
            x = conv(layer, x)
        elif isinstance(layer, SE3d) and train_ops.se3d_trainable(layer, x):
            x = train_ops.se3d(layer, x)     # one differentiable op on the library's kernels (round 6)
        else:
            x = layer(x)
        i += 1
    return x.max(dim=-1).values if reduce_max else x


class SharedMLP(nn.Module):
    """[1x1 conv -> AdaGN -> Swish] x len(out_channels) over [B,C,N] (dim=1) or [B,C,M,U] (dim=2)."""

    def __init__(self, in_channels, out_channels, dim=1, cfg={}):
        assert len(cfg) > 0, cfg
        super().__init__()
        conv = nn.Conv1d if dim == 1 else nn.Conv2d
        if not isinstance(out_channels, (list, tuple)):
            out_channels = [out_channels]
        layers = []
        for oc in out_channels:
            layers += [conv(in_channels, oc, 1), AdaGN(dim, cfg, oc), Swish()]
            in_channels = oc
        self.layers = nn.ModuleList(layers)

    def _fusable(self, x):
        return (FUSE_INFERENCE and not self.training and not torch.is_grad_enabled() and x.is_cuda
                and x.dtype == torch.float32 and not torch.is_autocast_enabled())

    def _run(self, x, style, reduce_max=False, add=None):
        """reduce_max: additionally take the max over the last dimension (the SA modules' pooling,
        reference :375-377), fused into the last layer's activation pass on the inference path; add: a tensor summed
        onto the output (PVConv's voxel features), in that same pass on the inference path."""
        if self._fusable(x):
            n = len(self.layers) // 3
            convs, gns = [self.layers[3 * i] for i in range(n)], [self.layers[3 * i + 1] for i in range(n)]
            return fused_ops.shared_mlp(x, convs, gns, style, reduce_max, add=add)
        x = run_layers(list(self.layers), x, style, conv1x1, reduce_max=reduce_max)
        return x if add is None else x + add

    def forward_max(self, x, style):
        return self._run(x, style, reduce_max=True)

    def forward(self, *inputs):
        if len(inputs) == 1 and len(inputs[0]) == 4:  # first layer of a Sequential: one 4-tuple
            inputs = inputs[0]
        if len(inputs) == 4:
            x, _, _, style = inputs
            return (self._run(x, style), *inputs[1:])
        if len(inputs) == 2:
            return self._run(*inputs)
        raise NotImplementedError


class Voxelization(nn.Module):
    """centre, scale by 2*max-norm, +0.5, *r, clamp, round -> mean-pool into the r^3 grid
    (reference :166-193).  One fused call: ``F.voxelize_points`` (P1 + K1 + K2)."""

    def __init__(self, resolution, normalize=True, eps=0):
        super().__init__()
        self.r = int(resolution)
        self.normalize = normalize
        self.eps = eps

    def forward(self, features, coords, return_counts=False, sparse_reader=False):
        """return_counts (inference only): also the per-voxel point counts int32 [B, r^3], from which
        the first convolution derives its empty tiles.  sparse_reader (PVConv's fused branch): the grid will be read by the
        sparse convolution only -- the tile occupancy of the (cloud, r) pair is computed first and the scatter leaves the
        z-rows outside every occupied tile's halo unwritten (round 5)."""
        coords = coords.detach()
        if return_counts:
            from ..functional.backend import _backend
            b, c = features.shape[:2]
            co = coords[:, :3].float().contiguous()
            plan = None
            if _VOX_PLANS is not None and hasattr(_backend, "voxel_index"):
                key = (co.data_ptr(), tuple(co.shape), self.r, bool(self.normalize), float(self.eps))
                hit = _VOX_PLANS.get(key)
                if hit is None:
                    plan = _backend.voxel_index(co, self.r, self.normalize, self.eps)
                    if plan is not None:
                        _VOX_PLANS[key] = (co, plan)
                else:
                    plan = hit[1]
            if plan is not None:   # phases B + C only: the voxel ids of this cloud at this resolution exist already
                occ_m1 = None
                if sparse_reader and SKIP_UNREAD and SPARSE_CONV1 and self.r in (16, 32) and hasattr(_backend, "voxel_scatter"):
                    occ_m1 = _occupancy(plan["cnt"], self.r, 64, b, int(sparse_reader))[0]
                out = _backend.voxel_scatter(features.float().contiguous(), plan, occ_m1)
                return out.view(b, c, self.r, self.r, self.r), plan["norm"], plan["cnt"]
            out, norm_coords, _, counts = _backend.voxelize_points_forward(
                features.float().contiguous(), co, self.r, self.normalize, self.eps)
            return out.view(b, c, self.r, self.r, self.r), norm_coords, counts
        if features is None:
            from ..functional.backend import _backend
            _, norm_coords, _, _ = _backend.voxelize_points_forward(
                None, coords[:, :3].float().contiguous(), self.r, self.normalize, self.eps)
            return features, norm_coords
        return F.voxelize_points(features, coords, self.r, self.normalize, self.eps)

    def extra_repr(self):
        return 'resolution={}{}'.format(
            self.r, ', normalized eps = {}'.format(self.eps) if self.normalize else '')


class PVConv(nn.Module):
    """voxel branch (voxelize -> Conv3d/AdaGN/Swish/Dropout/Conv3d/AdaGN[/SE3d] -> devoxelize)
    + point branch (SharedMLP), optional linear attention (reference :195-280)."""

    def __init__(self, in_channels, out_channels, kernel_size, resolution, normalize=1, eps=0,
                 with_se=False, add_point_feat=True, attention=False, dropout=0.1, verbose=True,
                 cfg={}):
        super().__init__()
        assert len(cfg) > 0, cfg
        self.resolution = resolution
        self.voxelization = Voxelization(resolution, normalize=normalize, eps=eps)
        norm = functools.partial(AdaGN, 3, cfg)
        voxel_layers = [
            nn.Conv3d(in_channels, out_channels, kernel_size, stride=1, padding=kernel_size // 2),
            norm(out_channels),
            Swish(),
            nn.Dropout(dropout),
            nn.Conv3d(out_channels, out_channels, kernel_size, stride=1, padding=kernel_size // 2),
            norm(out_channels),
        ]
        if with_se:
            voxel_layers.append(SE3d(out_channels))
        self.voxel_layers = nn.ModuleList(voxel_layers)
        self.attn = LinearAttention(out_channels, verbose=verbose) if attention else None
        if add_point_feat:
            self.point_features = SharedMLP(in_channels, out_channels, cfg=cfg)
        self.add_point_feat = add_point_feat

    def _aware_level(self):
        """consumer-aware level of this PVConv's occupancy buffers (fused_ops.conv3d_occupancy): 2 when the delta convolution
        runs on the split kernel (it then stages zeros for the rows of conv1's empty tiles without loading them, and conv1
        stores occupied tiles only), else 1; 0 with LION_CONV_SKIP_UNREAD=0."""
        if not SKIP_UNREAD:
            return 0
        from .. import conv_ops
        conv2 = self.voxel_layers[4]
        return 2 if (SKIP_UNREAD_LEVEL2 and conv_ops.use_split(None, conv2.in_channels, conv2.out_channels, self.resolution)) else 1

    def _fused_voxel_branch(self, grid, voxel_coords, style, counts=None):
        """eval-mode voxel branch with every pointwise stage folded into the convolutions / the
        devoxelisation (lion_amd/fused_ops.py): conv1 (+GN sums) -> fold -> conv2 with the
        swish(AdaGN1(.)) prologue (+GN sums) -> fold, SE gate from the channel means ->
        devoxelize(scale*grid + shift).  Dropout is the identity in eval mode."""
        conv1, gn1, conv2, gn2 = self.voxel_layers[0], self.voxel_layers[1], self.voxel_layers[4], self.voxel_layers[5]
        se = self.voxel_layers[6] if len(self.voxel_layers) > 6 else None
        r = self.resolution
        occ1 = occ2 = None
        if SPARSE_CONV1 and counts is not None and r >= 16:
            occ1, occ2 = _occupancy(counts, r, conv1.out_channels, grid.shape[0], self._aware_level())
        f1, g1 = gn1.affine(style)
        y1, (a1, b1) = fused_ops.conv3d_fused(grid, conv1, None, True, occ1,   # skips all-zero tiles
                                              fold=fused_ops.FoldSpec(gn1.norm, f1, g1, r ** 3))
        # conv2's activated input = per-channel constant + a delta that is non-zero only near the points
        f2, g2 = gn2.affine(style)
        y2, (a2, b2) = fused_ops.conv3d_fused(y1, conv2, (a1, b1), True, occ2, prev_conv=conv1,
                                              fold=fused_ops.FoldSpec(gn2.norm, f2, g2, r ** 3, se))
        return fused_ops.devoxelize_affine(y2, voxel_coords, r, a2, b2, plan=_devox_plan(voxel_coords, r))

    def forward(self, inputs):
        features, coords_input, time_emb, style = inputs
        coords = coords_input[:, :3] if coords_input.shape[1] > 3 else coords_input
        assert features.shape[0] == coords.shape[0] and features.shape[2] == coords.shape[2], \
            f'get feat: {features.shape} and {coords.shape}'
        assert coords.shape[1] == 3, f'expect coords: B,3,Npoint, get: {coords.shape}'
        counts = None
        if (FUSE_INFERENCE and not self.training and not torch.is_grad_enabled() and features.is_cuda
                and not torch.is_autocast_enabled()):
            # the fused voxel branch below is the grid's only reader, and its first convolution runs sparse
            will_fuse = fused_ops.fusable(self.voxel_layers[0], self.voxel_layers[4], self.resolution, features.float())
            grid, voxel_coords, counts = self.voxelization(features, coords, return_counts=True,
                                                           sparse_reader=self._aware_level() if will_fuse else 0)
        else:
            grid, voxel_coords = self.voxelization(features, coords)
        if (FUSE_INFERENCE and not self.training and not torch.is_grad_enabled()
                and fused_ops.fusable(self.voxel_layers[0], self.voxel_layers[4], self.resolution, grid)):
            pf = None
            if self.add_point_feat and OVERLAP_POINT_BRANCH:
                # the point branch (1x1 conv + AdaGN + Swish: a handful of short, latency-bound launches) does not
                # depend on the voxel branch: issue it on a second stream (a parallel branch under graph capture),
                # where it runs in the shadow of the MFMA convolutions
                main, side = torch.cuda.current_stream(features.device), _point_stream(features.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    pf = self.point_features(features, style)
            fused = self._fused_voxel_branch(grid, voxel_coords, style, counts)
            if pf is not None:
                main.wait_stream(side)
                pf.record_stream(main)
                fused = fused + pf
            elif self.add_point_feat:   # the residual sum rides on the point branch's last activation pass
                fused = self.point_features._run(features, style, add=fused)
            if self.attn is not None:
                fused = self.attn(fused)
            return fused, coords_input, time_emb, style
        layers = list(self.voxel_layers)
        from .. import train_ops
        if (self.training and len(layers) >= 2 and isinstance(layers[-1], SE3d) and isinstance(layers[-2], AdaGN)
                and train_ops.usable(grid) and layers[-2].n_channel <= 1024 and train_ops.adagn_se_devox_usable(grid, layers[-1])):
            # ... AdaGN -> SE3d -> devoxelize: one op, no gated grid, no dense gradient of the devoxelisation (train_ops._AdaGNSEDevox)
            grid = run_layers(layers[:-2], grid, style, conv3d_module)
            factor, bias = layers[-2].affine(style)
            fused = train_ops.adagn_se_devox(grid, layers[-2].norm, factor, bias, layers[-1], voxel_coords, self.resolution)
        else:
            grid = run_layers(layers, grid, style, conv3d_module)  # convs: fp32-MFMA implicit GEMM (csrc/conv3d.hip)
            fused = F.trilinear_devoxelize(grid, voxel_coords, self.resolution, self.training)
        if self.add_point_feat:
            fused = fused + self.point_features(features, style)
        if self.attn is not None:
            fused = self.attn(fused)
        return fused, coords_input, time_emb, style


class PointNetAModule(nn.Module):
    """global max-pool module (reference :282-318); not instantiated by the LION configs."""

    def __init__(self, in_channels, out_channels, include_coordinates=True, cfg={}):
        super().__init__()
        if not isinstance(out_channels, (list, tuple)):
            out_channels = [[out_channels]]
        elif not isinstance(out_channels[0], (list, tuple)):
            out_channels = [out_channels]
        mlps, total = [], 0
        for oc in out_channels:
            mlps.append(SharedMLP(in_channels + (3 if include_coordinates else 0), oc, dim=1, cfg=cfg))
            total += oc[-1]
        self.include_coordinates = include_coordinates
        self.out_channels = total
        self.mlps = nn.ModuleList(mlps)

    def forward(self, inputs):
        features, coords, time_emb, style = inputs
        if self.include_coordinates:
            features = torch.cat([features, coords], dim=1)
        coords = torch.zeros((coords.size(0), 3, 1), device=coords.device)
        pooled = [mlp(features, style).max(dim=-1, keepdim=True).values for mlp in self.mlps]
        return (torch.cat(pooled, dim=1) if len(pooled) > 1 else pooled[0]), coords, time_emb

    def extra_repr(self):
        return f'out_channels={self.out_channels}, include_coordinates={self.include_coordinates}'


class PointNetSAModule(nn.Module):
    """set abstraction: FPS centres -> ball query + grouping -> SharedMLP(2-D) -> max over the
    neighbourhood (reference :321-386)."""

    def __init__(self, num_centers, radius, num_neighbors, in_channels, out_channels,
                 include_coordinates=True, cfg={}):
        super().__init__()
        if not isinstance(radius, (list, tuple)):
            radius = [radius]
        if not isinstance(num_neighbors, (list, tuple)):
            num_neighbors = [num_neighbors] * len(radius)
        assert len(radius) == len(num_neighbors)
        if not isinstance(out_channels, (list, tuple)):
            out_channels = [[out_channels]] * len(radius)
        elif not isinstance(out_channels[0], (list, tuple)):
            out_channels = [out_channels] * len(radius)
        assert len(radius) == len(out_channels)
        groupers, mlps, total = [], [], 0
        for rad, oc, nn_ in zip(radius, out_channels, num_neighbors):
            groupers.append(BallQuery(radius=rad, num_neighbors=nn_, include_coordinates=include_coordinates))
            mlps.append(SharedMLP(in_channels + (3 if include_coordinates else 0), oc, dim=2, cfg=cfg))
            total += oc[-1]
        self.num_centers = num_centers
        self.out_channels = total
        self.groupers = nn.ModuleList(groupers)
        self.mlps = nn.ModuleList(mlps)

    def forward(self, inputs):
        features, coords, time_emb, style = inputs[0], inputs[1], inputs[2], inputs[3]
        if coords.shape[1] > 3:
            coords = coords[:, :3]
        centers_coords = F.furthest_point_sample(coords, self.num_centers)
        S = centers_coords.shape[-1]
        if time_emb is not None and type(time_emb) is not dict:
            time_emb = time_emb[:, :, :S]
        pooled = [mlp.forward_max(grouper(coords, centers_coords, features), style)
                  for grouper, mlp in zip(self.groupers, self.mlps)]
        out = torch.cat(pooled, dim=1) if len(pooled) > 1 else pooled[0]
        return out, centers_coords, time_emb, style

    def extra_repr(self):
        return f'num_centers={self.num_centers}, out_channels={self.out_channels}'


class PointNetFPModule(nn.Module):
    """feature propagation: 3-NN inverse-distance interpolation + SharedMLP (reference :388-414)."""

    def __init__(self, in_channels, out_channels, cfg={}):
        super().__init__()
        self.mlp = SharedMLP(in_channels=in_channels, out_channels=out_channels, dim=1, cfg=cfg)

    def forward(self, inputs):
        if len(inputs) == 5:
            points_coords, centers_coords, centers_features, time_emb, style = inputs
            points_features = None
        elif len(inputs) == 6:
            points_coords, centers_coords, centers_features, points_features, time_emb, style = inputs
        else:
            raise NotImplementedError
        interpolated = None
        if isinstance(centers_features, tuple):
            # inference (models/latent_points_ada.py): (features, temb) not yet concatenated -- the interpolation reads both
            # and appends the skip features in the same pass (was two torch.cat copies around it)
            cfeat, ctemb = centers_features
            interpolated = fused_ops.three_nn_interpolate_cat(points_coords, centers_coords, cfeat, ctemb, points_features)
            if interpolated is None:
                centers_features = torch.cat([cfeat, ctemb], dim=1)
        if interpolated is None:
            interpolated = F.nearest_neighbor_interpolate(points_coords, centers_coords, centers_features)
            if points_features is not None:
                interpolated = torch.cat([interpolated, points_features], dim=1)
        if time_emb is not None:
            time_emb = time_emb[:, :, 0:1].expand(-1, -1, points_coords.shape[-1])
        return self.mlp(interpolated, style), points_coords, time_emb, style


def _linear_gn_relu(in_channels, out_channels):
    return nn.Sequential(nn.Linear(in_channels, out_channels), nn.GroupNorm(8, out_channels), Swish())


def create_mlp_components(in_channels, out_channels, classifier=False, dim=2, width_multiplier=1, cfg={}):
    """classifier head builder (reference :416-446); entries < 1 in out_channels are dropout rates."""
    r = width_multiplier
    block = _linear_gn_relu if dim == 1 else SharedMLP
    if not isinstance(out_channels, (list, tuple)):
        out_channels = [out_channels]
    if len(out_channels) == 0 or (len(out_channels) == 1 and out_channels[0] is None):
        return nn.Sequential(), in_channels, in_channels
    layers = []
    for oc in out_channels[:-1]:
        if oc < 1:
            layers.append(nn.Dropout(oc))
        else:
            oc = int(r * oc)
            layers.append(block(in_channels, oc, cfg=cfg))
            in_channels = oc
    if dim == 1:
        layers.append(nn.Linear(in_channels, out_channels[-1]) if classifier
                      else _linear_gn_relu(in_channels, int(r * out_channels[-1])))
    else:
        layers.append(nn.Conv1d(in_channels, out_channels[-1], 1) if classifier
                      else SharedMLP(in_channels, int(r * out_channels[-1])))
    return layers, out_channels[-1] if classifier else int(r * out_channels[-1])


def create_pointnet2_sa_components(sa_blocks, extra_feature_channels, input_dim=3, embed_dim=64,
                                   use_att=False, force_att=0, dropout=0.1, with_se=False,
                                   normalize=True, eps=0, has_temb=1, width_multiplier=1,
                                   voxel_resolution_multiplier=1, verbose=True, cfg={}):
    """SA stack builder (reference :448-518).  Quirk preserved on purpose: for
    stages >= 1 only the FIRST conv block is instantiated (`if c == 0 ... elif k == 0`, :484-489),
    although the config asks for 3."""
    assert len(cfg) > 0, cfg
    r, vr = width_multiplier, voxel_resolution_multiplier
    in_channels = extra_feature_channels + input_dim
    sa_layers, sa_in_channels = [], []
    num_centers = None
    for c, (conv_configs, sa_configs) in enumerate(sa_blocks):
        k = 0
        sa_in_channels.append(in_channels)
        stage = []
        if conv_configs is not None:
            out_channels, num_blocks, voxel_resolution = conv_configs
            out_channels = int(r * out_channels)
            for p in range(num_blocks):
                attention = ((c + 1) % 2 == 0 and use_att and p == 0) or (force_att and c > 0)
                if voxel_resolution is None:
                    block = SharedMLP
                else:
                    block = functools.partial(
                        PVConv, kernel_size=3, resolution=int(vr * voxel_resolution),
                        attention=attention, dropout=dropout, with_se=with_se,
                        normalize=normalize, eps=eps, verbose=verbose, cfg=cfg)
                if c == 0:
                    stage.append(block(in_channels, out_channels, cfg=cfg))
                elif k == 0:
                    stage.append(block(in_channels + embed_dim * has_temb, out_channels, cfg=cfg))
                in_channels = out_channels
                k += 1
            extra_feature_channels = in_channels
        if sa_configs is not None:
            num_centers, radius, num_neighbors, out_channels = sa_configs
            out_channels = [[int(r * o) for o in oc] if isinstance(oc, (list, tuple)) else int(r * oc)
                            for oc in out_channels]
            if num_centers is None:
                block = PointNetAModule
            else:
                block = functools.partial(PointNetSAModule, num_centers=num_centers, radius=radius,
                                          num_neighbors=num_neighbors)
            stage.append(block(cfg=cfg,
                               in_channels=extra_feature_channels + (embed_dim * has_temb if k == 0 else 0),
                               out_channels=out_channels, include_coordinates=True))
            in_channels = extra_feature_channels = stage[-1].out_channels
        sa_layers.append(stage[0] if len(stage) == 1 else nn.Sequential(*stage))
    return sa_layers, sa_in_channels, in_channels, 1 if num_centers is None else num_centers


def create_pointnet2_fp_modules(fp_blocks, in_channels, sa_in_channels, embed_dim=64, use_att=False,
                                dropout=0.1, has_temb=1, with_se=False, normalize=True, eps=0,
                                width_multiplier=1, voxel_resolution_multiplier=1, verbose=True, cfg={}):
    """FP stack builder (reference :520-568).  Quirk preserved: the loop variable ``fp_blocks`` is
    shadowed by the per-stage list, so FP-stage attention is never enabled (:546)."""
    assert len(cfg) > 0, cfg
    r, vr = width_multiplier, voxel_resolution_multiplier
    fp_layers = []
    for fp_idx, (fp_configs, conv_configs) in enumerate(fp_blocks):
        stage = []
        out_channels = tuple(int(r * oc) for oc in fp_configs)
        stage.append(PointNetFPModule(
            in_channels=in_channels + sa_in_channels[-1 - fp_idx] + embed_dim * has_temb,
            out_channels=out_channels, cfg=cfg))
        in_channels = out_channels[-1]
        if conv_configs is not None:
            out_channels, num_blocks, voxel_resolution = conv_configs
            out_channels = int(r * out_channels)
            for p in range(num_blocks):
                # reference: (c+1) % 2 == 0 and c < len(fp_blocks) - 1 and use_att and p == 0, where
                # fp_blocks is by now the per-stage list of length 1 + p  ->  always False for c >= 1
                attention = (fp_idx + 1) % 2 == 0 and fp_idx < len(stage) - 1 and use_att and p == 0
                if voxel_resolution is None:
                    block = functools.partial(SharedMLP, cfg=cfg)
                else:
                    block = functools.partial(
                        PVConv, kernel_size=3, resolution=int(vr * voxel_resolution),
                        attention=attention, dropout=dropout, with_se=with_se,
                        normalize=normalize, eps=eps, verbose=verbose, cfg=cfg)
                stage.append(block(in_channels, out_channels))
                in_channels = out_channels
        fp_layers.append(stage[0] if len(stage) == 1 else nn.Sequential(*stage))
    return fp_layers, in_channels
