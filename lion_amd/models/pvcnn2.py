"""Non-adaptive PVCNN blocks (plain GroupNorm) -- mirror of the part of the reference's
``models/pvcnn2.py`` that LION instantiates: the VAE's global style encoder
(``shapelatent_modules.PointNetPlusEncoder``) builds its SA stack from here
(SharedMLP :117, PVConv :170, PointNetSAModule :288, create_pointnet2_sa_components :441).
Blocks exchange 3-tuples (features, coords, time_emb) instead of the ada variant's 4-tuples."""
import functools

import torch
import torch.nn as nn

from .. import functional as F
from ..conv_ops import conv3d_module
from .pvcnn2_ada import BallQuery, LinearAttention, SE3d, Swish, Voxelization


def run_layers(layers, x, reduce_max=False):
    """the walk over a [conv, GroupNorm, Swish, Dropout, ..., SE3d] layer list.  With gradients enabled on the GPU a GroupNorm
    (and the Swish behind it, when there is one) is ONE differentiable op on the library's kernels, as in the adaptive
    blocks (pvcnn2_ada.run_layers; lion_amd/train_ops.py with factor = bias = None), SE3d is train_ops.se3d, and a list that
    ends in GroupNorm + Swish on [B, C, M, U] can take the SA modules' max over the neighbours into the same op
    (reduce_max).  ATen runs each GroupNorm + Swish of the style encoder as 4 launches forward and 9 backward.
    These blocks have no separate fused inference path: without autograd (the frozen VAE's encode inside a prior training
    step) the same ops run, forward only."""
    from .. import train_ops
    i, n = 0, len(layers)
    while i < n:
        layer = layers[i]
        if isinstance(layer, nn.GroupNorm) and train_ops.usable(x, False) and layer.num_channels <= 1024 and layer.affine:
            fused_act = i + 1 < n and isinstance(layers[i + 1], Swish)
            if (not fused_act and i + 1 < n and isinstance(layers[i + 1], SE3d) and x.dim() == 5
                    and train_ops.se3d_trainable(layers[i + 1], x, False)):
                x = train_ops.adagn_se(x, layer, None, None, layers[i + 1])   # GroupNorm -> SE3d, nothing between: one op
                i += 2
                continue
            if reduce_max and i + (2 if fused_act else 1) == n and train_ops.adagn_act_max_usable(x, False):
                return train_ops.adagn_act_max(x, layer, None, None, act=fused_act)   # pooled: [B, C, M]
            i += 2 if fused_act else 1
            drop_p, used = train_ops.fusable_dropout(layers, i) if fused_act else (0.0, 0)
            x = train_ops.adagn_act(x, layer, None, None, act=fused_act, dropout_p=drop_p)
            i += used
            continue
        if isinstance(layer, nn.Conv3d):
            x = conv3d_module(layer, x)
        elif isinstance(layer, SE3d) and train_ops.se3d_trainable(layer, x, False):
            x = train_ops.se3d(layer, x)
        else:
            x = layer(x)
        i += 1
    return x.max(dim=-1).values if reduce_max else x


class SharedMLP(nn.Module):
    def __init__(self, in_channels, out_channels, dim=1):
        super().__init__()
        conv = nn.Conv1d if dim == 1 else nn.Conv2d
        if not isinstance(out_channels, (list, tuple)):
            out_channels = [out_channels]
        layers = []
        for oc in out_channels:
            layers += [conv(in_channels, oc, 1), nn.GroupNorm(8, oc), Swish()]
            in_channels = oc
        self.layers = nn.Sequential(*layers)

    def forward(self, inputs, reduce_max=False):
        if isinstance(inputs, (list, tuple)):
            return (run_layers(self.layers, inputs[0], reduce_max), *inputs[1:])
        return run_layers(self.layers, inputs, reduce_max)


class PVConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, resolution, normalize=1, eps=0,
                 with_se=False, add_point_feat=True, attention=False, dropout=0.1, verbose=True):
        super().__init__()
        self.resolution = resolution
        self.voxelization = Voxelization(resolution, normalize=normalize, eps=eps)
        voxel_layers = [
            nn.Conv3d(in_channels, out_channels, kernel_size, stride=1, padding=kernel_size // 2),
            nn.GroupNorm(8, out_channels),
            Swish(),
            nn.Dropout(dropout),
            nn.Conv3d(out_channels, out_channels, kernel_size, stride=1, padding=kernel_size // 2),
            nn.GroupNorm(8, out_channels),
        ]
        if with_se:
            voxel_layers.append(SE3d(out_channels))
        self.voxel_layers = nn.Sequential(*voxel_layers)
        self.attn = LinearAttention(out_channels, verbose=verbose) if attention else None
        if add_point_feat:
            self.point_features = SharedMLP(in_channels, out_channels)
        self.add_point_feat = add_point_feat

    def forward(self, inputs):
        features, coords_input, time_emb = inputs[0], inputs[1], inputs[2]
        coords = coords_input[:, :3] if coords_input.shape[1] > 3 else coords_input
        assert features.shape[0] == coords.shape[0] and features.shape[2] == coords.shape[2]
        assert coords.shape[1] == 3, f'expect coords: B,3,Npoint, get: {coords.shape}'
        grid, voxel_coords = self.voxelization(features, coords)
        layers = list(self.voxel_layers)
        from .. import train_ops
        if (self.training and len(layers) >= 2 and isinstance(layers[-1], SE3d) and isinstance(layers[-2], nn.GroupNorm)
                and layers[-2].affine and train_ops.usable(grid) and layers[-2].num_channels <= 1024
                and train_ops.adagn_se_devox_usable(grid, layers[-1])):
            grid = run_layers(layers[:-2], grid)   # GroupNorm -> SE3d -> devoxelize: one op (train_ops._AdaGNSEDevox)
            fused = train_ops.adagn_se_devox(grid, layers[-2], None, None, layers[-1], voxel_coords, self.resolution)
            grid = None   # the gated grid does not exist in this form (it only fed the debug payload below)
        else:
            grid = run_layers(layers, grid)
            fused = F.trilinear_devoxelize(grid, voxel_coords, self.resolution, self.training)
        if self.add_point_feat:
            fused = fused + self.point_features(features)
        if self.attn is not None:
            fused = self.attn(fused)
        if time_emb is None:  # reference :246-247 (debug payload; nothing consumes it)
            time_emb = {'voxel_features_4d': grid, 'resolution': self.resolution, 'training': self.training}
        return fused, coords_input, time_emb


class PointNetSAModule(nn.Module):
    def __init__(self, num_centers, radius, num_neighbors, in_channels, out_channels,
                 include_coordinates=True):
        super().__init__()
        if not isinstance(radius, (list, tuple)):
            radius = [radius]
        if not isinstance(num_neighbors, (list, tuple)):
            num_neighbors = [num_neighbors] * len(radius)
        if not isinstance(out_channels, (list, tuple)):
            out_channels = [[out_channels]] * len(radius)
        elif not isinstance(out_channels[0], (list, tuple)):
            out_channels = [out_channels] * len(radius)
        assert len(radius) == len(num_neighbors) == len(out_channels)
        groupers, mlps, total = [], [], 0
        for rad, oc, nn_ in zip(radius, out_channels, num_neighbors):
            groupers.append(BallQuery(radius=rad, num_neighbors=nn_, include_coordinates=include_coordinates))
            mlps.append(SharedMLP(in_channels + (3 if include_coordinates else 0), oc, dim=2))
            total += oc[-1]
        self.num_centers = num_centers
        self.out_channels = total
        self.groupers = nn.ModuleList(groupers)
        self.mlps = nn.ModuleList(mlps)

    def forward(self, inputs):
        features, coords, time_emb = inputs[0], inputs[1], inputs[2]
        if coords.shape[1] > 3:
            coords = coords[:, :3]
        centers_coords = F.furthest_point_sample(coords, self.num_centers)
        if time_emb is not None and type(time_emb) is not dict:
            time_emb = time_emb[:, :, :centers_coords.shape[-1]]
        pooled = [mlp(grouper(coords, centers_coords, features), reduce_max=True)   # max over the neighbours (reference :322)
                  for grouper, mlp in zip(self.groupers, self.mlps)]
        return (torch.cat(pooled, dim=1) if len(pooled) > 1 else pooled[0]), centers_coords, time_emb

    def extra_repr(self):
        return f'num_centers={self.num_centers}, out_channels={self.out_channels}'


def create_pointnet2_sa_components(sa_blocks, extra_feature_channels, input_dim=3, embed_dim=64,
                                   use_att=False, force_att=0, dropout=0.1, with_se=False,
                                   normalize=True, eps=0, has_temb=1, width_multiplier=1,
                                   voxel_resolution_multiplier=1, verbose=True):
    """reference :441-510 (same first-block-only quirk as the ada builder)."""
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
                        normalize=normalize, eps=eps, verbose=verbose)
                if c == 0:
                    stage.append(block(in_channels, out_channels))
                elif k == 0:
                    stage.append(block(in_channels + embed_dim * has_temb, out_channels))
                in_channels = out_channels
                k += 1
            extra_feature_channels = in_channels
        if sa_configs is not None:
            num_centers, radius, num_neighbors, out_channels = sa_configs
            out_channels = [[int(r * o) for o in oc] if isinstance(oc, (list, tuple)) else int(r * oc)
                            for oc in out_channels]
            assert num_centers is not None, "PointNetAModule (global pooling) is not used by LION"
            stage.append(PointNetSAModule(
                num_centers=num_centers, radius=radius, num_neighbors=num_neighbors,
                in_channels=extra_feature_channels + (embed_dim * has_temb if k == 0 else 0),
                out_channels=out_channels, include_coordinates=True))
            in_channels = extra_feature_channels = stage[-1].out_channels
        sa_layers.append(stage[0] if len(stage) == 1 else nn.Sequential(*stage))
    return sa_layers, sa_in_channels, in_channels, 1 if num_centers is None else num_centers
