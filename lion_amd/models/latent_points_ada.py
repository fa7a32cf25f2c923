"""PVCNN2Unet + the VAE's latent-point encoder / decoder -- mirror of the reference's
``models/latent_points_ada.py`` (PVCNN2Unet :19, PointTransPVC :175, LatentPointDecPVC :222)."""
import numpy as np
import torch
import torch.nn as nn

import contextlib

from .. import fused_ops, geometry
from . import pvcnn2_ada
from .adagn import StylePlan
from .pvcnn2_ada import (LinearAttention, SharedMLP, create_mlp_components,
                         create_pointnet2_fp_modules, create_pointnet2_sa_components)

# conv_configs = (out_channels, num_blocks, voxel_resolution), sa_configs = (centers, radius, U, mlp)
_VAE_SA_BLOCKS = [
    ((32, 2, 32), (1024, 0.1, 32, (32, 64))),
    ((64, 3, 16), (256, 0.2, 32, (64, 128))),
    ((128, 3, 8), (64, 0.4, 32, (128, 256))),
    (None, (16, 0.8, 32, (128, 128, 128))),
]
_FP_BLOCKS = [
    ((128, 128), (128, 3, 8)),
    ((128, 128), (128, 3, 8)),
    ((128, 128), (128, 2, 16)),
    ((128, 128, 64), (64, 2, 32)),
]


class PVCNN2Unet(nn.Module):
    """4 SA stages -> global linear attention -> 4 FP stages -> point-wise classifier."""

    def __init__(self, num_classes, embed_dim, use_att, dropout=0.1, extra_feature_channels=3,
                 input_dim=3, width_multiplier=1, voxel_resolution_multiplier=1, time_emb_scales=1.0,
                 verbose=True, condition_input=False, point_as_feat=1, cfg={}, sa_blocks={},
                 fp_blocks={}, clip_forge_enable=0, clip_forge_dim=512):
        super().__init__()
        if extra_feature_channels < 0:
            raise AssertionError("extra_feature_channels must be >= 0")
        for name, value in (("input_dim", input_dim), ("clip_forge_enable", clip_forge_enable),
                            ("sa_blocks", sa_blocks), ("fp_blocks", fp_blocks), ("point_as_feat", point_as_feat),
                            ("condition_input", condition_input), ("time_emb_scales", time_emb_scales),
                            ("embed_dim", embed_dim), ("in_channels", extra_feature_channels + 3)):
            setattr(self, name, value)
        # registration order below == the reference's (state_dict key order is part of the checkpoint format)
        if embed_dim > 0:  # the priors carry a time embedding, the VAE networks do not
            self.embedf = nn.Sequential(nn.Linear(embed_dim, embed_dim), nn.LeakyReLU(0.1, inplace=True),
                                        nn.Linear(embed_dim, embed_dim))
        if clip_forge_enable:
            d_style = cfg.latent_pts.style_dim
            self.clip_forge_mapping = nn.Linear(clip_forge_dim, embed_dim)
            self.style_clip = nn.Linear(d_style + embed_dim, d_style)

        shared = dict(with_se=True, embed_dim=embed_dim, use_att=use_att, dropout=dropout, cfg=cfg,
                      width_multiplier=width_multiplier, voxel_resolution_multiplier=voxel_resolution_multiplier,
                      verbose=verbose)
        down, skip_channels, c_bottom, _ = create_pointnet2_sa_components(
            sa_blocks=sa_blocks, extra_feature_channels=extra_feature_channels, input_dim=input_dim, **shared)
        self.sa_layers = nn.ModuleList(down)
        self.global_att = LinearAttention(c_bottom, 8, verbose=verbose) if use_att else None
        skip_channels[0] = extra_feature_channels + input_dim - 3  # the raw xyz is not fed back to the last FP stage
        up, c_top = create_pointnet2_fp_modules(fp_blocks=fp_blocks, in_channels=c_bottom,
                                                sa_in_channels=skip_channels, **shared)
        self.fp_layers = nn.ModuleList(up)
        head, _ = create_mlp_components(in_channels=c_top, out_channels=[128, dropout, num_classes], classifier=True,
                                        dim=2, width_multiplier=width_multiplier, cfg=cfg)
        self.classifier = nn.ModuleList(head)
        pvcnn2_ada.route_1x1_convs(self)

    def get_timestep_embedding(self, timesteps, device):
        """[sin, cos] of t * time_emb_scales * 10000^(-i/(half-1)); the frequency row is built in float64 numpy and
        cast, as in the reference (:101-115), but once per device (a per-call host->device copy is not
        graph-capturable); odd embed_dim is zero-padded by one column."""
        if timesteps.dim() == 2 and timesteps.shape[1] == 1:
            timesteps = timesteps.squeeze(1)
        if timesteps.dim() != 1:
            raise AssertionError(f"get shape: {timesteps.shape}")
        rows = self.__dict__.setdefault("_freq_cache", {})
        row = rows.get(device)
        if row is None:
            half = self.embed_dim // 2
            row = torch.from_numpy(np.exp(np.arange(0, half) * -(np.log(10000) / (half - 1)))).float().to(device)
            rows[device] = row
        if pvcnn2_ada.own_kernels(timesteps) and timesteps.dtype == torch.float32:   # inference: one launch
            from .. import _lib
            B = timesteps.shape[0]
            emb = torch.empty(B, self.embed_dim, device=timesteps.device, dtype=torch.float32)
            tc = timesteps.contiguous()
            _lib.check(_lib.load().lion_timestep_embedding(_lib.ptr(tc), _lib.ptr(row), float(self.time_emb_scales), B,
                                                           self.embed_dim // 2, self.embed_dim, _lib.ptr(emb),
                                                           _lib.stream_ptr(timesteps.device)), "timestep_embedding")
            return emb
        ang = (timesteps * self.time_emb_scales).unsqueeze(1) * row.unsqueeze(0)
        emb = torch.cat((ang.sin(), ang.cos()), dim=1)
        if self.embed_dim % 2:
            emb = nn.functional.pad(emb, (0, 1), "constant", 0)
        return emb

    def time_embedding(self, t):
        """[S] timesteps -> [S, embed_dim] (embedf of the sinusoidal embedding, reference :150-153): row i depends on t[i]
        alone, so a chain runner computes the rows of a whole chain once (lion_amd/chain.py)"""
        emb = self.get_timestep_embedding(t, t.device)
        if pvcnn2_ada.own_kernels(emb) and len(self.embedf) == 3:  # Linear -> LeakyReLU(0.1) -> Linear: 2 launches
            return pvcnn2_ada.linear(self.embedf[2], pvcnn2_ada.linear(self.embedf[0], emb, 2, self.embedf[1].negative_slope))
        return self.embedf(emb)

    def sa_modules(self):
        """the PointNetSAModule of every set-abstraction stage, in order (lion_amd/geometry.py)"""
        return [blk[-1] if isinstance(blk, nn.Sequential) else blk for blk in self.sa_layers]

    def forward(self, inputs, **kwargs):
        B = inputs.shape[0]
        # a caller that built `inputs` from a point-major latent has the two slices already (fused_ops.latent_unpack)
        coords = kwargs.get('coords', None)
        if coords is None:
            coords = inputs[:, :self.input_dim, :].contiguous()
        features = inputs
        temb = kwargs.get('t', None)
        if temb is not None:
            emb = kwargs.get('temb', None)       # rows of time_embedding(t) from a caller that already has them
            if emb is None:
                t = temb
                if t.ndim == 0 and not len(t.shape) == 1:
                    t = t.view(1).expand(B)
                emb = self.time_embedding(t)
            elif emb.shape[0] == 1 and B > 1:
                emb = emb.expand(B, -1)
            temb = emb[:, :, None].expand(-1, -1, inputs.shape[-1])
        style = kwargs['style']
        if self.clip_forge_enable:
            clip_feat = kwargs['clip_feat']
            assert clip_feat is not None, 'require clip_feat as input'
            clip_feat = self.clip_forge_mapping(clip_feat)
            style = self.style_clip(torch.cat([style, clip_feat], dim=1).contiguous())

        if getattr(self, '_style_plan', None) is None:
            self._style_plan = StylePlan(self)
        sa_mods = self.sa_modules()
        geo = geometry.prefetch(sa_mods, coords) if pvcnn2_ada.FUSE_INFERENCE and not self.training \
            else contextlib.nullcontext()
        # one GEMM for every AdaGN projection; FPS / ball-query chain on a side stream (inference)
        vplans = pvcnn2_ada.voxel_plans() if pvcnn2_ada.FUSE_INFERENCE and not self.training else contextlib.nullcontext()
        own = pvcnn2_ada.own_kernels(inputs) and not self.training
        with self._style_plan.projected(style), geo, vplans:
            coords_list, in_features_list = [], []
            for i, sa_blocks in enumerate(self.sa_layers):
                in_features_list.append(features)
                coords_list.append(coords)
                if i > 0 and temb is not None:
                    cat = fused_ops.concat_broadcast(features, temb) if own else None   # (ATen's cat otherwise)
                    features = torch.cat([features, temb], dim=1) if cat is None else cat
                features, coords, temb, _ = sa_blocks((features, coords, temb, style))

            rest = kwargs.get('rest', None)
            in_features_list[0] = inputs[:, 3:, :].contiguous() if rest is None else rest
            if self.global_att is not None:
                features = self.global_att(features)
            for fp_idx, fp_blocks in enumerate(self.fp_layers):
                if own:   # the FP module's fused interpolation concatenates (lion_three_nn_interpolate_cat_forward)
                    cf = (features, temb)
                else:
                    cf = torch.cat([features, temb], dim=1) if temb is not None else features
                features, coords, temb, _ = fp_blocks(
                    (coords_list[-1 - fp_idx], coords, cf, in_features_list[-1 - fp_idx], temb, style))

            for layer in self.classifier:
                features = layer(features, style) if isinstance(layer, SharedMLP) else layer(features)
        return features


class PointTransPVC(nn.Module):
    """VAE local encoder: [B,N,3] -> mu / log-sigma of the latent points (reference :175-220)."""
    sa_blocks = _VAE_SA_BLOCKS
    fp_blocks = _FP_BLOCKS

    def __init__(self, zdim, input_dim, args={}):
        super().__init__()
        self.zdim = zdim
        self.layers = PVCNN2Unet(2 * zdim + input_dim * 2, embed_dim=0, use_att=1,
                                 extra_feature_channels=0, input_dim=args.ddpm.input_dim, cfg=args,
                                 sa_blocks=self.sa_blocks, fp_blocks=self.fp_blocks,
                                 dropout=args.ddpm.dropout)
        self.skip_weight = args.latent_pts.skip_weight
        self.pts_sigma_offset = args.latent_pts.pts_sigma_offset
        self.input_dim = input_dim

    def forward(self, inputs):
        x, style = inputs
        B, N, D = x.shape
        output = self.layers(x.permute(0, 2, 1).contiguous(), style=style).permute(0, 2, 1).contiguous()
        pt_mu_1d = output[:, :, :self.input_dim].contiguous()
        pt_sigma_1d = output[:, :, self.input_dim:2 * self.input_dim].contiguous() - self.pts_sigma_offset
        pt_mu_1d = self.skip_weight * pt_mu_1d + x
        if self.zdim > 0:
            ft_mu_1d = output[:, :, 2 * self.input_dim:-self.zdim].contiguous()
            ft_sigma_1d = output[:, :, -self.zdim:].contiguous()
            mu_1d = torch.cat([pt_mu_1d, ft_mu_1d], dim=2).view(B, -1).contiguous()
            sigma_1d = torch.cat([pt_sigma_1d, ft_sigma_1d], dim=2).view(B, -1).contiguous()
        else:
            mu_1d = pt_mu_1d.view(B, -1).contiguous()
            sigma_1d = pt_sigma_1d.view(B, -1).contiguous()
        return {'mu_1d': mu_1d, 'sigma_1d': sigma_1d}


class LatentPointDecPVC(nn.Module):
    """VAE decoder: latent points [B, N*(3+D)] + style -> [B,N,3] (reference :222-273)."""
    sa_blocks = _VAE_SA_BLOCKS
    fp_blocks = _FP_BLOCKS

    def __init__(self, point_dim, context_dim, num_points=None, args={}, **kwargs):
        super().__init__()
        self.point_dim = point_dim
        self.context_dim = context_dim + self.point_dim
        self.num_points = args.data.tr_max_sample_points if num_points is None else num_points
        self.layers = PVCNN2Unet(point_dim, embed_dim=0, use_att=1,
                                 extra_feature_channels=context_dim, input_dim=args.ddpm.input_dim,
                                 cfg=args, sa_blocks=self.sa_blocks, fp_blocks=self.fp_blocks,
                                 dropout=args.ddpm.dropout)
        self.skip_weight = args.latent_pts.skip_weight

    def forward(self, x, beta, context, style):
        assert context.shape[1] == self.num_points * self.context_dim
        context = context.view(-1, self.num_points, self.context_dim)
        x = context[:, :, :self.point_dim]
        output = self.layers(context.permute(0, 2, 1).contiguous(), style=style) \
            .permute(0, 2, 1).contiguous()
        return output * self.skip_weight + x
