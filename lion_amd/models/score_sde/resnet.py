"""Global (style-latent) denoiser -- mirror of the reference's ``models/score_sde/resnet.py``
(SE :16, ResBlockSEClip :29, ResBlockSEDrop :60, ResBlock :92, Prior :124, PriorSEDrop :221,
PriorSEClip :226).  The input is [B, C, 1, 1]; every Conv2d is a 1x1 conv, i.e. a skinny GEMM whose
cost is streaming the 2048x2048 weights."""
import functools

import torch
import torch.nn as nn

from ..utils import init_temb_fun, mask_inactive_variables


class SE(nn.Module):
    def __init__(self, channel, reduction=8):
        super().__init__()
        self.fc = nn.Sequential(
            nn.Conv2d(channel, channel // reduction, 1, 1, bias=False),
            nn.ReLU(inplace=True),
            nn.Conv2d(channel // reduction, channel, 1, 1, bias=False),
            nn.Sigmoid())

    def forward(self, inputs):
        return inputs * self.fc(inputs)


class ResBlockSEClip(nn.Module):
    """x + SE(relu(conv2(relu(conv1(cat[x + temb, clip]))))); the second half of `t` is the CLIP feature."""

    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.non_linearity = nn.ReLU(inplace=True)
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.conv1 = nn.Conv2d(input_dim * 2, output_dim, 1, 1)
        self.conv2 = nn.Conv2d(output_dim, output_dim, 1, 1)
        self.SE = SE(self.output_dim)

    def forward(self, x, t):
        clip_feat = t[:, self.input_dim:].contiguous()
        t = t[:, :self.input_dim].contiguous()
        h = torch.cat([x + t, clip_feat], dim=1).contiguous()
        h = self.non_linearity(self.conv1(h))
        h = self.non_linearity(self.conv2(h))
        return x + self.SE(h)

    def __repr__(self):
        return "ResBlockSEClip(%d, %d)" % (self.input_dim, self.output_dim)


class ResBlockSEDrop(nn.Module):
    def __init__(self, input_dim, output_dim, dropout):
        super().__init__()
        self.non_linearity = nn.ReLU(inplace=True)
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.conv1 = nn.Conv2d(input_dim, output_dim, 1, 1)
        self.conv2 = nn.Conv2d(output_dim, output_dim, 1, 1)
        self.SE = SE(self.output_dim)
        self.dropout = nn.Dropout(dropout)
        self.dropout_ratio = dropout

    def forward(self, x, t):
        h = self.non_linearity(self.conv1(x + t))
        h = self.dropout(h)
        h = self.non_linearity(self.conv2(h))
        return x + self.SE(h)

    def __repr__(self):
        return "ResBlockSE_withdropout(%d, %d, drop=%f)" % (self.input_dim, self.output_dim, self.dropout_ratio)


class ResBlock(nn.Module):
    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.non_linearity = nn.ELU()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.conv1 = nn.Conv2d(input_dim, output_dim, 1, 1)
        self.conv2 = nn.Conv2d(output_dim, output_dim, 1, 1)
        g = min(self.output_dim // 4, 32)
        self.normalize1 = nn.GroupNorm(num_groups=g, num_channels=self.output_dim, eps=1e-6)
        self.normalize2 = nn.GroupNorm(num_groups=g, num_channels=self.output_dim, eps=1e-6)

    def forward(self, x, t):
        x = x + t
        h = self.non_linearity(self.normalize1(self.conv1(x)))
        h = self.non_linearity(self.normalize2(self.conv2(h)))
        return x + h

    def __repr__(self):
        return "ResBlock(%d, %d)" % (self.input_dim, self.output_dim)


class Prior(nn.Module):
    building_block = ResBlock

    def __init__(self, args, num_input_channels, *oargs, **kwargs):
        super().__init__()
        # args: cfg.sde ; oargs[0]: the global cfg
        self.condition_input = kwargs.get('condition_input', False)
        self.cfg = oargs[0]
        self.clip_forge_enable = self.cfg.clipforge.enable
        self.act = nn.SiLU()
        self.num_scales = args.num_scales_dae
        self.num_input_channels = num_input_channels
        self.nf = nf = args.num_channels_dae
        if self.clip_forge_enable:
            self.clip_feat_mapping = nn.Conv1d(self.cfg.clipforge.feat_dim, self.nf, 1)
        self.mixed_prediction = args.mixed_prediction
        if self.mixed_prediction:
            assert args.mixing_logit_init, 'require learning'
            init = args.mixing_logit_init * torch.ones(size=[1, num_input_channels, 1, 1])
            self.mixing_logit = torch.nn.Parameter(init, requires_grad=True)
        else:
            self.mixing_logit = None
        self.is_active = None
        self.embedding_dim = args.embedding_dim
        self.embedding_dim_mult = 4
        self.temb_fun = init_temb_fun(args.embedding_type, args.embedding_scale, args.embedding_dim)
        self.temb_layer = nn.Sequential(
            nn.Conv2d(self.embedding_dim, self.embedding_dim * 4, 1, 1),
            nn.Conv2d(self.embedding_dim * 4, nf, 1, 1))
        self.input_layer = nn.Conv2d(num_input_channels, nf, 1, 1)
        blocks = [self.building_block(nf, nf) for _ in range(args.num_cell_per_scale_dae)]
        self.output_layer = nn.Conv2d(nf, num_input_channels, 1, 1)  # registered before the blocks,
        self.all_modules = nn.ModuleList(blocks)                       # as in the reference (:188-191)
        from ..pvcnn2_ada import route_1x1_convs
        route_1x1_convs(self)

    def time_embedding(self, t):
        """[S] timesteps -> [S, nf, 1, 1]: row i depends on t[i] alone (the reference computes it per sample and per step,
        resnet.py:199; a chain runner computes the S rows of a chain once -- lion_amd/chain.py)"""
        if t.dim() == 0:
            t = t.expand(1)
        return self.temb_layer(self.temb_fun(t)[:, :, None, None])

    def forward(self, x, t, temb=None, **kwargs):
        """temb (not in the reference): the rows of time_embedding(t) for this call, [B or 1, nf, 1, 1], from a caller that
        already has them"""
        if temb is None:
            temb = self.time_embedding(t)
        if self.clip_forge_enable:
            clip_feat = kwargs['clip_feat']
            clip_feat = self.clip_feat_mapping(clip_feat[:, :, None])[:, :, :, None]
            if temb.shape[0] == 1 and temb.shape[0] < clip_feat.shape[0]:
                temb = temb.expand(clip_feat.shape[0], -1, -1, -1)
            temb = torch.cat([temb, clip_feat], dim=1)
        if self.mixed_prediction and self.is_active is not None:
            x = mask_inactive_variables(x, self.is_active)
        if self._skinny_ok(x):
            return self._forward_skinny(x, temb)
        x = self.input_layer(x)
        for layer in self.all_modules:
            x = layer(x, temb)
        return self.output_layer(x)

    def _skinny_ok(self, x):
        """inference on the GPU with ResBlockSEDrop blocks: every layer is a 32-row GEMM -> csrc/skinny.hip."""
        from .. import pvcnn2_ada
        return (pvcnn2_ada.FUSE_INFERENCE and not self.training and not torch.is_grad_enabled() and x.is_cuda
                and x.dtype == torch.float32 and not torch.is_autocast_enabled() and not self.clip_forge_enable
                and all(type(m) is ResBlockSEDrop for m in self.all_modules)
                and x.dim() == 4 and x.shape[2] == 1 and x.shape[3] == 1)

    def _forward_skinny(self, x, temb):
        """4 launches per residual block (round 6; 5 before) instead of ~15 (channel-major [C, 32] activations throughout):
        h1 = relu(conv1(x + t)); h2 = relu(conv2(h1)); s = relu(fc1 h2); x = x + h2 * sigmoid(fc2 s)."""
        from ... import fused_ops as fo
        b = x.shape[0]
        if temb.shape[0] != 1 and temb.shape[0] != b:
            temb = temb.expand(b, -1, -1, -1)
        xt, tt = fo.to_channel_major_pair(x, temb)      # one launch (a single temb row is broadcast by the kernel)
        bias = lambda conv: conv.bias.detach() if conv.bias is not None else None  # noqa: E731
        h = fo.skinny_finish(fo.skinny_conv(xt, self.input_layer), bias(self.input_layer))
        for blk in self.all_modules:
            p1 = fo.skinny_conv(h, blk.conv1, add=tt)                               # conv1(x + t)
            p2 = fo.skinny_conv(p1, blk.conv2, bias_in=bias(blk.conv1), act_in=1)   # conv2(relu(. + b1))
            p3 = fo.skinny_conv(p2, blk.SE.fc[0], bias_in=bias(blk.conv2), act_in=1)  # fc1(h2), h2 = relu(. + b2)
            hn = fo.skinny_conv_se_finish(p3, blk.SE.fc[2], p2, bias(blk.conv2), h)   # fc2(relu(.)) + the tail, one launch
            if hn is None:
                p4 = fo.skinny_conv(p3, blk.SE.fc[2], act_in=1)                     # fc2(relu(.))
                hn = fo.skinny_finish(p2, bias(blk.conv2), p4, h)                   # x + h2 * sigmoid(.)
            h = hn
        out = fo.skinny_finish(fo.skinny_conv(h, self.output_layer), bias(self.output_layer))
        return fo.from_channel_major(out, b)


class PriorSEDrop(Prior):
    def __init__(self, *args, **kwargs):
        self.building_block = functools.partial(ResBlockSEDrop, dropout=args[0].dropout)
        super().__init__(*args, **kwargs)


class PriorSEClip(Prior):
    building_block = ResBlockSEClip
