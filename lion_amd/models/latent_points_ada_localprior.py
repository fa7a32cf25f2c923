"""Denoiser of the latent points -- mirror of the reference's
``models/latent_points_ada_localprior.py:16-84`` (PVCNN2Prior)."""
import torch

from .. import fused_ops
from .latent_points_ada import PVCNN2Unet, _FP_BLOCKS
from .pvcnn2_ada import own_kernels
from .utils import mask_inactive_variables


class PVCNN2Prior(PVCNN2Unet):
    sa_blocks = [
        ((32, 2, 32), (1024, 0.1, 32, (32, 64))),
        ((64, 3, 16), (256, 0.2, 32, (64, 128))),
        ((128, 3, 8), (64, 0.4, 32, (128, 128))),
        (None, (16, 0.8, 32, (128, 128, 128))),
    ]
    fp_blocks = _FP_BLOCKS

    def __init__(self, args, num_input_channels, cfg):
        # only cfg is used (reference :30-31)
        self.clip_forge_enable = cfg.clipforge.enable
        num_input_channels = num_classes = cfg.shapelatent.latent_dim + cfg.ddpm.input_dim
        self.num_classes = num_classes
        self.num_points = cfg.data.tr_max_sample_points
        super().__init__(
            num_classes, cfg.ddpm.time_dim, True, dropout=cfg.ddpm.dropout,
            input_dim=cfg.ddpm.input_dim, extra_feature_channels=cfg.shapelatent.latent_dim,
            time_emb_scales=cfg.sde.embedding_scale, verbose=True, condition_input=False, cfg=cfg,
            sa_blocks=self.sa_blocks, fp_blocks=self.fp_blocks,
            clip_forge_enable=self.clip_forge_enable, clip_forge_dim=cfg.clipforge.feat_dim)
        self.mixed_prediction = cfg.sde.mixed_prediction
        if self.mixed_prediction:
            init = cfg.sde.mixing_logit_init * torch.ones(size=[1, num_input_channels * self.num_points, 1, 1])
            self.mixing_logit = torch.nn.Parameter(init, requires_grad=True)
        else:
            self.mixing_logit = None
        self.is_active = None

    def geometry_source(self, x):
        """(set-abstraction modules, coordinates f32[B,3,N]) exactly as forward() derives them from x: what a chain runner
        needs to compute the FPS / ball-query chain of a step ahead of the forward (lion_amd/chain.py, geometry.py)"""
        if self.mixed_prediction and self.is_active is not None:   # forward() masks first: the geometry must see the same x
            x = mask_inactive_variables(x, self.is_active)
        if own_kernels(x) and self.input_dim == 3 and 3 <= self.num_classes <= 8:
            return self.sa_modules(), fused_ops.latent_unpack(x, self.num_points, self.num_classes, False, True, False)[1]
        pts = x.view(-1, self.num_points, self.num_classes).permute(0, 2, 1).contiguous()
        return self.sa_modules(), pts[:, :self.input_dim, :].contiguous()

    def forward(self, x, t, *args, **kwargs):
        """x: [B, N*D] or [B, N*D, 1, 1] -> same shape (predicted noise).
        channel_major_out=True (chain runners, lion_amd/chain.py): return the network's own [B, D, N] output instead -- the
        chain's update kernel reads it in that layout (lion_chain_update_noise_cm), saving the transposing copy."""
        assert 'condition_input' in kwargs, 'require condition_input'
        if self.mixed_prediction and self.is_active is not None:
            x = mask_inactive_variables(x, self.is_active)
        input_shape = x.shape
        extra = {}
        if own_kernels(x) and self.input_dim == 3 and 3 < self.num_classes <= 8:
            # one launch: [B, N, D] -> [B, D, N] + the coordinate and feature slices the forward would cut out of it
            x, extra['coords'], extra['rest'] = fused_ops.latent_unpack(x, self.num_points, self.num_classes)
        else:
            x = x.view(-1, self.num_points, self.num_classes).permute(0, 2, 1).contiguous()
        out = super().forward(x, t=t, style=kwargs['condition_input'].squeeze(-1).squeeze(-1),
                              clip_feat=kwargs.get('clip_feat', None), temb=kwargs.get('temb', None), **extra)
        if kwargs.get('channel_major_out', False):
            return out.contiguous()
        return out.permute(0, 2, 1).contiguous().view(input_shape)
