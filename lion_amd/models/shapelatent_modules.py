"""Global (style) encoder of the VAE -- mirror of the reference's
``models/shapelatent_modules.py:13-54``: two SA stages -> max over points -> Linear -> (mu, log sigma)."""
import torch.nn as nn

from .pvcnn2 import create_pointnet2_sa_components


class PointNetPlusEncoder(nn.Module):
    sa_blocks = [
        [[32, 2, 32], [1024, 0.1, 32, [32, 32]]],
        [[32, 1, 16], [256, 0.2, 32, [32, 64]]],
    ]
    force_att = 0

    def __init__(self, zdim, input_dim, extra_feature_channels=0, args={}):
        super().__init__()
        layers, _, channels_sa_features, _ = create_pointnet2_sa_components(
            self.sa_blocks, extra_feature_channels, input_dim=input_dim, embed_dim=0,
            force_att=self.force_att, use_att=True, with_se=True)
        self.mlp = nn.Linear(channels_sa_features, zdim * 2)
        self.zdim = zdim
        self.layers = nn.ModuleList(layers)
        self.voxel_dim = [n[1][-1][-1] for n in self.sa_blocks]

    def forward(self, x):
        x = x.transpose(1, 2)  # [B, 3, N]
        xyz, features = x, x
        for layer in self.layers:
            features, xyz, _ = layer((features, xyz, None))
        features = self.mlp(features.max(-1)[0])
        return {'mu_1d': features[:, :self.zdim], 'sigma_1d': features[:, self.zdim:]}
