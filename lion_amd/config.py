"""Configuration object for the hot-path modules.

The reference builds a yacs ``CfgNode`` from ``default_config.py`` + a YAML file; the model
constructors only ever read attributes (``cfg.latent_pts.style_dim`` ...).  ``Cfg`` is a small
attribute-dict that provides the same read interface, so the modules in ``lion_amd.models`` accept
either a reference ``CfgNode`` or a ``Cfg``.  The defaults below are the VALUES (facts) of
``default_config.py:14-355`` for the keys the hot path reads, overridden by the three released prior
configs (config/{airplane,chair,car}_prior_cfg.yml), which differ only in sde.dropout / cates / seed.
"""
from __future__ import annotations

import copy

import yaml


class Cfg(dict):
    """dict with attribute access, recursively."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = Cfg(v) if isinstance(v, dict) and not isinstance(v, Cfg) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def merge(self, other):
        for k, v in (other or {}).items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge(v)
            else:
                self[k] = Cfg(v) if isinstance(v, dict) else v
        return self

    def clone(self):
        return Cfg(copy.deepcopy(dict(self)))


_RELEASED_PRIOR = {
    "ddpm": dict(beta_1=1e-4, beta_T=0.02, sched_mode="linear", num_steps=1000, time_dim=64,
                 dropout=0.1, input_dim=3, model_var_type="fixedlarge", model_mean_type="eps",
                 loss_type="l1_sum", use_p2_weight=0, p2_k=1.0, p2_gamma=1.0, ddim_step=200),
    "sde": dict(mixed_prediction=False, mixing_logit_init=-6, learn_mixing_logit=1,
                num_channels_dae=2048, num_cell_per_scale_dae=8, num_scales_dae=2,
                embedding_dim=128, embedding_scale=1.0, embedding_type="positional", dropout=0.2,
                ddim_skip_type="uniform", ddim_kappa=1.0, ode_sample=0, ema_decay=0.9999,
                learning_rate_dae=2e-4, weight_decay=3e-4,
                prior_model="models.latent_points_ada_localprior.PVCNN2Prior",
                kl_anneal_portion_vada=0.5, kl_const_portion_vada=0.0, kl_const_coeff_vada=1e-7,
                kl_max_coeff_vada=0.5),
    "shapelatent": dict(latent_dim=1, kl_weight=0.5, log_sigma_offset=6.0,
                        decoder_type="models.latent_points_ada.LatentPointDecPVC",
                        encoder_type="models.latent_points_ada.PointTransPVC", model="models.vae_adain"),
    "latent_pts": dict(style_dim=128, ada_mlp_init_scale=0.1, skip_weight=0.01, pts_sigma_offset=0.0,
                       style_prior="models.score_sde.resnet.PriorSEDrop",
                       style_encoder="models.shapelatent_modules.PointNetPlusEncoder", style_mlp="",
                       weight_kl_pt=1.0, weight_kl_feat=1.0, weight_kl_glb=1.0, latent_dim_ext=[64]),
    "clipforge": dict(enable=0, feat_dim=512),
    "data": dict(tr_max_sample_points=2048, te_max_sample_points=2048, cond_on_cat=0, cates="airplane",
                 batch_size=20, batch_size_test=10, num_workers=4, train_drop_last=1, dataset_scale=1,
                 normalize_global=True, normalize_per_shape=False, normalize_shape_box=False,
                 normalize_std_per_axis=False, recenter_per_shape=False, random_subsample=1,
                 sample_with_replacement=1, clip_forge_enable=0, clip_model="ViT-B/32",
                 type="datasets.pointflow_datasets"),
    "trainer": dict(type="trainers.train_2prior", seed=1, anneal_kl=1,
                    opt=dict(type="adam", lr=1e-3, beta1=0.9, beta2=0.99, weight_decay=0.0,
                             ema_decay=0.9999, grad_clip=-1.0)),
    "weight_recont": 1.0,
    "eval_ddim_step": 0,
}


def released_prior_cfg(category: str = "airplane", clip: bool = False) -> Cfg:
    """Values of config/<category>_prior_cfg.yml restricted to the keys the hot path reads.
    ``clip=True`` gives the text2shape variant (train_prior_clip.sh): CLIP-conditioned priors."""
    c = Cfg(copy.deepcopy(_RELEASED_PRIOR))
    c.data.cates = category
    c.sde.dropout = {"airplane": 0.2, "chair": 0.4, "car": 0.3}.get(category, 0.2)
    c.trainer.seed = 100 if category == "chair" else 1
    if clip:
        c.clipforge.enable = 1
        c.latent_pts.style_prior = "models.score_sde.resnet.PriorSEClip"
    return c


def load_yaml(path: str, base: Cfg | None = None) -> Cfg:
    """Merge a reference-style YAML file over ``base`` (default: the released airplane prior)."""
    c = (base or released_prior_cfg()).clone()
    with open(path) as f:
        c.merge(yaml.safe_load(f) or {})
    return c
