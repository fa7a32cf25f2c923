"""Hierarchical point-cloud VAE -- mirror of the reference's ``models/vae_adain.py:18-339``
(``Model``: encode :56, recont :137, get_loss :209, sample :301, latent_shape :335)."""
import torch
import torch.nn as nn

from . import import_model
from .distributions import Normal


def loss_fn(predict, target, loss_type, point_dim, batch_size):
    """utils/model_helper.py:17-40 -- only the variants the released configs use."""
    if loss_type == 'mse_sum':
        return ((predict - target) ** 2).view(batch_size, -1).sum(1)
    if loss_type == 'l1_sum':
        return torch.abs(predict - target).view(batch_size, -1).sum(1)
    if loss_type == 'mse':
        return ((predict - target) ** 2).view(batch_size, -1).mean(1)
    if loss_type == 'l1':
        return torch.abs(predict - target).view(batch_size, -1).mean(1)
    raise NotImplementedError(loss_type)


def kl_coeff(step, total_step, constant_step, min_kl_coeff, max_kl_coeff):
    """utils/utils.py kl_coeff: linear warm-up of the KL weight."""
    return max(min(max_kl_coeff * (step - constant_step) / total_step, max_kl_coeff), min_kl_coeff)


class Model(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.num_total_iter = 0
        self.args = args
        self.input_dim = args.ddpm.input_dim
        self.latent_dim = args.shapelatent.latent_dim
        self.kl_weight = args.shapelatent.kl_weight
        self.num_points = args.data.tr_max_sample_points
        self.style_encoder = import_model(args.latent_pts.style_encoder)(
            zdim=args.latent_pts.style_dim, input_dim=self.input_dim, args=args)
        self.style_mlp = import_model(args.latent_pts.style_mlp)(args) if len(args.latent_pts.style_mlp) else None
        self.encoder = import_model(args.shapelatent.encoder_type)(
            zdim=self.latent_dim, input_dim=self.input_dim, args=args)
        self.decoder = import_model(args.shapelatent.decoder_type)(
            context_dim=self.latent_dim, point_dim=args.ddpm.input_dim, args=args)
        from .pvcnn2_ada import route_1x1_convs
        route_1x1_convs(self)  # 1x1 convs as matrix products: GEMM backward instead of the conv library's

    # ---- latent bookkeeping ------------------------------------------------------------------
    def compose_eps(self, all_eps):
        return torch.cat(all_eps, dim=1)

    def decompose_eps(self, all_eps):
        d = self.args.latent_pts.style_dim
        return [all_eps[:, :d], all_eps[:, d:]]

    def latent_shape(self):
        return [[self.args.latent_pts.style_dim, 1, 1],
                [self.num_points * (self.latent_dim + self.input_dim), 1, 1]]

    def global2style(self, style):
        ndim = len(style.shape)
        if ndim == 4:
            style = style.squeeze(-1).squeeze(-1)
        if self.style_mlp is not None:
            style = self.style_mlp(style)
        if ndim == 4:
            style = style.unsqueeze(-1).unsqueeze(-1)
        return style

    # ---- encoders ----------------------------------------------------------------------------
    def _encode_both(self, x, cls_emb=None):
        from .. import geometry
        with geometry.shared():   # both encoders sample / query the same cloud in their first two set abstractions: once
            return self._encode_both_impl(x, cls_emb)

    def _encode_both_impl(self, x, cls_emb=None):
        from .. import geometry
        # one channel-major copy of the cloud for both encoders (each made its own): their first set abstractions then see the
        # SAME coordinate tensor, which is what geometry.shared() keys on; x stays [B, N, 3] for them, as a view of that copy
        if x.is_cuda and x.dim() == 3 and geometry.SHARED:
            x = x.transpose(1, 2).contiguous().transpose(1, 2)
        enc_input = (x, cls_emb) if self.args.data.cond_on_cat else x
        z = self.style_encoder(enc_input)
        g_mu, g_sigma = z['mu_1d'], z['sigma_1d']
        g_dist = Normal(mu=g_mu, log_sigma=g_sigma)
        z_global = g_dist.sample()[0]
        style = torch.cat([z_global, cls_emb], dim=1) if (self.args.data.cond_on_cat and cls_emb is not None) else z_global
        style = self.style_mlp(style) if self.style_mlp is not None else style
        z = self.encoder([x, style])
        l_mu = z['mu_1d']
        l_sigma = z['sigma_1d'] - self.args.shapelatent.log_sigma_offset
        l_dist = Normal(mu=l_mu, log_sigma=l_sigma)
        z_local = l_dist.sample()[0]
        return (z_global, g_mu, g_sigma, g_dist), (z_local, l_mu, l_sigma, l_dist), style

    @torch.no_grad()
    def encode(self, x, class_label=None):
        assert x.shape[2] == self.input_dim
        cls_emb = self.class_embedding(class_label) if self.args.data.cond_on_cat else None
        (zg, gm, gs, gd), (zl, lm, ls, ld), _ = self._encode_both(x, cls_emb)
        all_eps = self.compose_eps([zg, zl])
        all_log_q = [gd.log_p(zg), ld.log_p(zl)]
        latent_list = [[zg, gm, gs], [zl, lm, ls]]
        if self.args.data.cond_on_cat:
            return all_eps, all_log_q, latent_list, cls_emb
        return all_eps, all_log_q, latent_list

    def encode_global(self, x, class_label=None):
        enc_input = (x, self.class_embedding(class_label)) if self.args.data.cond_on_cat else x
        z = self.style_encoder(enc_input)
        return Normal(mu=z['mu_1d'], log_sigma=z['sigma_1d'])

    def encode_local(self, x, style):
        z = self.encoder([x, style])
        return Normal(mu=z['mu_1d'], log_sigma=z['sigma_1d'] - self.args.shapelatent.log_sigma_offset)

    # ---- reconstruction / loss ---------------------------------------------------------------
    def recont(self, x, target=None, class_label=None, cls_emb=None):
        batch_size = x.shape[0]
        assert x.shape[2] == self.input_dim
        x_0_target = x if target is None else target
        if self.args.data.cond_on_cat and class_label is not None:
            cls_emb = self.class_embedding(class_label)
        (zg, gm, gs, gd), (zl, lm, ls, ld), style = self._encode_both(x, cls_emb)
        x_0_pred = self.decoder(None, beta=None, context=zl, style=style)
        make_4d = lambda e: e.unsqueeze(-1).unsqueeze(-1) if len(e.shape) == 2 else e.unsqueeze(-1)
        output = {
            'all_eps': [make_4d(zg), make_4d(zl)],
            'all_log_q': [make_4d(gd.log_p(zg)), make_4d(ld.log_p(zl))],
            'latent_list': [[zg, gm, gs], [zl, lm, ls]],
            'x_0_pred': x_0_pred, 'x_0_target': x_0_target, 'x_t': torch.zeros_like(x_0_target),
            't': torch.zeros(batch_size), 'x_0': x_0_target,
        }
        output['hist/global_var'] = gs.exp()
        if 'LatentPoint' in self.args.shapelatent.decoder_type:
            latent_pts = zl.view(batch_size, -1, self.latent_dim + self.input_dim)[:, :, :self.input_dim]
            # the reference copies this to the host on every call (vae_adain.py:206): a device->host sync per training
            # step, and not capturable in a hipGraph.  It stays on the device; the (out-of-scope) visualiser that reads
            # it calls .cpu() itself.
            output['vis/latent_pts'] = latent_pts.detach().reshape(batch_size, -1, self.input_dim)
        output['final_pred'] = output['x_0_pred']
        return output

    def get_loss(self, x, writer=None, it=None, noisy_input=None, class_label=None, kl_weight=None, **kwargs):
        """L1 reconstruction + KL of both latents (reference :209-296).  kl_weight (not in the reference): the KL weight
        as a 0-d tensor, for callers that keep the annealing schedule in device memory (a captured training step)."""
        a = self.args
        if kl_weight is not None:
            pass
        elif a.trainer.anneal_kl and self.num_total_iter > 0:
            if x.is_cuda and torch.cuda.is_current_stream_capturing():
                # a captured step would replay the weight of THIS iteration for ever: the schedule has to
                # live in device memory -- pass kl_weight as a 0-d tensor and refresh it per step (GraphedTrainStep.set_scalar)
                raise RuntimeError("get_loss under graph capture with trainer.anneal_kl: the annealed KL weight is derived "
                                   "from `it` on the host and would be frozen into the graph; pass kl_weight= as a 0-d "
                                   "device tensor instead")
            kl_weight = kl_coeff(step=it, total_step=a.sde.kl_anneal_portion_vada * self.num_total_iter,
                                 constant_step=a.sde.kl_const_portion_vada * self.num_total_iter,
                                 min_kl_coeff=a.sde.kl_const_coeff_vada, max_kl_coeff=a.sde.kl_max_coeff_vada)
        else:
            kl_weight = self.kl_weight
        batch_size = x.shape[0]
        assert x.shape[2] == self.input_dim
        output = self.recont(noisy_input if noisy_input is not None else x, target=x, class_label=class_label)
        rec_loss = loss_fn(output['x_0_pred'], output['x_0_target'], a.ddpm.loss_type,
                           self.input_dim, batch_size).mean()
        output['print/loss_0'] = rec_loss
        output['rec_loss'] = rec_loss
        weighted, plain = [], []
        for pid, (cz, cmu, log_sigma) in enumerate(output['latent_list']):
            kl = (0.5 * log_sigma.exp() ** 2 + 0.5 * cmu ** 2 - log_sigma - 0.5).view(batch_size, -1)
            if 'LatentPoint' in a.shapelatent.decoder_type and 'Hir' not in a.shapelatent.decoder_type:
                if pid == 1:
                    shp = [batch_size, -1, self.latent_dim + self.input_dim]
                    kl_pt = kl.view(*shp)[:, :, :self.input_dim].sum(2).sum(1)
                    kl_feat = kl.view(*shp)[:, :, self.input_dim:].sum(2).sum(1)
                    weighted += [kl_pt * a.latent_pts.weight_kl_pt, kl_feat * a.latent_pts.weight_kl_feat]
                    output['print/kl_pt%d' % pid] = kl_pt
                    output['print/kl_feat%d' % pid] = kl_feat
                else:
                    weighted.append(kl.sum(-1) * a.latent_pts.weight_kl_glb)
                    output['print/kl_glb%d' % pid] = kl.sum(-1)
            plain.append(kl.sum(-1))
            output['print/kl_%d' % pid] = kl.sum(-1)
            output['print/z_mean_%d' % pid] = cmu.mean()
            output['print/z_var_%d' % pid] = log_sigma.exp() ** 2
            output['print/kl_weight'] = kl_weight
        kl = kl_weight * (sum(weighted) if weighted else sum(plain))
        output['msg/kl'] = kl
        output['msg/rec'] = rec_loss
        output['loss'] = kl + rec_loss * a.weight_recont
        return output

    def pz(self, w):
        return w

    def sample(self, num_samples=10, temp=None, decomposed_eps=[], enable_autocast=False,
               device_str='cuda', cls_emb=None):
        """decode latents to points [B, N, 3] (reference :301-333).  NB the decoder is conditioned on
        z_global itself, not on style_mlp(z_global) (:329-331); style_mlp is '' in every released cfg."""
        if 'LatentPoint' not in self.args.shapelatent.decoder_type:
            raise NotImplementedError
        latent_shape = (num_samples, self.num_points * (self.latent_dim + self.input_dim))
        style_shape = (num_samples, self.args.latent_pts.style_dim)
        if len(decomposed_eps) == 0:
            dev = torch.device(device_str)
            z_local = torch.randn(*latent_shape, device=dev)
            z_global = torch.randn(*style_shape, device=dev)
        else:
            z_global = decomposed_eps[0].view(style_shape)
            z_local = decomposed_eps[1].view(*latent_shape)
        return self.decoder(None, beta=None, context=z_local, style=z_global)
