"""Drop-in for UPFlow/model/upflow.py: `network_tools` and `UPFlow_net` (PWC-style pyramid, dense
estimator, context network, unsupervised losses) with the per-frame-pair hot path on HIP:

  * feature warps            -> WarpingLayer_no_div            (fs_warp2d, PWC mode + validity mask)
  * 9x9 cost volumes         -> CorrelationFunction            (fs_corr2d; `if_use_cor_pytorch` is
                                                                 ignored: there is no PyTorch fallback)
  * occlusion-check warps    -> tools.torch_warp               (fs_warp2d, PWC mode)
  * photometric warps        -> boundary_dilated_warp.warp_im  (fs_warp2d, DILATED mode)
  * photo / msd losses       -> network_tools.photo_loss_multi_type (fs_robust_sum; fs_wssim for 'SSIM')
  * census loss              -> loss_functions.census_loss_torch    (fs_census_dist + fs_robust_sum)

Convolutions are stock torch.nn (MIOpen).  Module tree and construction order follow the
reference (upflow.py:327-366), so checkpoints and seeds are interchangeable.  The self-guided
upsampling variant (`if_sgu_upsample`, upflow.py:21-92; off in the reference's own training config,
scripts/simple_train.py:327) is `network_tools.sgu_model`: its feature warp and its flow warp are the
same two HIP ops (masked PWC warp, `tools.torch_warp`).
"""
import collections

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ..utils.loss import loss_functions
from ..utils.tools import tools
from .correlation_package.correlation import CorrelationFunction
from .pwc_modules import (ContextNetwork_v2_, FeatureExtractor, FlowEstimatorDense_v2, WarpingLayer_no_div,
                          conv, initialize_msra, upsample2d_flow_as, upsample_flow)


class _SguDenseEstimator(nn.Module):
    """upflow.py:26-63 (`FlowEstimatorDense_temp`): five densely connected 3x3 convolutions + a linear head; returns
    (features, head output).  Children conv1..conv5, conv_last as in the reference (state_dict keys)."""

    def __init__(self, ch_in, f_channels, ch_out):
        super().__init__()
        n = ch_in
        for i, f in enumerate(f_channels):
            setattr(self, "conv%d" % (i + 1), conv(n, f))
            n += f
        self.num_feature_channel = n
        self.conv_last = conv(n, ch_out, isReLU=False)

    def forward(self, x):
        for i in range(1, 6):
            x = torch.cat([getattr(self, "conv%d" % i)(x), x], dim=1)
        return x, self.conv_last(x)


class network_tools:
    class sgu_model(nn.Module):
        """upflow.py:21-92: self-guided upsampling.  From the two 32-channel feature maps of a level (the second warped to
        the first by the bilinearly up-sampled flow) a small dense estimator predicts an interpolation flow and a mask; the
        output is `warp(flow, inter_flow) * (1 - mask) + flow * mask`.  Module tree (`warping_layer`,
        `dense_estimator_mask`, `upsample_output_conv`) and construction order follow the reference, so seeds and
        checkpoints carry over."""

        def __init__(self):
            super().__init__()
            self.warping_layer = WarpingLayer_no_div()
            self.dense_estimator_mask = _SguDenseEstimator(64, f_channels=(32, 32, 32, 16, 8), ch_out=3)
            self.upsample_output_conv = nn.Sequential(conv(3, 16, kernel_size=3, stride=1, dilation=1),
                                                      conv(16, 16, stride=2),
                                                      conv(16, 32, kernel_size=3, stride=1, dilation=1),
                                                      conv(32, 32, stride=2))

        def forward(self, flow_init, feature_1, feature_2, output_level_flow=None):
            if flow_init.shape[2:] != feature_1.shape[2:]:
                flow_init = upsample2d_flow_as(flow_init, feature_1, mode="bilinear", if_rate=True)
            feature_2_warp = self.warping_layer(feature_2, flow_init)  # HIP: warp + validity mask
            _, x_out = self.dense_estimator_mask(torch.cat((feature_1, feature_2_warp), dim=1))
            inter_flow = x_out[:, :2]
            inter_mask = torch.sigmoid(x_out[:, 2:3])
            if output_level_flow is not None:
                inter_flow = upsample2d_flow_as(inter_flow, output_level_flow, mode="bilinear", if_rate=True)
                inter_mask = upsample2d_flow_as(inter_mask, output_level_flow, mode="bilinear")
                flow_init = output_level_flow
            flow_up = tools.torch_warp(flow_init, inter_flow) * (1 - inter_mask) + flow_init * inter_mask  # HIP warp
            return flow_init, flow_up, inter_flow, inter_mask

        def output_conv(self, x):
            return self.upsample_output_conv(x)

    @classmethod
    def normalize_features(cls, feature_list, normalize, center, moments_across_channels=True,
                           moments_across_images=True):
        """upflow.py:96-138: centre / scale features before the cost volume."""
        axes = [1, 2, 3] if moments_across_channels else [2, 3]
        stats = collections.defaultdict(list)
        for f in feature_list:
            stats['mean'].append(torch.mean(f, dim=axes, keepdim=True))
            stats['var'].append(torch.var(f, dim=axes, keepdim=True))
        if moments_across_images:
            stats['mean'] = [torch.mean(torch.stack(stats['mean'], dim=0), dim=(0,))] * len(feature_list)
            stats['var'] = [torch.var(torch.stack(stats['var'], dim=0), dim=(0,))] * len(feature_list)
        stats['std'] = [torch.sqrt(v + 1e-16) for v in stats['var']]
        if center:
            feature_list = [f - m for f, m in zip(feature_list, stats['mean'])]
        if normalize:
            feature_list = [f / s for f, s in zip(feature_list, stats['std'])]
        return feature_list

    @classmethod
    def weighted_ssim(cls, x, y, weight, c1=float('inf'), c2=9e-6, weight_epsilon=0.01):
        """upflow.py:141-196, per-pixel maps with stock ops (API completeness); the loss itself goes
        through the fused kernel, see photo_loss_multi_type."""
        if c1 == float('inf') and c2 == float('inf'):
            raise ValueError('Both c1 and c2 are infinite, SSIM loss is zero. This is likely unintended.')
        pool = lambda z: F.avg_pool2d(z, (3, 3), (1, 1))
        apw = pool(weight)
        wpe = weight + weight_epsilon
        inv = 1.0 / (apw + weight_epsilon)
        wpool = lambda z: pool(z * wpe) * inv
        mu_x, mu_y = wpool(x), wpool(y)
        sigma_x = wpool(x ** 2) - mu_x ** 2
        sigma_y = wpool(y ** 2) - mu_y ** 2
        sigma_xy = wpool(x * y) - mu_x * mu_y
        if c1 == float('inf'):
            n, d = (2 * sigma_xy + c2), (sigma_x + sigma_y + c2)
        elif c2 == float('inf'):
            n, d = 2 * mu_x * mu_y + c1, mu_x ** 2 + mu_y ** 2 + c1
        else:
            n = (2 * mu_x * mu_y + c1) * (2 * sigma_xy + c2)
            d = (mu_x ** 2 + mu_y ** 2 + c1) * (sigma_x + sigma_y + c2)
        return torch.clamp((1 - n / d) / 2, 0, 1), apw

    @classmethod
    def edge_aware_smoothness_order1(cls, img, pred):
        """upflow.py:198-219."""
        gx = lambda t: t[:, :, :-1, :] - t[:, :, 1:, :]
        gy = lambda t: t[:, :, :, :-1] - t[:, :, :, 1:]
        wx = torch.exp(-torch.mean(torch.abs(gx(img)), 1, keepdim=True))
        wy = torch.exp(-torch.mean(torch.abs(gy(img)), 1, keepdim=True))
        return torch.mean(torch.abs(gx(pred)) * wx) + torch.mean(torch.abs(gy(pred)) * wy)

    @classmethod
    def edge_aware_smoothness_order2(cls, img, pred):
        """upflow.py:221-243."""
        gx = lambda t, s=1: t[:, :, :-s, :] - t[:, :, s:, :]
        gy = lambda t, s=1: t[:, :, :, :-s] - t[:, :, :, s:]
        wx = torch.exp(-torch.mean(torch.abs(gx(img, 2)), 1, keepdim=True))
        wy = torch.exp(-torch.mean(torch.abs(gy(img, 2)), 1, keepdim=True))
        return torch.mean(torch.abs(gx(gx(pred))) * wx) + torch.mean(torch.abs(gy(gy(pred))) * wy)

    @classmethod
    def flow_smooth_delta(cls, flow, if_second_order=False):
        """upflow.py:245-264."""
        def grad(x):
            return x[:, :, :, 1:] - x[:, :, :, :-1], x[:, :, 1:] - x[:, :, :-1]
        dx, dy = grad(flow)
        loss = dx.abs().mean() + dy.abs().mean()
        if if_second_order:
            dx2, dxdy = grad(dx)
            dydx, dy2 = grad(dy)
            loss = loss + dx2.abs().mean() + dxdy.abs().mean() + dydx.abs().mean() + dy2.abs().mean()
        return loss

    @classmethod
    def photo_loss_multi_type(cls, x, y, occ_mask, photo_loss_type='abs_robust', photo_loss_delta=0.4,
                              photo_loss_use_occ=False):
        """upflow.py:267-289 (a9): every type is one fused HIP pass (fs_robust_sum / fs_wssim)."""
        if photo_loss_type not in ('abs_robust', 'charbonnier', 'L1', 'SSIM'):
            raise ValueError('wrong photo_loss type: %s' % photo_loss_type)
        return ops.photo_loss_multi_type(x, y, occ_mask, photo_loss_type, photo_loss_delta,
                                         photo_loss_use_occ)


class UPFlow_net(tools.abstract_model):
    class config(tools.abstract_config):
        def __init__(self):  # defaults: upflow.py:296-325
            self.occ_type = 'for_back_check'
            self.alpha_1 = 0.1
            self.alpha_2 = 0.5
            self.occ_check_obj_out_all = 'obj'
            self.stop_occ_gradient = False
            self.smooth_level = 'final'
            self.smooth_type = 'edge'
            self.smooth_order_1_weight = 1
            self.smooth_order_2_weight = 0
            self.photo_loss_type = 'abs_robust'
            self.photo_loss_delta = 0.4
            self.photo_loss_use_occ = False
            self.photo_loss_census_weight = 0
            self.if_norm_before_cost_volume = False
            self.norm_moments_across_channels = True
            self.norm_moments_across_images = True
            self.multi_scale_distillation_weight = 0
            self.multi_scale_distillation_style = 'upup'
            self.multi_scale_distillation_occ = True
            self.if_froze_pwc = False
            self.input_or_sp_input = 1
            self.if_use_boundary_warp = True
            self.if_sgu_upsample = False
            self.if_use_cor_pytorch = False  # accepted for compatibility; the HIP kernel is always used

        def __call__(self):
            return UPFlow_net(self)

    def __init__(self, conf):
        super().__init__()
        self.conf = conf
        self.search_range = 4
        self.num_chs = [3, 16, 32, 64, 96, 128, 196]
        self.estimator_f_channels = (128, 128, 96, 64, 32)
        self.context_f_channels = (128, 128, 128, 96, 64, 32, 2)
        self.output_level = 4
        self.num_levels = 7
        self.leakyRELU = nn.LeakyReLU(0.1, inplace=True)
        self.feature_pyramid_extractor = FeatureExtractor(self.num_chs)
        self.warping_layer = WarpingLayer_no_div()
        self.dim_corr = (self.search_range * 2 + 1) ** 2
        self.num_ch_in = self.dim_corr + 32 + 2
        self.flow_estimators = FlowEstimatorDense_v2(self.num_ch_in, f_channels=self.estimator_f_channels)
        self.context_networks = ContextNetwork_v2_(self.flow_estimators.n_channels + 2,
                                                   f_channels=self.context_f_channels)
        self.conv_1x1 = nn.ModuleList([conv(c, 32, kernel_size=1, stride=1, dilation=1)
                                       for c in (196, 128, 96, 64, 32)])
        self.sgi_model = network_tools.sgu_model() if conf.if_sgu_upsample else None  # upflow.py:361-365
        self.occ_check_model = tools.occ_check_model(occ_type=conf.occ_type, occ_alpha_1=conf.alpha_1,
                                                     occ_alpha_2=conf.alpha_2,
                                                     obj_out_all=conf.occ_check_obj_out_all)
        initialize_msra(self.modules())
        if conf.if_froze_pwc:
            self.froze_PWC()

    def _device(self):
        return next(self.parameters()).device

    def _as_tensor(self, v):
        if isinstance(v, torch.Tensor):
            return v.to(self._device(), torch.float32)
        return torch.as_tensor(np.array(v), dtype=torch.float32).to(self._device())  # upflow.py:415-418

    def forward(self, input_dict: dict):
        """input: im1, im2 (+ if_loss; im1_sp / im2_sp when input_or_sp_input != 1)
        output: flow_f_out, flow_b_out, occ_fw, occ_bw and, with if_loss, the loss terms
        (upflow.py:423-578)."""
        conf = self.conf
        im1_ori, im2_ori = self._as_tensor(input_dict['im1']), self._as_tensor(input_dict['im2'])
        if input_dict['if_loss'] and conf.input_or_sp_input != 1:
            im1, im2 = self._as_tensor(input_dict['im1_sp']), self._as_tensor(input_dict['im2_sp'])
        else:
            im1, im2 = im1_ori, im2_ori
        out = {}
        flow_f, flow_b, flows = self.forward_2_frame_v3(im1, im2, if_loss=input_dict['if_loss'])
        occ_fw, occ_bw = self.occ_check_model(flow_f=flow_f, flow_b=flow_b)  # 0 in occluded areas
        out.update(flow_f_out=flow_f, flow_b_out=flow_b, occ_fw=occ_fw, occ_bw=occ_bw)
        if not input_dict['if_loss']:
            return out

        if conf.smooth_level == 'final':
            s_flow_f, s_flow_b, s_im1, s_im2 = flow_f, flow_b, im1_ori, im2_ori
        elif conf.smooth_level == '1/4':
            s_flow_f, s_flow_b = flows[0]
            th, tw = s_flow_f.shape[2:]
            s_im1 = F.interpolate(im1_ori, (th, tw), mode='area')
            s_im2 = F.interpolate(im2_ori, (th, tw), mode='area')
        else:
            raise ValueError('wrong smooth level choosed: %s' % conf.smooth_level)
        smooth_loss = 0
        for weight, order in ((conf.smooth_order_1_weight, 1), (conf.smooth_order_2_weight, 2)):
            if weight > 0:
                if conf.smooth_type == 'edge':
                    fn = network_tools.edge_aware_smoothness_order1 if order == 1 else \
                        network_tools.edge_aware_smoothness_order2
                    smooth_loss += weight * fn(img=s_im1, pred=s_flow_f)
                    smooth_loss += weight * fn(img=s_im2, pred=s_flow_b)
                elif conf.smooth_type == 'delta':
                    smooth_loss += weight * network_tools.flow_smooth_delta(s_flow_f, order == 2)
                    smooth_loss += weight * network_tools.flow_smooth_delta(s_flow_b, order == 2)
                else:
                    raise ValueError('wrong smooth_type: %s' % conf.smooth_type)
        out['smooth_loss'] = smooth_loss

        if conf.if_use_boundary_warp:
            start = torch.zeros(1, 2, 1, 1, device=im1_ori.device)  # upflow.py:503
            im1_warp = tools.boundary_dilated_warp.warp_im(im2_ori, flow_f, start)
            im2_warp = tools.boundary_dilated_warp.warp_im(im1_ori, flow_b, start)
        else:
            im1_warp = tools.torch_warp(im2_ori, flow_f)
            im2_warp = tools.torch_warp(im1_ori, flow_b)
        if conf.stop_occ_gradient:
            occ_fw, occ_bw = occ_fw.clone().detach(), occ_bw.clone().detach()
        pl = dict(photo_loss_type=conf.photo_loss_type, photo_loss_delta=conf.photo_loss_delta,
                  photo_loss_use_occ=conf.photo_loss_use_occ)
        photo_loss = network_tools.photo_loss_multi_type(im1_ori, im1_warp, occ_fw, **pl) + \
            network_tools.photo_loss_multi_type(im2_ori, im2_warp, occ_bw, **pl)
        out.update(photo_loss=photo_loss, im1_warp=im1_warp, im2_warp=im2_warp)

        census_loss = None
        if conf.photo_loss_census_weight > 0:
            cl = dict(q=conf.photo_loss_delta, charbonnier_or_abs_robust=False,
                      if_use_occ=conf.photo_loss_use_occ, averge=True)
            census_loss = loss_functions.census_loss_torch(img1=im1_ori, img1_warp=im1_warp, mask=occ_fw, **cl) + \
                loss_functions.census_loss_torch(img1=im2_ori, img1_warp=im2_warp, mask=occ_bw, **cl)
            census_loss = census_loss * conf.photo_loss_census_weight
        out['census_loss'] = census_loss

        msd_loss = None
        if conf.multi_scale_distillation_weight > 0:  # upflow.py:537-566
            label_f, label_b = flow_f.clone().detach(), flow_b.clone().detach()
            terms = []
            for scale_fw, scale_bw in flows:
                if conf.multi_scale_distillation_style == 'down':
                    lf = upsample_flow(label_f, target_flow=scale_fw)
                    of = F.interpolate(occ_fw, list(scale_fw.shape[2:]), mode='nearest')
                    lb = upsample_flow(label_b, target_flow=scale_bw)
                    ob = F.interpolate(occ_bw, list(scale_bw.shape[2:]), mode='nearest')
                elif conf.multi_scale_distillation_style == 'upup':
                    lf, lb, of, ob = label_f, label_b, occ_fw, occ_bw
                    scale_fw = upsample_flow(scale_fw, target_flow=lf)
                    scale_bw = upsample_flow(scale_bw, target_flow=lb)
                else:
                    raise ValueError('wrong multi_scale_distillation_style: %s' %
                                     conf.multi_scale_distillation_style)
                for s, l, o in ((scale_fw, lf, of), (scale_bw, lb, ob)):
                    terms.append(network_tools.photo_loss_multi_type(
                        x=s, y=l, occ_mask=o, photo_loss_type='abs_robust',
                        photo_loss_use_occ=conf.multi_scale_distillation_occ))
            msd_loss = conf.multi_scale_distillation_weight * sum(terms)
        out['msd_loss'] = msd_loss
        out['loss_dict'] = {'photo_loss': photo_loss, 'smooth_loss': smooth_loss,
                            'census_loss': census_loss, 'msd_loss': msd_loss}
        return out

    def forward_2_frame_v3(self, x1_raw, x2_raw, if_loss=False):
        """upflow.py:580-619: both directions, coarse to fine over 5 pyramid levels."""
        x1_pyramid = self.feature_pyramid_extractor(x1_raw) + [x1_raw]
        x2_pyramid = self.feature_pyramid_extractor(x2_raw) + [x2_raw]
        b, _, h, w = x1_pyramid[0].shape
        flow_f = x1_raw.new_zeros(b, 2, h, w)
        flow_b = x1_raw.new_zeros(b, 2, h, w)
        levels = []
        for l, (x1, x2) in enumerate(zip(x1_pyramid, x2_pyramid)):
            levels.append((x1, self.conv_1x1[l](x1), x2, self.conv_1x1[l](x2)))
            if l == self.output_level:
                break
        flows = []
        for level, (x1, x1_1by1, x2, x2_1by1) in enumerate(levels):
            flow_f, flow_b, res_f, res_b = self.decode_level_res(level, flow_f, flow_b, x1, x1_1by1, x2,
                                                                 x2_1by1)
            flow_f = flow_f + res_f
            flow_b = flow_b + res_b
            flows.append([flow_f, flow_b])
        flow_f_out = upsample2d_flow_as(flow_f, x1_raw, mode="bilinear", if_rate=True)
        flow_b_out = upsample2d_flow_as(flow_b, x1_raw, mode="bilinear", if_rate=True)
        if self.conf.if_sgu_upsample:  # upflow.py:612-616: the 1/4-resolution flow guided up to full resolution
            f1, f2 = self.sgi_model.output_conv(x1_raw), self.sgi_model.output_conv(x2_raw)
            flow_f_out = self.self_guided_upsample(flow_f, f1, f2, output_level_flow=flow_f_out)
            flow_b_out = self.self_guided_upsample(flow_b, f2, f1, output_level_flow=flow_b_out)
        return flow_f_out, flow_b_out, flows[::-1]

    def self_guided_upsample(self, flow_up_bilinear, feature_1, feature_2, output_level_flow=None):
        """upflow.py:677-679."""
        return self.sgi_model(flow_up_bilinear, feature_1, feature_2, output_level_flow=output_level_flow)[1]

    def decode_level_res(self, level, flow_1, flow_2, feature_1, feature_1_1x1, feature_2, feature_2_1x1):
        """upflow.py:621-663: warp, (normalise,) correlate, estimate, refine."""
        conf = self.conf
        flow_1_up = upsample2d_flow_as(flow_1, feature_1, mode="bilinear", if_rate=True)
        flow_2_up = upsample2d_flow_as(flow_2, feature_2, mode="bilinear", if_rate=True)
        if level == 0:
            feature_2_warp, feature_1_warp = feature_2, feature_1
        else:
            if conf.if_sgu_upsample:  # upflow.py:629-631
                flow_1_up = self.self_guided_upsample(flow_1_up, feature_1_1x1, feature_2_1x1)
                flow_2_up = self.self_guided_upsample(flow_2_up, feature_2_1x1, feature_1_1x1)
            feature_2_warp = self.warping_layer(feature_2, flow_1_up)   # HIP: warp + validity mask
            feature_1_warp = self.warping_layer(feature_1, flow_2_up)
        if (conf.if_norm_before_cost_volume and not conf.norm_moments_across_channels
                and not conf.norm_moments_across_images and feature_1.is_cuda):
            # §8f.4: per-plane normalisation folded into the cost-volume kernels' tile loads (the
            # normalised maps feed nothing else, upflow.py:635-652)
            # both directions of the level in one launch per pass (moments, cost volume, and their adjoints)
            out_corr_1, out_corr_2 = ops.corr2d_pair(feature_1, feature_2_warp, feature_2, feature_1_warp, 4,
                                                     normalize=True)
        else:
            if conf.if_norm_before_cost_volume:
                kw = dict(normalize=True, center=True,
                          moments_across_channels=conf.norm_moments_across_channels,
                          moments_across_images=conf.norm_moments_across_images)
                feature_1, feature_2_warp = network_tools.normalize_features((feature_1, feature_2_warp), **kw)
                feature_2, feature_1_warp = network_tools.normalize_features((feature_2, feature_1_warp), **kw)
            # HIP cost volume through the reference's own autograd shim (upflow.py:649,652)
            out_corr_1 = CorrelationFunction.apply(feature_1, feature_2_warp, 4, 1, 4, 1, 1, 1)
            out_corr_2 = CorrelationFunction.apply(feature_2, feature_1_warp, 4, 1, 4, 1, 1, 1)
        out_corr_relu_1 = self.leakyRELU(out_corr_1)
        out_corr_relu_2 = self.leakyRELU(out_corr_2)
        feat_1, res_1 = self.flow_estimators(torch.cat([out_corr_relu_1, feature_1_1x1, flow_1_up], dim=1))
        feat_2, res_2 = self.flow_estimators(torch.cat([out_corr_relu_2, feature_2_1x1, flow_2_up], dim=1))
        fine_1 = self.context_networks(torch.cat([feat_1, flow_1_up + res_1], dim=1))
        fine_2 = self.context_networks(torch.cat([feat_2, flow_2_up + res_2], dim=1))
        return flow_1_up, flow_2_up, res_1 + fine_1, res_2 + fine_2

    def froze_PWC(self):
        for m in (self.feature_pyramid_extractor, self.flow_estimators, self.context_networks, self.conv_1x1):
            for p in m.parameters():
                p.requires_grad = False

    @classmethod
    def demo(cls, device="cuda", size=(320, 320), seed=0):
        """Counterpart of UPFlow_net.demo() (upflow.py:681-730): random pair, prints the loss terms."""
        conf = cls.config()
        conf.update({'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
                     'norm_moments_across_images': False, 'photo_loss_census_weight': 1,
                     'multi_scale_distillation_weight': 1})
        torch.manual_seed(seed)
        net = conf().to(device)
        g = torch.Generator().manual_seed(seed)
        im = torch.rand(2, 3, *size, generator=g)
        out = net({'im1': im, 'im2': im.roll(2, 3), 'if_loss': True})
        for k, v in out['loss_dict'].items():
            print(k, None if v is None else float(v))
        return out
