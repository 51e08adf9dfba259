"""Drop-in for the parts of UPFlow/model/pwc_modules.py that UPFlow_net uses: `conv`,
`initialize_msra`, the flow up-samplers, `FeatureExtractor`, `WarpingLayer_no_div` (HIP),
`FlowEstimatorDense_v2`, `ContextNetwork_v2_`.  Same module trees => same state_dict keys."""
import torch
import torch.nn as nn
import torch.nn.functional as tf

from ... import ops


def conv(in_planes, out_planes, kernel_size=3, stride=1, dilation=1, isReLU=True):
    """pwc_modules.py:10-53 without the (unused) IN/BN variants."""
    layers = [nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, dilation=dilation,
                        padding=((kernel_size - 1) * dilation) // 2, bias=True)]
    if isReLU:
        layers.append(nn.LeakyReLU(0.1, inplace=True))
    return nn.Sequential(*layers)


def initialize_msra(modules):
    """pwc_modules.py:56-72."""
    for layer in modules:
        if isinstance(layer, (nn.Conv2d, nn.ConvTranspose2d)):
            nn.init.kaiming_normal_(layer.weight)
            if layer.bias is not None:
                nn.init.constant_(layer.bias, 0)


def upsample2d_flow_as(inputs, target_as, mode="bilinear", if_rate=False):
    """pwc_modules.py:80-90: align_corners=True resize; flow vectors scaled by the size ratio."""
    _, _, h, w = target_as.size()
    res = tf.interpolate(inputs, [h, w], mode=mode, align_corners=True)
    if if_rate:
        _, _, h_, w_ = inputs.size()
        u, v = res.chunk(2, dim=1)
        res = torch.cat([u * (w / w_), v * (h / h_)], dim=1)
    return res


def upsample_flow(inputs, target_size=None, target_flow=None, mode="bilinear"):
    """pwc_modules.py:93-105."""
    if target_size is not None:
        h, w = target_size
    elif target_flow is not None:
        _, _, h, w = target_flow.size()
    else:
        raise ValueError('wrong input')
    _, _, h_, w_ = inputs.size()
    res = tf.interpolate(inputs, [h, w], mode=mode, align_corners=True)
    scale = torch.tensor([w / w_, h / h_], dtype=res.dtype, device=res.device).view(1, 2, 1, 1)
    return res * scale


class FeatureExtractor(nn.Module):
    """pwc_modules.py:122-143: 6 stride-2 stages, returned coarse -> fine."""

    def __init__(self, num_chs):
        super().__init__()
        self.num_chs = num_chs
        self.convs = nn.ModuleList()
        for ch_in, ch_out in zip(num_chs[:-1], num_chs[1:]):
            self.convs.append(nn.Sequential(conv(ch_in, ch_out, stride=2), conv(ch_out, ch_out)))

    def forward(self, x):
        pyr = []
        for c in self.convs:
            x = c(x)
            pyr.append(x)
        return pyr[::-1]


class WarpingLayer_no_div(nn.Module):
    """pwc_modules.py:179-207 (a5): zero-padded bilinear warp x validity mask, one HIP kernel."""

    def forward(self, x, flow):
        return ops.warp2d_pwc(x, flow, with_mask=True)


class FlowEstimatorDense_v2(nn.Module):
    """pwc_modules.py:260-291: densely connected estimator."""

    def __init__(self, ch_in, f_channels=(128, 128, 96, 64, 32), out_channel=2):
        super().__init__()
        n = ch_in
        for i, f in enumerate(f_channels):
            setattr(self, "conv%d" % (i + 1), conv(n, f))
            n += f
        self.n_channels = n
        self.conv_last = conv(n, out_channel, isReLU=False)

    def forward(self, x):
        for i in range(1, 6):
            x = torch.cat([getattr(self, "conv%d" % i)(x), x], dim=1)
        return x, self.conv_last(x)


class ContextNetwork_v2_(nn.Module):
    """pwc_modules.py:396-412: dilated context network."""

    def __init__(self, ch_in, f_channels=(128, 128, 128, 96, 64, 32, 2)):
        super().__init__()
        self.convs = nn.Sequential(
            conv(ch_in, f_channels[0], 3, 1, 1), conv(f_channels[0], f_channels[1], 3, 1, 2),
            conv(f_channels[1], f_channels[2], 3, 1, 4), conv(f_channels[2], f_channels[3], 3, 1, 8),
            conv(f_channels[3], f_channels[4], 3, 1, 16), conv(f_channels[4], f_channels[5], 3, 1, 1),
            conv(f_channels[5], f_channels[6], isReLU=False))

    def forward(self, x):
        return self.convs(x)
