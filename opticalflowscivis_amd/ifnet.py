"""RIFE-style IFNet for 2-D frames and 3-D volumes, built around the HIP warp kernels.

Mirrors Flow-2D/model/IFNet.py:34-335 and Flow-3D/model/IFNet.py:31-280 (one dimension-generic
implementation instead of two copies).  The module tree (block0/block1/block2/block_tea, and
conv0/convblock0..3/conv1/conv2 inside each block) and the construction order are the
reference's, so `state_dict()` keys match reference checkpoints and `torch.manual_seed(s)`
followed by construction yields the reference's initial weights.

The per-frame-pair hot path -- the two backward warps per block -- is `ops.warp_pair`, one HIP launch
on the 4/6-channel flow in place.  2-D convolutions are stock `torch.nn` (MIOpen); the 3-D ones are
`torch.nn` subclasses running on this package's implicit-GEMM kernels (convgrad.py), fused per
conv + PReLU pair, per residual unit and with the flow / mask accumulation of every block.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import convgrad, ops

# 3-D layers: torch.nn subclasses on fs_conv3d_{fwd,tr,wrw} (convgrad.py); same parameters / keys
_CONV = {2: nn.Conv2d, 3: convgrad.Conv3d}
_DECONV = {2: nn.ConvTranspose2d, 3: convgrad.ConvTranspose3d}
_INTERP = {2: "bilinear", 3: "trilinear"}


def _conv(nd, cin, cout, kernel_size=3, stride=1, padding=1):
    # Sequential(conv, PReLU) as in the reference (same keys / init order); the pair trains as one autograd node
    # (convgrad.ConvPReLU: fused epilogues on the 3-D kernels; in 2-D the stock MIOpen convolution with its bias
    # gradient taken from the PReLU backward pass)
    return convgrad.ConvPReLU(_CONV[nd](cin, cout, kernel_size, stride, padding, bias=True), convgrad.PReLU(cout))


class _Head(nn.Sequential):
    """Sequential(deconv, PReLU, deconv) of the reference heads (children 0, 1, 2); the first pair runs
    as the fused node in 3-D training, the last deconv can accumulate onto `addend` in its epilogue."""

    def forward(self, x, addend=None):
        if isinstance(self[0], convgrad.ConvTranspose3d):
            if convgrad.head_fused_ok(self, x, addend):
                # one autograd node for the whole head: PReLU's backward runs in the epilogue of the second
                # deconvolution's input gradient
                return convgrad._HeadFn.apply(x, self[0].weight, self[0].bias, self[1].weight, self[2].weight,
                                              self[2].bias, addend)
            h = convgrad.ConvPReLU.forward(self, x)
            return self[2](h) if addend is None else self[2](h, addend)
        y = self[2](convgrad.ConvPReLU.forward(self, x))  # 2-D: stock kernels, (deconv, PReLU) as one node
        return y if addend is None else y + addend


def _head(nd, c, cout):
    return _Head(_DECONV[nd](c, c // 2, 4, 2, 1), convgrad.PReLU(c // 2),
                 _DECONV[nd](c // 2, cout, 4, 2, 1))


def _resize(t, factor, mode, mul=1.0):
    """mul * F.interpolate(t, scale_factor=factor, mode, align_corners=False) on the HIP resize kernels
    (csrc/interp.hip: ATen's arithmetic forward, gather-form adjoint backward; `mul` -- the flow's
    `* scale` / `* 1/scale` of IFBlock.forward -- folded into the same pass)."""
    if factor == 1:
        # scale_factor=1 with align_corners=False reproduces its input exactly (source index ==
        # destination index, weights (1, 0)); the reference still launches it.  Skip the pass.
        return t if mul == 1.0 else t * mul
    if mode == "trilinear":
        return ops.interpolate3d(t, factor, mul)
    return ops.interpolate2d(t, factor, mul)


def _min_spatial(a, b):
    return tuple(min(x, y) for x, y in zip(a.shape[2:], b.shape[2:]))


def _crop(t, spatial):
    if tuple(t.shape[2:]) == tuple(spatial):
        return t
    return t[(slice(None), slice(None)) + tuple(slice(0, s) for s in spatial)]


class IFBlock(nn.Module):
    """Flow-2D/model/IFNet.py:34-122 (conv0 kernel 3) / Flow-3D/model/IFNet.py:31-120 (kernel 4)."""

    def __init__(self, nd, in_planes, c=64):
        super().__init__()
        self.nd = nd
        k0 = 3 if nd == 2 else 4
        self.conv0 = nn.Sequential(_conv(nd, in_planes, c // 2, k0, 2, 1),
                                   _conv(nd, c // 2, c, k0, 2, 1))
        self.convblock0 = nn.Sequential(_conv(nd, c, c), _conv(nd, c, c))
        self.convblock1 = nn.Sequential(_conv(nd, c, c), _conv(nd, c, c))
        self.convblock2 = nn.Sequential(_conv(nd, c, c), _conv(nd, c, c))
        self.convblock3 = nn.Sequential(_conv(nd, c, c), _conv(nd, c, c))
        self.conv1 = _head(nd, c, 2 * nd)  # flow: 4 (2-D) or 6 (3-D) channels
        self.conv2 = _head(nd, c, 1)       # blend-mask logit

    def forward(self, x, flow, scale, flow_base=None, mask_base=None, accumulate=False):
        """Returns (flow_delta, mask_delta) as the reference's IFBlock does.  With `accumulate=True` returns
        (flow, mask, kind):
          "sum"    -- flow_base + flow_delta, mask_base + mask_delta, formed inside the producing kernels
                      (3-D, GPU, scale 1, delta and base of equal extent; a base may be None = zero);
          "lowres" -- scales 2 / 4: `flow` is the flow head's output at the block's working resolution (the
                      caller fuses its up-sampling, the accumulation onto flow_base and the two warps into
                      one launch, ops.upsample_warp_pair), `mask` is mask_base + upsample(mask_delta);
          "delta"  -- (flow_delta, mask_delta) at full resolution, nothing accumulated."""
        mode = _INTERP[self.nd]
        h0 = None
        scale_x = scale  # 1 once x is at the block's working resolution
        if isinstance(x, (tuple, list)):
            # the caller's pieces (img0, img1, warped, mask, ...): at scale 1 the first convolution reads them
            # and the flow where they lie (3-D training: convgrad.conv_prelu_cat) or they are concatenated with
            # the flow in ONE pass instead of cat(cat(pieces), flow)
            if scale == 1 and flow is not None:
                if self.nd == 3:
                    h0 = convgrad.conv_prelu_cat(self.conv0[0], tuple(x) + (flow,))
                if h0 is None:
                    x = torch.cat(tuple(x) + (flow,), 1)
                flow = None
            else:
                # scales 2 / 4 (3-D, GPU): the down-sampling reads the pieces in place as well
                xd = ops.interpolate3d_cat(tuple(x), scale) if (self.nd == 3 and x[0].is_cuda and scale in (2, 4)) else None
                if xd is not None:
                    x, scale_x = xd, 1
                else:
                    x = torch.cat(tuple(x), 1)
        if scale_x != 1 and h0 is None:
            x = _resize(x, 1. / scale, mode)
        if flow is not None:
            if scale != 1:
                flow = _resize(flow, 1. / scale, mode, 1. / scale)
            x = torch.cat((x, flow), 1)
        x = self.conv0(x) if h0 is None else self.conv0[1](h0)
        res = convgrad.res_unit if self.nd == 3 else (lambda blk, t: blk(t) + t)
        x = res(self.convblock0, x)
        x = res(self.convblock1, x)
        x = res(self.convblock2, x)
        x = res(self.convblock3, x)
        if accumulate and self.nd == 3 and x.is_cuda and x.dtype == torch.float32 and scale in (1, 2, 4):
            full = tuple(4 * scale * n for n in x.shape[2:])  # two stride-2 deconvs, then x scale
            if all(b is None or tuple(b.shape[2:]) == full for b in (flow_base, mask_base)):
                if scale == 1:
                    return self.conv1(x, flow_base), self.conv2(x, mask_base), "sum"
                # mask: prev + upsample(delta) in one pass (csrc/interp.hip); flow: left to the caller
                return self.conv1(x), ops.upsample3d_scale_add(self.conv2(x), mask_base, scale, 1.0), "lowres"
        flow = self.conv1(x)
        mask = self.conv2(x)
        if scale != 1:
            flow = _resize(flow, scale, mode, scale)
            mask = _resize(mask, scale, mode)
        return (flow, mask, "delta") if accumulate else (flow, mask)


class IFNet(nn.Module):
    """Three student blocks (scales 4, 2, 1) + a teacher that also sees the ground-truth middle
    frame.  Flow-2D/model/IFNet.py:124-335, Flow-3D/model/IFNet.py:122-280."""

    def __init__(self, nd):
        super().__init__()
        assert nd in (2, 3)
        self.nd = nd
        fc = 2 * nd
        c1 = 96 if nd == 2 else 64
        self.block0 = IFBlock(nd, 2, c=128)
        self.block1 = IFBlock(nd, 5 + fc, c=c1)
        self.block2 = IFBlock(nd, 5 + fc, c=64)
        self.block_tea = IFBlock(nd, 6 + fc, c=64)

    def forward(self, x, scale=(4, 2, 1), timestep=0.5):
        # channel slices of [B,3,...] are strided: split once into contiguous frames (every warp and
        # epilogue launch would otherwise copy them again).  `x` may also be the pair (imgs, gt) the training step
        # holds anyway: the reference concatenates the two only for this method to split them again.
        if isinstance(x, (tuple, list)):
            x, gt = x
        else:
            gt = x[:, 2:3] if self.nd == 2 else x[:, 2:]  # empty at inference time
        img0, img1 = x[:, :1].contiguous(), x[:, 1:2].contiguous()
        gt = gt.contiguous()
        flow_list, merged, mask_list, mask_logits = [], [], [], []
        warped_img0, warped_img1 = img0, img1
        flow = mask = None
        # In 3-D every block's flow is handed to its three other consumers through three aliases (f_in: the next
        # block's input concatenation, f_base: the next block's accumulation, f_dist: the distillation term), so
        # that their gradients reach the warp's backward launch one by one and are summed there
        # (ops._WarpPairAcc) instead of by autograd `add`s over the full-size flow.
        f_in = f_base = f_dist = None
        loss_distill = 0
        stu = [self.block0, self.block1, self.block2]
        for i in range(3):
            if flow is not None:
                # the reference crops everything to the common extent (sizes that are not
                # multiples of 16 make block outputs and inputs differ)
                sp = _min_spatial(img0, warped_img0)
                img0, img1 = _crop(img0, sp), _crop(img1, sp)
                warped_img0, warped_img1 = _crop(warped_img0, sp), _crop(warped_img1, sp)
                mask, f_in, f_base = _crop(mask, sp), _crop(f_in, sp), _crop(f_base, sp)
                flow_d, mask_d, kind = stu[i]((img0, img1, warped_img0, warped_img1, mask),
                                              f_in, scale[i], f_base, mask, accumulate=True)
            else:
                flow_d, mask_d, kind = stu[i](torch.cat((img0, img1), 1), None, scale[i], accumulate=True)
            warped = None
            aliases = None
            if kind == "lowres":
                full = tuple(scale[i] * n for n in flow_d.shape[2:])
                if all(f <= n for f, n in zip(full, img0.shape[2:])):
                    # §8f.1: up-sample x scale, accumulate onto the running flow and warp both frames in ONE
                    # launch (no crop can follow: the flow is not larger than the frames)
                    aliases, w0, w1 = ops.upsample_warp_pair(img0, img1, flow_d, f_base, scale[i])
                    flow = aliases[0]
                    warped = (w0, w1)
                else:
                    flow = ops.upsample3d_scale_add(flow_d, f_base, scale[i], float(scale[i]))
                mask = mask_d
            elif kind == "sum":  # flow + flow_d, mask + mask_d formed inside the producing kernels
                flow, mask = flow_d, mask_d
            elif flow is not None:
                flow = f_base + _crop(flow_d, img0.shape[2:])
                mask = mask + _crop(mask_d, img0.shape[2:])
            else:
                flow, mask = flow_d, mask_d
            if self.nd == 2:
                flow, mask = _crop(flow, img0.shape[2:]), _crop(mask, img0.shape[2:])
            sp = _min_spatial(img0, warped_img0)
            if self.nd == 3:
                flow, mask = _crop(flow, sp), _crop(mask, sp)
                if aliases is not None:
                    aliases = tuple(_crop(a, sp) for a in aliases)
            img0, img1 = _crop(img0, sp), _crop(img1, sp)
            mask_logits.append(mask)
            # hot path: both backward warps of this block in one HIP launch.  In 3-D the launch also
            # hands the flow on to its other consumers (next block, distillation), so that their
            # gradients are folded into the warp's backward launch instead of separate autograd adds.
            if warped is None:
                if self.nd == 3:
                    w0, w1, aliases = ops.warp_pair_acc(img0, img1, flow)
                    warped = (w0, w1)
                else:
                    warped = ops.warp_pair(img0, img1, flow)
            f_in, f_base, f_dist = aliases if aliases is not None else (flow, flow, flow)
            warped_img0, warped_img1 = warped
            flow_list.append(f_dist)
            merged.append((warped_img0, warped_img1))

        if gt.shape[1] == 1:
            sp = _min_spatial(img0, warped_img0)
            img0, img1 = _crop(img0, sp), _crop(img1, sp)
            warped_img0, warped_img1 = _crop(warped_img0, sp), _crop(warped_img1, sp)
            mask, f_in, f_base, gt = _crop(mask, sp), _crop(f_in, sp), _crop(f_base, sp), _crop(gt, sp)
            flow_d, mask_d, kind = self.block_tea(
                (img0, img1, warped_img0, warped_img1, mask, gt), f_in, 1, f_base, mask, accumulate=True)
            if kind == "sum":
                flow_teacher, mask_teacher = flow_d, mask_d
            else:
                flow_teacher, mask_teacher = f_base + _crop(flow_d, sp), mask + _crop(mask_d, sp)
            w0t, w1t = ops.warp_pair(img0, img1, flow_teacher)
            merged_teacher, _ = ops.merge(w0t, w1t, mask_teacher)
        else:
            flow_teacher = None
            merged_teacher = None

        for i in range(3):
            # fused epilogues (a12): sigmoid + blend in one pass, mask test + distillation in another
            m, sig = ops.merge(merged[i][0], merged[i][1], mask_logits[i])
            mask_list.append(sig)
            merged[i] = _crop(m, _min_spatial(m, gt))
            if gt.shape[1] == 1:
                flow_list[i] = _crop(flow_list[i], flow_teacher.shape[2:])
        if gt.shape[1] == 1:
            ft = flow_teacher.detach()
            if (x.is_cuda and all(m.shape == merged[0].shape == merged_teacher.shape for m in merged) and
                    all(f.shape == ft.shape for f in flow_list)):
                # the three terms against the one teacher in one launch each way
                loss_distill = ops.distill_terms3(merged, merged_teacher, gt, flow_list, ft)
            else:
                for i in range(3):
                    loss_distill = loss_distill + ops.distill_term(merged[i], merged_teacher, gt, flow_list[i], ft)
        # Flow-2D returns every block's mask (IFNet.py:276), Flow-3D the last one (IFNet.py:280)
        masks = mask_list if self.nd == 2 else mask_list[2]
        return flow_list, masks, merged, flow_teacher, merged_teacher, loss_distill
