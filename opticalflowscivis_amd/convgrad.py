"""The IFNet-3D convolutions on this package's HIP kernels, as drop-in `torch.nn` subclasses.

Why this exists.  The convolutions are not a SURVEY §8(a) row, but ROCm 7.2 ships no gfx950 tuning
database for MIOpen: its 3-D weight-gradient solvers took 10-80 ms PER LAYER at 128^3 (>90 % of the
train step, profiles/r01_flow3d_128_*), its forward runs CK kernels behind NCDHW<->NDHWC transposes and
every strided / transposed layer goes through GEMM + Col2Im one sample at a time
(profiles/r01_bench_256_kernel_stats_before_conv_kernels.csv).  `Conv3d`, `ConvTranspose3d` and `PReLU`
below keep the parameters, state_dict keys and initialisation of their `torch.nn` bases and route

    forward            -> fs_conv3d_fwd   (k3 s1 p1, k4 s2 p1)   /  fs_conv3d_tr (ConvTranspose3d k4 s2 p1)
    input gradient     -> fs_conv3d_fwd   (flipped weights; strided conv of grad_out for the transposed
                          layers)         /  fs_conv3d_tr (for the k4 s2 convolutions)
    weight gradient    -> fs_conv3d_wrw
    PReLU backward     -> fs_prelu_bwd    (also emits the producing convolution's bias gradient)

with fused autograd nodes for the patterns IFBlock is made of: `ConvPReLU` (conv + bias + PReLU),
`res_unit` (`convblock(x) + x`), and heads that accumulate onto the running flow / mask.  Layers outside
the supported (kernel, stride, padding, size) set and non-fp32 tensors run the `torch.nn` base class
(ATen); there is no alternative dispatch mode -- scripts/{fwd,tr,wrw}bench.py hold the MIOpen comparisons.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F


def _tuple(v, nd):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * nd


_stock_seen = set()


def _note_stock_path(what, x, w, stride, padding):
    """A 3-D fp32 GPU layer that leaves this package's kernels for ATen / MIOpen (a (kernel, stride, padding) outside
    k3 s1 p1 / k4 s2 p1, or a volume beyond the kernels' 32-bit offsets -- 512^3 activations): correct, but on ROCm
    7.2's untuned MIOpen it is the 6 s/step class of run, so say so ONCE per layer shape instead of silently."""
    if x.dim() != 5 or not x.is_cuda or x.dtype != torch.float32:
        return  # 2-D layers and CPU tensors are MIOpen's / ATen's by design
    key = (what, tuple(x.shape[1:]), tuple(w.shape), tuple(stride), tuple(padding))
    if key in _stock_seen:
        return
    _stock_seen.add(key)
    import warnings
    warnings.warn("opticalflowscivis_amd: %s of a 3-D convolution runs on stock ATen / MIOpen kernels, not on the HIP "
                  "implicit-GEMM kernels (input %s, weight %s, stride %s, padding %s: unsupported geometry or beyond "
                  "32-bit offsets)" % (what, tuple(x.shape), tuple(w.shape), tuple(stride), tuple(padding)),
                  RuntimeWarning, stacklevel=3)


from ._lib import ablation_env as _ab_env

_MIN_WORKGROUPS = int(_ab_env("FLOWSCI_CONV_FWD_MIN_WG") or 0)      # (measurement switches: ablation mode only)
_MIN_TR_POSITIONS = int(_ab_env("FLOWSCI_CONV_TR_MIN_POS") or 0)    # input voxels (all samples)


def _hip_fwd_ok(x, cout, out_dhw, k, stride, padding):
    """Route this convolution through fs_conv3d_fwd?  Supported (k,s) pairs, fp32 on the GPU.  (The
    workgroup threshold is a tuning hook; the kernel picks quarter-size bricks for small layers and
    beats MIOpen on every IFNet-3D shape measured, scripts/fwdbench.py.)"""
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 5:
        return False
    from . import ops
    if not ops.conv3d_wrw_supported(k, stride, padding):
        return False
    if not ops.conv3d_fwd_fits(x.shape[2:], out_dhw, k[0]):
        return False  # beyond the kernel's 32-bit offsets: the torch.nn base class takes the layer
    return ops.conv3d_fwd_workgroups(x.shape[0], cout, out_dhw, k[0]) >= _MIN_WORKGROUPS


def _hip_tr_ok(x, cout, k, stride, padding):
    """Route a ConvTranspose3d(4,2,1) forward / Conv3d(4,2,1) input gradient through fs_conv3d_tr?"""
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 5:
        return False
    from . import ops
    if not ops.conv3d_tr_supported(cout, k, stride, padding) or not ops.conv3d_tr_fits(x.shape[2:]):
        return False
    return x.shape[0] * x.shape[2] * x.shape[3] * x.shape[4] >= _MIN_TR_POSITIONS


def _conv_out(n, k, s, p):
    return (n + 2 * p - k) // s + 1


def _conv_forward(x, w, b, stride, padding, transposed, prelu_weight=None):
    """y = conv(x); with `prelu_weight` returns (y, prelu(y)) -- from the convolution's own epilogue on
    the HIP paths, as a separate pass otherwise."""
    nd = x.dim() - 2
    y = None
    if transposed and nd == 3 and _hip_tr_ok(x, w.shape[1], tuple(w.shape[2:]), stride, padding):
        from . import ops
        y = ops.conv3d_tr(x, w, b, None, prelu_weight)
    elif transposed:
        _note_stock_path("the forward", x, w, stride, padding)
        y = (F.conv_transpose3d if nd == 3 else F.conv_transpose2d)(x, w, b, stride, padding)
    elif nd == 3 and _hip_fwd_ok(x, w.shape[0], [_conv_out(n, kk, s, p) for n, kk, s, p in
                                                 zip(x.shape[2:], w.shape[2:], stride, padding)],
                                 tuple(w.shape[2:]), stride, padding):
        from . import ops
        y = ops.conv3d_fwd(x, w, b, w.shape[2], stride[0], padding[0], 0, prelu_weight)
    else:
        _note_stock_path("the forward", x, w, stride, padding)
        y = (F.conv3d if nd == 3 else F.conv2d)(x, w, b, stride, padding)
    if prelu_weight is None or isinstance(y, tuple):
        return y
    return y, F.prelu(y, prelu_weight)


def _conv_grad_input(x, w, gy, stride, padding, transposed):
    nd = x.dim() - 2
    k = tuple(w.shape[2:])
    if (nd == 3 and not transposed and all(s == 1 for s in stride) and all(2 * p == kk - 1 for p, kk in zip(padding, k))):
        # stride-1 "same" convolution: its input gradient IS a forward convolution of grad_out with the
        # flipped, channel-transposed filter
        if nd == 3 and _hip_fwd_ok(gy, w.shape[1], x.shape[2:], k, stride, padding):
            from . import ops  # flip + transpose happen in the kernel's weight re-layout
            return ops.conv3d_fwd(gy, w, None, k[0], 1, padding[0], 1)
        _note_stock_path("the input gradient", gy, w, stride, padding)
        wf = w.transpose(0, 1).flip(*range(2, 2 + nd)).contiguous()
        return (F.conv3d if nd == 3 else F.conv2d)(gy, wf, None, stride, padding)
    if (transposed and nd == 3 and _hip_fwd_ok(gy, w.shape[0], x.shape[2:], k, stride, padding) and
            all(_conv_out(n, kk, s, p) == m for n, kk, s, p, m in
                zip(gy.shape[2:], k, stride, padding, x.shape[2:]))):
        # input gradient of a transposed convolution = the strided convolution of grad_out with the same
        # weight tensor read as [out = Cin_t][in = Cout_t]
        from . import ops
        return ops.conv3d_fwd(gy, w, None, k[0], stride[0], padding[0], 0)
    if (not transposed and nd == 3 and _hip_tr_ok(gy, w.shape[1], k, stride, padding) and
            all(n in (2 * m, 2 * m + 1) for n, m in zip(x.shape[2:], gy.shape[2:]))):
        # input gradient of Conv3d(4, 2, 1) = the transposed convolution of grad_out with the layer's
        # weight [Cout][Cin][64] read as [in][out][64]
        from . import ops
        return ops.conv3d_tr(gy, w, None, x.shape[2:])
    _note_stock_path("the input gradient", gy, w, stride, padding)
    return torch.ops.aten.convolution_backward(  # MIOpen backward-data
        gy, x, w, None, list(stride), list(padding), [1] * nd, transposed, [0] * nd, 1,
        [True, False, False])[0]


def _conv_grad_weight(x, w, gy, stride, padding, transposed):
    """dW[gc, sc, *k] = sum_{b, o} g[b, gc, o] * src_padded[b, sc, o*stride + koff] on fs_conv3d_wrw, with
    (src, g) = (x, gy) for a convolution and (gy, x) for a transposed one (y = conv_transpose(x, w[Cin,Cout,k]):
    dW[ci,co,k] = sum x[b,ci,i] * gy[b,co,i*s+k-p]).  Unsupported layers: ATen's convolution backward."""
    k = tuple(w.shape[2:])
    nd = x.dim() - 2
    src, g = (gy, x) if transposed else (x, gy)
    if nd == 3 and src.is_cuda and src.dtype == torch.float32:
        from . import ops
        if ops.conv3d_wrw_supported(k, stride, padding) and ops.conv3d_wrw_fits(src.shape[2:], g.shape[2:]):
            return ops.conv3d_wrw(g, src, k[0], stride[0], padding[0])
    _note_stock_path("the weight gradient", x, w, stride, padding)
    return torch.ops.aten.convolution_backward(
        gy, x, w, None, list(stride), list(padding), [1] * nd, transposed, [0] * nd, 1,
        [False, True, False])[1]


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, padding, transposed):
        y = _conv_forward(x, w, b, stride, padding, transposed)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, transposed, b is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, padding, transposed, has_bias = ctx.cfg
        nd = x.dim() - 2
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _conv_grad_input(x, w, gy, stride, padding, transposed)
        if ctx.needs_input_grad[1]:
            gw = _conv_grad_weight(x, w, gy, stride, padding, transposed)
        if has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum(dim=(0,) + tuple(range(2, 2 + nd)))
        return gx, gw, gb, None, None, None


class _ConvTrAddFn(torch.autograd.Function):
    """y = conv_transpose(x, w, b) + addend with the addition done in the kernel's epilogue
    (fs_conv3d_tr_add); d y / d addend is the identity."""

    @staticmethod
    def forward(ctx, x, w, b, addend, stride, padding):
        from . import ops
        y = ops.conv3d_tr(x, w, b, None, None, addend)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, b is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, padding, has_bias = ctx.cfg
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _conv_grad_input(x, w, gy, stride, padding, True)
        if ctx.needs_input_grad[1]:
            gw = _conv_grad_weight(x, w, gy, stride, padding, True)
        if has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum(dim=(0, 2, 3, 4))
        return gx, gw, gb, (gy if ctx.needs_input_grad[3] else None), None, None


class _ConvPReLUFn(torch.autograd.Function):
    """conv (or transposed conv) + bias + per-channel PReLU as ONE autograd node: the PReLU backward
    pass (csrc/prelu.hip) already streams the convolution's grad_out, so it also produces the bias
    gradient -- no separate reduction over the activation."""

    @staticmethod
    def forward(ctx, x, w, b, a, stride, padding, transposed):
        y, z = _conv_forward(x, w, b, stride, padding, transposed, a)
        ctx.save_for_backward(x, w, y, a)
        ctx.cfg = (stride, padding, transposed, b is not None)
        return z

    @staticmethod
    def backward(ctx, gz):
        from . import ops
        x, w, y, a = ctx.saved_tensors
        stride, padding, transposed, has_bias = ctx.cfg
        gy, ga, gb = ops.prelu_backward(y, gz.contiguous(), a, want_bias_grad=has_bias)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _conv_grad_input(x, w, gy, stride, padding, transposed)
        if ctx.needs_input_grad[1]:
            gw = _conv_grad_weight(x, w, gy, stride, padding, transposed)
        return gx, gw, gb, ga, None, None, None


class _ConvPReLUCatFn(torch.autograd.Function):
    """PReLU(conv3d(torch.cat(pieces, 1), w, b)) -- IFBlock's first convolution over `torch.cat((img0, img1,
    warped_img0, warped_img1, mask, flow), 1)` (Flow-3D/model/IFNet.py:183, 190-191) -- WITHOUT the concatenation:
    forward and weight gradient read every channel where it already lies (fs_conv3d_fwd_prelu_ms /
    fs_conv3d_wrw_ms: one buffer descriptor per channel in the loader waves); the input gradient is computed as one
    tensor and handed out as channel slices, exactly what torch.cat's backward would return.  Shapes without such a
    kernel concatenate inside this node."""

    @staticmethod
    def forward(ctx, w, b, a, stride, padding, *pieces):
        from . import ops
        k = w.shape[2]
        res = ops.conv3d_fwd_prelu_ms(pieces, w, b, a, k, stride[0], padding[0])
        ctx.ms = res is not None
        if res is None:
            x = torch.cat(pieces, 1)
            y, z = _conv_forward(x, w, b, stride, padding, False, a)
            ctx.save_for_backward(w, y, a, x)
        else:
            y, z = res
            ctx.save_for_backward(w, y, a, *pieces)
        ctx.cfg = (stride, padding, b is not None, [int(t.shape[1]) for t in pieces])
        return z

    @staticmethod
    def backward(ctx, gz):
        from . import ops
        w, y, a = ctx.saved_tensors[:3]
        src = ctx.saved_tensors[3:]
        stride, padding, has_bias, splits = ctx.cfg
        gy, ga, gb = ops.prelu_backward(y, gz.contiguous(), a, want_bias_grad=has_bias)
        need = ctx.needs_input_grad[5:]
        gx = None
        if any(need):
            # (the HIP transposed-convolution path only looks at the input's extent; ATen's wants the tensor)
            k3 = tuple(w.shape[2:])
            hip = _hip_tr_ok(gy, w.shape[1], k3, stride, padding) and all(
                n in (2 * m, 2 * m + 1) for n, m in zip(src[0].shape[2:], gy.shape[2:]))
            gx = _conv_grad_input(src[0] if (hip or not ctx.ms) else torch.cat(src, 1), w, gy, stride, padding, False)
        gw = None
        if ctx.needs_input_grad[0]:
            if ctx.ms:
                gw = ops.conv3d_wrw_ms(gy, src, w.shape[2], stride[0], padding[0])
                if gw is None:
                    gw = _conv_grad_weight(torch.cat(src, 1), w, gy, stride, padding, False)
            else:
                gw = _conv_grad_weight(src[0], w, gy, stride, padding, False)
        gp = [None] * len(splits)
        if gx is not None:
            gp = [g if n else None for g, n in zip(gx.split(splits, 1), need)]
        return (gw, gb, ga, None, None) + tuple(gp)


def conv_prelu_cat(block, pieces):
    """`block` = ConvPReLU(Conv3d, PReLU) applied to torch.cat(pieces, 1), the concatenation skipped where the HIP
    kernels can read the pieces in place (training on the GPU, fp32, k = 4 / stride 2); None otherwise."""
    conv, act = block[0], block[1]
    x0 = pieces[0]
    if _ab_env("FLOWSCI_CONV0_CAT") == "1":  # A/B switch (ablation mode): concatenate as the reference does
        return None
    if not (_hip_autograd(x0) and x0.dim() == 5 and x0.dtype == torch.float32 and isinstance(conv, Conv3d)
            and act.weight.numel() in (1, conv.out_channels) and conv.groups == 1
            and _tuple(conv.dilation, 3) == (1, 1, 1) and conv.padding_mode == "zeros"
            and tuple(conv.weight.shape[2:]) == (4, 4, 4) and _tuple(conv.stride, 3) == (2, 2, 2)
            and sum(int(t.shape[1]) for t in pieces) == conv.in_channels <= 12 and conv.out_channels <= 32):
        return None
    return _ConvPReLUCatFn.apply(conv.weight, conv.bias, act.weight, _tuple(conv.stride, 3), _tuple(conv.padding, 3),
                                 *pieces)


class _HeadFn(torch.autograd.Function):
    """out = deconv2(PReLU(deconv1(x))) [+ addend] -- an IFBlock head (Flow-3D/model/IFNet.py:66-77) -- as ONE autograd
    node, so that the backward pass can fold the PReLU backward into the epilogue of deconv2's input-gradient
    convolution (fs_conv3d_fwd_dprelu): the gradient w.r.t. PReLU's output is never written, and PReLU's own pass
    over three 537 MB tensors (at 256^3) disappears.  Falls back to the two-pass form where the fused kernel does
    not apply."""

    @staticmethod
    def forward(ctx, x, w1, b1, a1, w2, b2, addend):
        from . import ops
        y1, z1 = ops.conv3d_tr(x, w1, b1, None, a1)
        out = ops.conv3d_tr(z1, w2, b2, None, None, addend)
        ctx.save_for_backward(x, w1, a1, y1, z1, w2)
        ctx.has = (b1 is not None, b2 is not None, addend is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        from . import ops
        x, w1, a1, y1, z1, w2 = ctx.saved_tensors
        s3, p3 = (2, 2, 2), (1, 1, 1)
        gout = gout.contiguous()
        gw2 = _conv_grad_weight(z1, w2, gout, s3, p3, True) if ctx.needs_input_grad[4] else None
        gb2 = gout.sum(dim=(0, 2, 3, 4)) if (ctx.has[1] and ctx.needs_input_grad[5]) else None
        fused = ops.conv3d_deconv_grad_input_dprelu(gout, w2, y1, a1)
        if fused is not None:
            gy1, ga1, gb1 = fused
        else:
            gz1 = _conv_grad_input(z1, w2, gout, s3, p3, True)
            gy1, ga1, gb1 = ops.prelu_backward(y1, gz1, a1, want_bias_grad=ctx.has[0])
        if not ctx.has[0]:
            gb1 = None
        gw1 = _conv_grad_weight(x, w1, gy1, s3, p3, True) if ctx.needs_input_grad[1] else None
        gx = _conv_grad_input(x, w1, gy1, s3, p3, True) if ctx.needs_input_grad[0] else None
        return gx, gw1, gb1, ga1, gw2, gb2, (gout if (ctx.has[2] and ctx.needs_input_grad[6]) else None)


def head_fused_ok(head, x, addend):
    """Can `head` = Sequential(ConvTranspose3d, PReLU, ConvTranspose3d) run as _HeadFn on `x`?"""
    d1, act, d2 = head[0], head[1], head[2]
    if not (_hip_autograd(x) and x.dim() == 5 and x.dtype == torch.float32 and isinstance(d1, ConvTranspose3d)
            and isinstance(d2, ConvTranspose3d) and isinstance(act, nn.PReLU)):
        return False
    for d in (d1, d2):
        if not (_tuple(d.kernel_size, 3) == (4, 4, 4) and _tuple(d.stride, 3) == (2, 2, 2) and _tuple(d.padding, 3) == (1, 1, 1)
                and d.groups == 1 and _tuple(d.dilation, 3) == (1, 1, 1) and _tuple(d.output_padding, 3) == (0, 0, 0)):
            return False
    if act.weight.numel() not in (1, d1.out_channels):
        return False
    if not (_hip_tr_ok(x, d1.out_channels, (4, 4, 4), (2, 2, 2), (1, 1, 1))):
        return False
    mid = tuple(2 * n for n in x.shape[2:])
    from . import ops
    return ops.conv3d_tr_supported(d2.out_channels, (4, 4, 4), (2, 2, 2), (1, 1, 1)) and ops.conv3d_tr_fits(mid) and \
        (addend is None or (addend.is_cuda and addend.dtype == torch.float32))


_FUSE_2D = _ab_env("FLOWSCI_CONV2D_UNFUSED") != "1"  # A/B switch (ablation mode): 2-D (conv, PReLU) pairs as two stock nodes


class ConvPReLU(nn.Sequential):
    """nn.Sequential(conv, PReLU) -- the reference's `conv()` / `deconv()` helpers
    (Flow-3D/model/IFNet.py:13-29) -- with the same children (`0` = convolution, `1` = PReLU: identical
    state_dict keys and initialisation order), whose training forward on the GPU is the fused node."""

    def forward(self, x):
        conv, act = self[0], self[1]
        nd = x.dim() - 2
        # 3-D: this package's convolution kernels; 2-D (Flow-2D's IFNet): the stock MIOpen convolution inside the same
        # node, so that the PReLU backward pass (csrc/prelu.hip), which streams the convolution's grad_out anyway, also
        # delivers its bias gradient -- ATen's separate reduction per layer was 48 launches x 14 us of the C2 step
        kinds = {3: (Conv3d, ConvTranspose3d), 2: (nn.Conv2d, nn.ConvTranspose2d) if _FUSE_2D else ()}.get(nd, ())
        if (_hip_autograd(x) and x.dtype == torch.float32 and kinds and isinstance(conv, kinds)
                and act.weight.numel() in (1, conv.out_channels) and conv.groups == 1
                and _tuple(conv.dilation, nd) == (1,) * nd and getattr(conv, "padding_mode", "zeros") == "zeros"
                and _tuple(getattr(conv, "output_padding", 0), nd) == (0,) * nd):
            return _ConvPReLUFn.apply(x, conv.weight, conv.bias, act.weight, _tuple(conv.stride, nd),
                                      _tuple(conv.padding, nd), isinstance(conv, kinds[1]))
        return act(conv(x))


class _ResUnitFn(torch.autograd.Function):
    """out = PReLU(conv2(PReLU(conv1(x)))) + x  -- one `convblockN(x) + x` line of IFBlock.forward
    (Flow-3D/model/IFNet.py:101-104) -- as ONE autograd node on the HIP kernels: the skip addition
    happens in conv2's epilogue, and in the backward pass the skip branch's gradient (grad_out itself)
    is added in the epilogue of conv1's input-gradient convolution.  Stride-1 "same" k = 3 layers only."""

    @staticmethod
    def forward(ctx, x, w1, b1, a1, w2, b2, a2):
        from . import ops
        k, p = w1.shape[2], (w1.shape[2] - 1) // 2
        y1, z1 = ops.conv3d_fwd(x, w1, b1, k, 1, p, 0, a1)
        y2, out = ops.conv3d_fwd(z1, w2, b2, k, 1, p, 0, a2, x)
        ctx.save_for_backward(x, w1, a1, y1, z1, w2, a2, y2)
        ctx.has_bias = (b1 is not None, b2 is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        from . import ops
        x, w1, a1, y1, z1, w2, a2, y2 = ctx.saved_tensors
        k, p = w1.shape[2], (w1.shape[2] - 1) // 2
        s3, p3 = (1, 1, 1), (p, p, p)
        gout = gout.contiguous()
        gy2, ga2, gb2 = ops.prelu_backward(y2, gout, a2, want_bias_grad=ctx.has_bias[1])
        gw2 = _conv_grad_weight(z1, w2, gy2, s3, p3, False) if ctx.needs_input_grad[4] else None
        # the unit's inner PReLU backward runs in the epilogue of conv2's input-gradient convolution where that kernel
        # exists (fs_conv3d_fwd_dprelu, kernel 3); else two passes
        fused = ops.conv3d_k3_grad_input_dprelu(gy2, w2, y1, a1) if k == 3 else None
        if fused is not None:
            gy1, ga1, gb1 = fused
            if not ctx.has_bias[0]:
                gb1 = None
        else:
            gz1 = ops.conv3d_fwd(gy2, w2, None, k, 1, p, 1)
            gy1, ga1, gb1 = ops.prelu_backward(y1, gz1, a1, want_bias_grad=ctx.has_bias[0])
        gw1 = _conv_grad_weight(x, w1, gy1, s3, p3, False) if ctx.needs_input_grad[1] else None
        gx = ops.conv3d_fwd(gy1, w1, None, k, 1, p, 1, None, gout) if ctx.needs_input_grad[0] else None
        return gx, gw1, gb1, ga1, gw2, gb2, ga2


def res_unit(block, x):
    """`block(x) + x` for block = Sequential(ConvPReLU, ConvPReLU) of stride-1 k=3 "same" Conv3d layers;
    the fused node when the HIP path applies, else the plain expression."""
    ok = (_hip_autograd(x) and x.dim() == 5 and x.dtype == torch.float32 and len(block) == 2 and
          all(isinstance(m, ConvPReLU) and isinstance(m[0], Conv3d) and isinstance(m[1], nn.PReLU) for m in block))
    if ok:
        for m in block:
            c = m[0]
            kk = tuple(c.weight.shape[2:])
            ok = ok and (kk == (3, 3, 3) and _tuple(c.stride, 3) == (1, 1, 1) and _tuple(c.padding, 3) == (1, 1, 1)
                         and c.groups == 1 and _tuple(c.dilation, 3) == (1, 1, 1) and c.padding_mode == "zeros"
                         and c.in_channels == c.out_channels == x.shape[1]
                         and m[1].weight.numel() in (1, c.out_channels)
                         and _hip_fwd_ok(x, c.out_channels, x.shape[2:], kk, (1, 1, 1), (1, 1, 1)))
    if not ok:
        return block(x) + x
    (c1, p1), (c2, p2) = block[0], block[1]
    return _ResUnitFn.apply(x, c1.weight, c1.bias, p1.weight, c2.weight, c2.bias, p2.weight)


def _hip_autograd(x):
    """Training on the GPU: run the layer as this module's autograd node (HIP kernels forward and backward)."""
    return x.is_cuda and torch.is_grad_enabled()


class Conv3d(nn.Conv3d):
    def forward(self, x):
        if not _hip_autograd(x):
            st, pd, k = _tuple(self.stride, 3), _tuple(self.padding, 3), tuple(self.weight.shape[2:])
            if x.dim() == 5 and self.padding_mode == "zeros" and self.groups == 1 and _tuple(self.dilation, 3) == (1, 1, 1) \
                    and _hip_fwd_ok(x, self.weight.shape[0], [_conv_out(n, kk, s, p) for n, kk, s, p in
                                                              zip(x.shape[2:], k, st, pd)], k, st, pd):
                from . import ops  # inference (grad mode off: nothing is recorded): same kernel, no autograd node; the
                # parameter itself is passed so that its re-laid-out slab can be kept (ops.prepared_weights)
                return ops.conv3d_fwd(x, self.weight, None if self.bias is None else self.bias.detach(),
                                      k[0], st[0], pd[0], 0)
            return super().forward(x)
        return _ConvFn.apply(x, self.weight, self.bias, _tuple(self.stride, 3), _tuple(self.padding, 3), False)


class ConvTranspose3d(nn.ConvTranspose3d):
    def _plain(self):
        return (self.groups == 1 and _tuple(self.dilation, 3) == (1, 1, 1) and
                _tuple(self.output_padding, 3) == (0, 0, 0))

    def forward(self, x, addend=None):
        """`addend` (optional, the output's shape): returns conv_transpose(x) + addend, accumulated in the
        kernel's epilogue where the HIP path applies."""
        st, pd, k = _tuple(self.stride, 3), _tuple(self.padding, 3), tuple(self.weight.shape[2:])
        hip = x.dim() == 5 and self._plain() and _hip_tr_ok(x, self.weight.shape[1], k, st, pd)
        if addend is not None:
            if hip and tuple(addend.shape[2:]) == tuple(2 * n for n in x.shape[2:]) and \
                    addend.shape[1] == self.weight.shape[1] and addend.dtype == torch.float32:
                if _hip_autograd(x):
                    return _ConvTrAddFn.apply(x, self.weight, self.bias, addend, st, pd)
                from . import ops
                return ops.conv3d_tr(x, self.weight, None if self.bias is None else self.bias.detach(),
                                     None, None, addend)
            return self.forward(x) + addend
        if not _hip_autograd(x):
            if hip:
                from . import ops  # inference: same kernel, no autograd node
                return ops.conv3d_tr(x, self.weight, None if self.bias is None else self.bias.detach())
            return super().forward(x)
        return _ConvFn.apply(x, self.weight, self.bias, st, pd, True)


class PReLU(nn.PReLU):
    """torch.nn.PReLU with the one-pass HIP backward (csrc/prelu.hip); same parameters / keys."""

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and x.dim() >= 3:
            from . import ops
            return ops.prelu(x, self.weight)
        return super().forward(x)
