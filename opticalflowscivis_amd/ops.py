"""torch.autograd bindings of the HIP hot-path kernels (C-ABI in include/flowsci_hip.h).

Every op validates its operands on the host (dtype fp32, CUDA device, matching shapes; operands
are made contiguous) and raises ValueError on a mismatch, then launches on PyTorch's current
stream of the operand's device.  Nothing here computes on the CPU: without a GPU build of
libflowsci_hip.so these functions raise.
"""
import ctypes

import torch

from . import _lib

WARP2D_RIFE, WARP2D_PWC, WARP2D_PHOTO, WARP2D_DILATED = 0, 1, 2, 3


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_cuda_f32(name, t, ndim):
    if not isinstance(t, torch.Tensor):
        raise ValueError("%s must be a tensor" % name)
    if t.dtype != torch.float32:
        raise ValueError("%s must be float32, got %s" % (name, t.dtype))
    if t.dim() != ndim:
        raise ValueError("%s must be %d-D, got shape %s" % (name, ndim, tuple(t.shape)))
    if not t.is_cuda:
        raise ValueError("%s must live on a GPU (the HIP hot path has no CPU fallback); got %s" %
                         (name, t.device))
    return t.contiguous()


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _in_dhw(inp, flow):
    """Host int[3] with the sampled volume's extent, or None when it equals the flow's."""
    if tuple(inp.shape[2:]) == tuple(flow.shape[2:]):
        return None
    return (ctypes.c_int * 3)(*inp.shape[2:])


# Optional per-launch timing with HIP events recorded on the launch stream (bench.py's roofline
# leg).  Off by default: no events, no overhead.
_timing = None


def enable_kernel_timing(on=True):
    """Start (on=True: clears previous records) or stop collecting (start, end) event pairs."""
    global _timing
    _timing = {} if on else None


def kernel_timings():
    """{entry point: [ms per launch]} for the launches recorded so far (synchronises)."""
    if _timing is None:
        return {}
    torch.cuda.synchronize()
    return {k: [a.elapsed_time(b) for a, b in v] for k, v in _timing.items()}


def _call(name, *args):
    fn = getattr(_lib.lib(), name)
    if _timing is None:
        _lib.check(fn(*args), name)
        return
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    code = fn(*args)
    e1.record()
    _lib.check(code, name)
    _timing.setdefault(name, []).append((e0, e1))


# --------------------------------------------------------------------------------------------
# a2: Flow-3D/model/warplayer.py:9-41
# --------------------------------------------------------------------------------------------
class _Warp3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, flow):
        inp = _need_cuda_f32("tenInput", inp, 5)
        flow = _need_cuda_f32("tenFlow", flow, 5)
        B, C = inp.shape[:2]
        if flow.shape[0] != B or flow.shape[1] != 3:
            raise ValueError("tenFlow must be [B,3,D,H,W] with tenInput's batch %s, got %s" %
                             (tuple(inp.shape), tuple(flow.shape)))
        if inp.device != flow.device:
            raise ValueError("tenInput and tenFlow are on different devices")
        D, H, W = flow.shape[2:]  # the output takes the flow's extent (warplayer.py:11-22, 36)
        out = inp.new_empty((B, C, D, H, W))
        with torch.cuda.device(inp.device):
            _call("fs_warp3d_fwd", inp.data_ptr(), flow.data_ptr(), out.data_ptr(),
                  B, C, _in_dhw(inp, flow), D, H, W, _stream(inp))
        ctx.save_for_backward(inp, flow)
        return out

    @staticmethod
    def backward(ctx, gout):
        inp, flow = ctx.saved_tensors
        need_in, need_flow = ctx.needs_input_grad
        if not (need_in or need_flow):
            return None, None
        gout = gout.contiguous()
        B, C = inp.shape[:2]
        D, H, W = flow.shape[2:]
        gin = torch.zeros_like(inp) if need_in else None
        gflow = torch.empty_like(flow) if need_flow else None
        with torch.cuda.device(inp.device):
            _call("fs_warp3d_bwd", inp.data_ptr(), flow.data_ptr(), gout.data_ptr(),
                  _ptr(gin), _ptr(gflow), B, C, _in_dhw(inp, flow), D, H, W, _stream(inp))
        return gin, gflow


def warp3d(tenInput, tenFlow):
    """Trilinear backward warp with the reference's axis-rotating grid (Flow-3D warplayer.warp)."""
    return _Warp3D.apply(tenInput, tenFlow)


# --------------------------------------------------------------------------------------------
# 2-D warps: a1, a5, a6, a7, a11
# --------------------------------------------------------------------------------------------
class _Warp2D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, flow, start, mode, with_mask):
        inp = _need_cuda_f32("input", inp, 4)
        flow = _need_cuda_f32("flow", flow, 4)
        B, C, H, W = inp.shape
        if tuple(flow.shape) != (B, 2, H, W):
            raise ValueError("flow must be [B,2,H,W] matching input %s, got %s" %
                             (tuple(inp.shape), tuple(flow.shape)))
        if inp.device != flow.device:
            raise ValueError("input and flow are on different devices")
        if start is not None:
            start = _need_cuda_f32("start", start.reshape(start.shape[0], 2), 2)
            if start.shape[0] != B:
                raise ValueError("start must be [B,2,1,1]")
        out = torch.empty_like(inp)
        with torch.cuda.device(inp.device):
            _call("fs_warp2d_fwd", inp.data_ptr(), flow.data_ptr(), _ptr(start),
                                                out.data_ptr(), B, C, H, W, mode, int(with_mask),
                                                _stream(inp))
        ctx.save_for_backward(inp, flow, start)
        ctx.mode, ctx.with_mask = mode, int(with_mask)
        return out

    @staticmethod
    def backward(ctx, gout):
        inp, flow, start = ctx.saved_tensors
        need_in, need_flow = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_in or need_flow):
            return None, None, None, None, None
        gout = gout.contiguous()
        B, C, H, W = inp.shape
        gin = torch.zeros_like(inp) if need_in else None
        gflow = torch.empty_like(flow) if need_flow else None
        with torch.cuda.device(inp.device):
            _call("fs_warp2d_bwd", inp.data_ptr(), flow.data_ptr(), _ptr(start),
                                                gout.data_ptr(), _ptr(gin), _ptr(gflow),
                                                B, C, H, W, ctx.mode, ctx.with_mask,
                                                _stream(inp))
        return gin, gflow, None, None, None


def warp2d(tenInput, tenFlow):
    """a1: Flow-2D/model/warplayer.py:7-26 (border pad, align_corners=True)."""
    return _Warp2D.apply(tenInput, tenFlow, None, WARP2D_RIFE, 0)


def warp2d_pwc(x, flow, with_mask):
    """a5/a6: pwc_modules.WarpingLayer_no_div (with_mask) / tools.torch_warp (no mask)."""
    return _Warp2D.apply(x, flow, None, WARP2D_PWC, 1 if with_mask else 0)


def warp2d_photo(frame, flow):
    """a11: the `backwrd_warp` closure of Flow-2D/model/RIFE.py:244-262."""
    return _Warp2D.apply(frame, flow, None, WARP2D_PHOTO, 0)


def warp2d_dilated(I, flow, start=None):
    """a7: tools.boundary_dilated_warp.warp_im (UPFlow/utils/tools.py:533-541)."""
    return _Warp2D.apply(I, flow, start, WARP2D_DILATED, 0)


# --------------------------------------------------------------------------------------------
# IFNet call site: both frames warped by the two halves of one flow tensor, one launch
# (Flow-3D/model/IFNet.py:190-191, Flow-2D/model/IFNet.py:191-192)
# --------------------------------------------------------------------------------------------
class _WarpPair(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img0, img1, flow):
        nd = flow.dim() - 2
        if nd not in (2, 3):
            raise ValueError("flow must be [B,4,H,W] or [B,6,D,H,W], got %s" % (tuple(flow.shape),))
        img0 = _need_cuda_f32("img0", img0, nd + 2)
        img1 = _need_cuda_f32("img1", img1, nd + 2)
        flow = _need_cuda_f32("flow", flow, nd + 2)
        if img0.shape != img1.shape:
            raise ValueError("img0 %s and img1 %s differ" % (tuple(img0.shape), tuple(img1.shape)))
        if flow.shape[0] != img0.shape[0] or flow.shape[1] != 2 * nd:
            raise ValueError("flow %s does not match the images %s" %
                             (tuple(flow.shape), tuple(img0.shape)))
        if nd == 2 and flow.shape[2:] != img0.shape[2:]:
            raise ValueError("2-D flow %s and images %s must have the same extent" %
                             (tuple(flow.shape), tuple(img0.shape)))
        if not (img0.device == img1.device == flow.device):
            raise ValueError("operands are on different devices")
        oshape = tuple(img0.shape[:2]) + tuple(flow.shape[2:])
        out0, out1 = img0.new_empty(oshape), img1.new_empty(oshape)
        with torch.cuda.device(flow.device):
            if nd == 3:
                B, C = img0.shape[:2]
                D, H, W = flow.shape[2:]
                _call("fs_warp3d_pair_fwd", img0.data_ptr(), img1.data_ptr(), flow.data_ptr(),
                      out0.data_ptr(), out1.data_ptr(), B, C, _in_dhw(img0, flow), D, H, W,
                      _stream(flow))
            else:
                B, C, H, W = img0.shape
                _call("fs_warp2d_pair_fwd", img0.data_ptr(), img1.data_ptr(), flow.data_ptr(),
                                                out0.data_ptr(), out1.data_ptr(), B, C, H, W,
                                                WARP2D_RIFE, _stream(flow))
        ctx.save_for_backward(img0, img1, flow)
        return out0, out1

    @staticmethod
    def backward(ctx, g0, g1):
        img0, img1, flow = ctx.saved_tensors
        need_img = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        need_flow = ctx.needs_input_grad[2]
        if not (need_img or need_flow):
            return None, None, None
        g0, g1 = g0.contiguous(), g1.contiguous()
        gi0 = torch.zeros_like(img0) if need_img else None
        gi1 = torch.zeros_like(img1) if need_img else None
        gflow = torch.empty_like(flow) if need_flow else None
        with torch.cuda.device(flow.device):
            if flow.dim() == 5:
                B, C = img0.shape[:2]
                D, H, W = flow.shape[2:]
                _call("fs_warp3d_pair_bwd", img0.data_ptr(), img1.data_ptr(), flow.data_ptr(),
                      g0.data_ptr(), g1.data_ptr(), _ptr(gi0), _ptr(gi1), _ptr(gflow), B, C,
                      _in_dhw(img0, flow), D, H, W, _stream(flow))
            else:
                B, C, H, W = img0.shape
                _call("fs_warp2d_pair_bwd", img0.data_ptr(), img1.data_ptr(), flow.data_ptr(),
                                                g0.data_ptr(), g1.data_ptr(), _ptr(gi0), _ptr(gi1),
                                                _ptr(gflow), B, C, H, W, WARP2D_RIFE, _stream(flow))
        return (gi0 if ctx.needs_input_grad[0] else None, gi1 if ctx.needs_input_grad[1] else None,
                gflow)


def warp_pair(img0, img1, flow):
    """(warp(img0, flow[:, :nd]), warp(img1, flow[:, nd:2nd])) in one launch; nd = 2 or 3."""
    return _WarpPair.apply(img0, img1, flow)


# --------------------------------------------------------------------------------------------
# a11: photometric term of Flow-2D Model.update (Flow-2D/model/RIFE.py:190-191, 244-279)
# --------------------------------------------------------------------------------------------
def rife2d_photometric(flow4, merged, img0, img1):
    """loss_photo = mean over the two directions of sum_pixels ((warp(merged) - frame)^2 + eps^2)^0.25
    / 3 / B, with `backwrd_warp`'s half-pixel-shifted zero-padded sampling done by the HIP kernel.
    The reference's two F.interpolate calls (:248, :269) resize to the size the tensors already
    have and are exact identities."""
    def term(flow2, frame):
        w = warp2d_photo(merged, flow2)
        p = torch.pow(torch.pow(w - frame, 2) + 1.e-9 ** 2, 0.25)
        return torch.sum(torch.sum(p, dim=1) / 3) / frame.size(0)

    return (term(flow4[:, 2:4], img0) + term(flow4[:, :2], img1)) / 2
