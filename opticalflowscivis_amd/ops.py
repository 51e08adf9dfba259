"""torch.autograd bindings of the HIP hot-path kernels (C-ABI in include/flowsci_hip.h).

Every op validates its operands on the host (dtype fp32, CUDA device, matching shapes; operands
are made contiguous) and raises ValueError on a mismatch, then launches on PyTorch's current
stream of the operand's device.  Nothing here computes on the CPU: without a GPU build of
libflowsci_hip.so these functions raise.
"""
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


# --------------------------------------------------------------------------------------------
# a2: Flow-3D/model/warplayer.py:9-41
# --------------------------------------------------------------------------------------------
class _Warp3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, flow):
        inp = _need_cuda_f32("tenInput", inp, 5)
        flow = _need_cuda_f32("tenFlow", flow, 5)
        B, C, D, H, W = inp.shape
        if tuple(flow.shape) != (B, 3, D, H, W):
            raise ValueError("tenFlow must be [B,3,D,H,W] matching tenInput %s, got %s" %
                             (tuple(inp.shape), tuple(flow.shape)))
        if inp.device != flow.device:
            raise ValueError("tenInput and tenFlow are on different devices")
        out = torch.empty_like(inp)
        with torch.cuda.device(inp.device):
            _lib.check(_lib.lib().fs_warp3d_fwd(inp.data_ptr(), flow.data_ptr(), out.data_ptr(),
                                                B, C, D, H, W, _stream(inp)), "fs_warp3d_fwd")
        ctx.save_for_backward(inp, flow)
        return out

    @staticmethod
    def backward(ctx, gout):
        inp, flow = ctx.saved_tensors
        need_in, need_flow = ctx.needs_input_grad
        if not (need_in or need_flow):
            return None, None
        gout = gout.contiguous()
        B, C, D, H, W = inp.shape
        gin = torch.zeros_like(inp) if need_in else None
        gflow = torch.empty_like(flow) if need_flow else None
        with torch.cuda.device(inp.device):
            _lib.check(_lib.lib().fs_warp3d_bwd(inp.data_ptr(), flow.data_ptr(), gout.data_ptr(),
                                                _ptr(gin), _ptr(gflow), B, C, D, H, W,
                                                _stream(inp)), "fs_warp3d_bwd")
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
            _lib.check(_lib.lib().fs_warp2d_fwd(inp.data_ptr(), flow.data_ptr(), _ptr(start),
                                                out.data_ptr(), B, C, H, W, mode, int(with_mask),
                                                _stream(inp)), "fs_warp2d_fwd")
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
            _lib.check(_lib.lib().fs_warp2d_bwd(inp.data_ptr(), flow.data_ptr(), _ptr(start),
                                                gout.data_ptr(), _ptr(gin), _ptr(gflow),
                                                B, C, H, W, ctx.mode, ctx.with_mask,
                                                _stream(inp)), "fs_warp2d_bwd")
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
