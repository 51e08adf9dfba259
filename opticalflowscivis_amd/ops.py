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


def _need_cuda_f32_strided(name, t, ndim):
    """_need_cuda_f32 without the `.contiguous()`: for operands whose strides are handed to the kernel."""
    if not isinstance(t, torch.Tensor):
        raise ValueError("%s must be a tensor" % name)
    if t.dtype != torch.float32 or t.dim() != ndim or not t.is_cuda:
        raise ValueError("%s must be a %d-D float32 GPU tensor, got %s %s on %s" %
                         (name, ndim, t.dtype, tuple(t.shape), t.device))
    return t


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
_timing_only = None


def enable_kernel_timing(on=True, only=None):
    """Start (on=True: clears previous records) or stop collecting (start, end) event pairs.  `only`: an
    iterable of entry-point names -- every other launch goes out without events (bench.py's timed region
    records just the dominant entry point, so that the headline time carries no profiling overhead)."""
    global _timing, _timing_only
    _timing = {} if on else None
    _timing_only = frozenset(only) if (on and only is not None) else None


def kernel_timings():
    """{entry point: [(ms, algorithmic HBM bytes, matrix-core flops EXECUTED, flops of the direct formulation, kernel
    symbol or None) per launch]} recorded so far (synchronises).  The two flop counts differ for the Winograd
    convolutions only (1/3, 1/2); the kernel symbol is known where the entry point's dispatch can be asked for it."""
    if _timing is None:
        return {}
    torch.cuda.synchronize()
    return {k: [(a.elapsed_time(b), nb, fl, fe, sym) for a, b, nb, fl, fe, sym in v] for k, v in _timing.items()}


def _call(name, *args, algo_bytes=0, algo_flops=0, record_as=None, equiv_flops=None, kernel=None):
    """Launch a C-ABI entry point.  `algo_bytes` / `algo_flops` = compulsory HBM bytes / useful flops of
    this launch, only used by the optional timing records (`record_as`: file the record
    under another entry point's name -- the *_prelu variants are the same kernels with one more store)."""
    fn = getattr(_lib.lib(), name)
    if _timing is None or (_timing_only is not None and (record_as or name) not in _timing_only):
        _lib.check(fn(*args), name)
        return
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    code = fn(*args)
    e1.record()
    _lib.check(code, name)
    _timing.setdefault(record_as or name, []).append((e0, e1, algo_bytes, algo_flops,
                                                      algo_flops if equiv_flops is None else equiv_flops, kernel))


def _call_rc(name, *args, algo_bytes=0, algo_flops=0, record_as=None, allow=(), equiv_flops=None, kernel=None):
    """_call for entry points that may answer with a status in `allow` (FS_ERR_UNSUPPORTED: "no such kernel for
    this shape, take the unfused path"): returns the status instead of raising on those."""
    fn = getattr(_lib.lib(), name)
    timed = not (_timing is None or (_timing_only is not None and (record_as or name) not in _timing_only))
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = fn(*args)
    if timed:
        e1.record()
        if rc == 0:
            _timing.setdefault(record_as or name, []).append((e0, e1, algo_bytes, algo_flops,
                                                              algo_flops if equiv_flops is None else equiv_flops, kernel))
    if rc not in allow:
        _lib.check(rc, name)
    return rc


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


def _empty_like_graph(shape, *tensors):
    """Empty-batch result: zeros of `shape` (numel 0) that still hang on the inputs' autograd graph, as
    `F.grid_sample` / `Corr_pyTorch` return for B = 0.  No kernel is launched for an empty batch."""
    out = tensors[0].new_zeros(shape)
    for t in tensors:
        if t is not None and t.requires_grad:
            out = out + 0 * t.sum()
    return out


def warp3d(tenInput, tenFlow):
    """Trilinear backward warp with the reference's axis-rotating grid (Flow-3D warplayer.warp)."""
    if tenInput.dim() == 5 and tenFlow.dim() == 5 and tenInput.shape[0] == 0 and tenFlow.shape[0] == 0:
        return _empty_like_graph((0, tenInput.shape[1]) + tuple(tenFlow.shape[2:]), tenInput, tenFlow)
    return _Warp3D.apply(tenInput, tenFlow)


# --------------------------------------------------------------------------------------------
# 2-D warps: a1, a5, a6, a7, a11
# --------------------------------------------------------------------------------------------
def _in_hw(inp, flow):
    """Host int[2] with the sampled image's extent, or None when it equals the flow's."""
    if tuple(inp.shape[2:]) == tuple(flow.shape[2:]):
        return None
    return (ctypes.c_int * 2)(*inp.shape[2:])


class _Warp2D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, flow, start, mode, with_mask):
        inp = _need_cuda_f32("input", inp, 4)
        flow = _need_cuda_f32("flow", flow, 4)
        B, C = inp.shape[:2]
        H, W = flow.shape[2:]  # the output takes the flow's extent
        same = tuple(flow.shape[2:]) == tuple(inp.shape[2:])
        if flow.shape[0] != B or flow.shape[1] != 2 or not (same or mode == WARP2D_RIFE):
            # only the RIFE warp defines input extent != flow extent (Flow-2D/model/warplayer.py:10-20)
            raise ValueError("flow must be [B,2,H,W] matching input %s, got %s" %
                             (tuple(inp.shape), tuple(flow.shape)))
        if inp.device != flow.device:
            raise ValueError("input and flow are on different devices")
        if start is not None:
            start = _need_cuda_f32("start", start.reshape(start.shape[0], 2), 2)
            if start.shape[0] != B:
                raise ValueError("start must be [B,2,1,1]")
        out = inp.new_empty((B, C, H, W))
        with torch.cuda.device(inp.device):
            _call("fs_warp2d_fwd", inp.data_ptr(), flow.data_ptr(), _ptr(start),
                  out.data_ptr(), B, C, _in_hw(inp, flow), H, W, mode, int(with_mask), _stream(inp),
                  algo_bytes=4 * flow.numel() + 8 * out.numel())  # flow 8 + per channel gather 4 + store 4
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
        B, C = inp.shape[:2]
        H, W = flow.shape[2:]
        gin = torch.zeros_like(inp) if need_in else None
        gflow = torch.empty_like(flow) if need_flow else None
        with torch.cuda.device(inp.device):
            _call("fs_warp2d_bwd", inp.data_ptr(), flow.data_ptr(), _ptr(start),
                  gout.data_ptr(), _ptr(gin), _ptr(gflow), B, C, _in_hw(inp, flow), H, W, ctx.mode,
                  ctx.with_mask, _stream(inp),
                  algo_bytes=4 * flow.numel() * (2 if need_flow else 1) + 4 * gout.numel() * (3 if need_in else 2))
        return gin, gflow, None, None, None


def _empty2d(x, flow):
    return x.dim() == 4 and flow.dim() == 4 and x.shape[0] == 0 and flow.shape[0] == 0


def warp2d(tenInput, tenFlow):
    """a1: Flow-2D/model/warplayer.py:7-26 (border pad, align_corners=True)."""
    if _empty2d(tenInput, tenFlow):
        return _empty_like_graph(tuple(tenInput.shape[:2]) + tuple(tenFlow.shape[2:]), tenInput, tenFlow)
    return _Warp2D.apply(tenInput, tenFlow, None, WARP2D_RIFE, 0)


def warp2d_pwc(x, flow, with_mask):
    """a5/a6: pwc_modules.WarpingLayer_no_div (with_mask) / tools.torch_warp (no mask)."""
    if _empty2d(x, flow):
        return _empty_like_graph(x.shape, x, flow)
    return _Warp2D.apply(x, flow, None, WARP2D_PWC, 1 if with_mask else 0)


def occ_check2d(flow_f, flow_b, alpha1, alpha2, scale=1, obj_out_all="obj"):
    """§8f.2: UPFlow/utils/tools.py:560-719 `occ_check_model.__call__` as one launch -- both
    torch_warp calls, the L1 magnitudes, the threshold test, the outgoing masks and their
    combination.  Returns (occ_fw, occ_bw) [B,1,H,W] float {0,1}; no gradient (bool in the
    reference)."""
    mode = {"all": 0, "obj": 1, "out": 2}[obj_out_all]
    flow_f = _need_cuda_f32("flow_f", flow_f.detach(), 4)
    flow_b = _need_cuda_f32("flow_b", flow_b.detach(), 4)
    if flow_f.shape != flow_b.shape or flow_f.shape[1] != 2:
        raise ValueError("flows must both be [B,2,H,W], got %s / %s" % (tuple(flow_f.shape), tuple(flow_b.shape)))
    B, _, H, W = flow_f.shape
    occ_f = torch.empty((B, 1, H, W), device=flow_f.device, dtype=torch.float32)
    occ_b = torch.empty_like(occ_f)
    with torch.cuda.device(flow_f.device):
        _call("fs_occ_check2d", flow_f.data_ptr(), flow_b.data_ptr(), occ_f.data_ptr(), occ_b.data_ptr(),
              B, H, W, float(alpha1), float(alpha2 / scale), mode, _stream(flow_f),
              algo_bytes=24 * B * H * W)
    return occ_f, occ_b


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
def _check_pair(img0, img1, flow):
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
    if not (img0.device == img1.device == flow.device):
        raise ValueError("operands are on different devices")
    return img0, img1, flow, nd


_W3_SYMBOLS = {(0, 1): "warp3d_rc_kernel<false, 2, 6, 0>", (1, 1): "warp3d_rc_kernel<true, 4, 5, 0>"}


def _warp3d_symbol(img0, img1, flow, backward, with_grad_in=False):
    """Kernel symbol a trilinear-warp pair launch of this geometry dispatches to, where ops.py can name it for bench.py's
    per-symbol records: the round-5 row-cache ring kernels (fs_warp3d_kernel_id; asked only while launches are being
    timed).  None = the gather kernels (several instantiations)."""
    if _timing is None:
        return None
    B, C = img0.shape[:2]
    D, H, W = flow.shape[2:]
    kid = int(_lib.lib().fs_warp3d_kernel_id(img0.data_ptr(), img1.data_ptr(), flow.data_ptr(), B, C, _in_dhw(img0, flow),
                                             D, H, W, int(backward), int(with_grad_in)))
    return _W3_SYMBOLS.get((int(backward), kid))


def _pair_forward(img0, img1, flow, nd):
    oshape = tuple(img0.shape[:2]) + tuple(flow.shape[2:])  # the warps take the flow's extent
    out0, out1 = img0.new_empty(oshape), img1.new_empty(oshape)
    B, C = img0.shape[:2]
    with torch.cuda.device(flow.device):
        if nd == 3:
            D, H, W = flow.shape[2:]
            _call("fs_warp3d_pair_fwd", img0.data_ptr(), img1.data_ptr(), flow.data_ptr(),
                  out0.data_ptr(), out1.data_ptr(), B, C, _in_dhw(img0, flow), D, H, W,
                  _stream(flow), algo_bytes=4 * flow.numel() + 8 * out0.numel() * 2,
                  kernel=_warp3d_symbol(img0, img1, flow, 0))
        else:
            H, W = flow.shape[2:]
            _call("fs_warp2d_pair_fwd", img0.data_ptr(), img1.data_ptr(), flow.data_ptr(),
                  out0.data_ptr(), out1.data_ptr(), B, C, _in_hw(img0, flow), H, W,
                  WARP2D_RIFE, _stream(flow), algo_bytes=4 * flow.numel() + 8 * out0.numel() * 2)
    return out0, out1


def _flow_addends(grads, flow):
    """The (pointer, batch stride) pairs fs_*_bwd*3 take for up to three gradients of a [B,C,D,H,W] flow:
    contiguous tensors or channel slices of wider contiguous ones (what `torch.cat`'s backward hands out) go
    through as they are; anything else is made contiguous.  Returns (flat argument list, tensors to keep alive,
    bytes read)."""
    vol = flow.shape[2] * flow.shape[3] * flow.shape[4]
    want_inner = (vol, flow.shape[3] * flow.shape[4], flow.shape[4], 1)
    keep, args, nbytes = [], [], 0
    for g in grads:
        if g is None:
            continue
        g = _need_cuda_f32_strided("grad_flow", g, 5)
        if tuple(g.shape) != tuple(flow.shape):
            raise ValueError("gradient %s does not match the flow %s" % (tuple(g.shape), tuple(flow.shape)))
        if tuple(g.stride()[1:]) != want_inner or g.stride(0) < flow.shape[1] * vol or g.data_ptr() % 16 or g.stride(0) % 4:
            g = g.contiguous()
        keep.append(g)
        args += [g.data_ptr(), int(g.stride(0))]
        nbytes += 4 * g.numel()
    if len(keep) > 3:  # (never in IFNet: three consumers per flow)
        extra = keep[3]
        for g in keep[4:]:
            extra = extra + g
        keep = keep[:2] + [keep[2] + extra]
        args = []
        for g in keep:
            g = g.contiguous()
            args += [g.data_ptr(), int(g.stride(0))]
    while len(args) < 6:
        args += [None, 0]
    return args, keep, nbytes


def _gout_strided(g, like):
    """(tensor, batch stride or 0) of a warped frame's gradient for fs_*_bwd*3: dense, or a channel slice of a wider
    contiguous tensor (torch.cat's backward) handed to the kernel with its batch stride; anything else is copied."""
    vol = like.shape[2] * like.shape[3] * like.shape[4]
    inner = (vol, like.shape[3] * like.shape[4], like.shape[4], 1)
    if g.is_contiguous():
        return g, 0
    if (g.dim() == 5 and tuple(g.stride()[1:]) == inner and g.stride(0) >= g.shape[1] * vol and g.stride(0) % 4 == 0
            and g.data_ptr() % 16 == 0):
        return g, int(g.stride(0))
    return g.contiguous(), 0


def _pair_backward(img0, img1, flow, g0, g1, need_img, need_flow, gflow_add=None):
    """(grad_img0, grad_img1, grad_flow); `gflow_add` (3-D only): a gradient, or a list of up to three, reaching
    the flow from its other consumers, summed into grad_flow by the same launch."""
    if flow.dim() == 5 and gflow_add is not None:
        (g0, s0), (g1, s1) = _gout_strided(g0, flow), _gout_strided(g1, flow)
    else:
        g0, g1, s0, s1 = g0.contiguous(), g1.contiguous(), 0, 0
    gi0 = torch.zeros_like(img0) if need_img else None
    gi1 = torch.zeros_like(img1) if need_img else None
    B, C = img0.shape[:2]
    gflow = None
    with torch.cuda.device(flow.device):
        if flow.dim() == 5:
            D, H, W = flow.shape[2:]
            nb = 8 * flow.numel() + 8 * g0.numel() * 2 + (8 * img0.numel() if need_img else 0)
            if gflow_add is not None:
                adds = gflow_add if isinstance(gflow_add, (list, tuple)) else [gflow_add]
                aargs, keep, abytes = _flow_addends(adds, flow)
                gflow = torch.empty_like(flow)
                _call("fs_warp3d_pair_bwd_acc3", img0.data_ptr(), img1.data_ptr(), flow.data_ptr(),
                      g0.data_ptr(), s0, g1.data_ptr(), s1, _ptr(gi0), _ptr(gi1), *aargs, gflow.data_ptr(),
                      B, C, _in_dhw(img0, flow), D, H, W, _stream(flow), algo_bytes=nb + abytes,
                      kernel=_warp3d_symbol(img0, img1, flow, 1, need_img))
                del keep
            else:
                gflow = torch.empty_like(flow) if need_flow else None
                _call("fs_warp3d_pair_bwd", img0.data_ptr(), img1.data_ptr(), flow.data_ptr(),
                      g0.data_ptr(), g1.data_ptr(), _ptr(gi0), _ptr(gi1), _ptr(gflow), B, C,
                      _in_dhw(img0, flow), D, H, W, _stream(flow), algo_bytes=nb,
                      kernel=_warp3d_symbol(img0, img1, flow, 1, need_img))
        else:
            H, W = flow.shape[2:]
            gflow = torch.empty_like(flow) if need_flow else None
            _call("fs_warp2d_pair_bwd", img0.data_ptr(), img1.data_ptr(), flow.data_ptr(),
                  g0.data_ptr(), g1.data_ptr(), _ptr(gi0), _ptr(gi1), _ptr(gflow), B, C,
                  _in_hw(img0, flow), H, W, WARP2D_RIFE, _stream(flow),
                  algo_bytes=4 * flow.numel() * (2 if need_flow else 1) + 8 * g0.numel() * (3 if need_img else 2))
            if gflow_add is not None and gflow is not None:
                for g in (gflow_add if isinstance(gflow_add, (list, tuple)) else [gflow_add]):
                    if g is not None:
                        gflow = gflow + g
    return gi0, gi1, gflow


class _WarpPair(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img0, img1, flow):
        img0, img1, flow, nd = _check_pair(img0, img1, flow)
        out0, out1 = _pair_forward(img0, img1, flow, nd)
        ctx.save_for_backward(img0, img1, flow)
        return out0, out1

    @staticmethod
    def backward(ctx, g0, g1):
        img0, img1, flow = ctx.saved_tensors
        need_img = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        need_flow = ctx.needs_input_grad[2]
        if not (need_img or need_flow):
            return None, None, None
        gi0, gi1, gflow = _pair_backward(img0, img1, flow, g0, g1, need_img, need_flow)
        return (gi0 if ctx.needs_input_grad[0] else None, gi1 if ctx.needs_input_grad[1] else None,
                gflow)


class _WarpPairAcc(torch.autograd.Function):
    """(w0, w1, flow_a, flow_b, flow_c) with flow_* three aliases of `flow`: the caller hands ONE alias to each
    of the flow's other consumers (next block's input, next block's accumulation, distillation), so autograd
    delivers their gradients to this node one by one, and the warp's backward launch adds all of them to its own
    result (fs_warp3d_pair_bwd_acc3) -- no autograd `add` over the full-size flow, and the 6-channel slice of the
    next block's 11-channel input gradient is read in place."""

    @staticmethod
    def forward(ctx, img0, img1, flow):
        img0, img1, flow_c, nd = _check_pair(img0, img1, flow)
        out0, out1 = _pair_forward(img0, img1, flow_c, nd)
        ctx.save_for_backward(img0, img1, flow_c)
        ctx.set_materialize_grads(False)
        return out0, out1, flow_c.view_as(flow_c), flow_c.view_as(flow_c), flow_c.view_as(flow_c)

    @staticmethod
    def backward(ctx, g0, g1, ga, gb, gc):
        img0, img1, flow = ctx.saved_tensors
        need_img = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        need_flow = ctx.needs_input_grad[2]
        if not (need_img or need_flow):
            return None, None, None
        adds = [g for g in (ga, gb, gc) if g is not None]
        gflow_out = adds if adds else None
        if g0 is None and g1 is None:
            tot = None
            for g in adds:
                tot = g if tot is None else tot + g
            return None, None, (tot if need_flow else None)
        oshape = tuple(img0.shape[:2]) + tuple(flow.shape[2:])
        g0 = flow.new_zeros(oshape) if g0 is None else g0
        g1 = flow.new_zeros(oshape) if g1 is None else g1
        gi0, gi1, gflow = _pair_backward(img0, img1, flow, g0, g1, need_img, need_flow,
                                         gflow_out if need_flow else None)
        return (gi0 if ctx.needs_input_grad[0] else None, gi1 if ctx.needs_input_grad[1] else None,
                gflow)


def warp_pair(img0, img1, flow):
    """(warp(img0, flow[:, :nd]), warp(img1, flow[:, nd:2nd])) in one launch; nd = 2 or 3."""
    if img0.shape[0] == 0 and img1.shape[0] == 0 and flow.shape[0] == 0 and flow.dim() in (4, 5):
        shape = (0, img0.shape[1]) + tuple(flow.shape[2:])
        return _empty_like_graph(shape, img0, flow), _empty_like_graph(shape, img1, flow)
    return _WarpPair.apply(img0, img1, flow)


def warp_pair_acc(img0, img1, flow):
    """warp_pair that also returns the flow for its OTHER consumers: (w0, w1, (flow_a, flow_b, flow_c)).  Use one
    alias per downstream consumer instead of `flow`; the gradients arriving there are folded into the warp's
    backward launch."""
    if img0.shape[0] == 0 and img1.shape[0] == 0 and flow.shape[0] == 0 and flow.dim() in (4, 5):
        return warp_pair(img0, img1, flow) + ((flow, flow, flow),)
    w0, w1, fa, fb, fc = _WarpPairAcc.apply(img0, img1, flow)
    return w0, w1, (fa, fb, fc)


class _UpsampleWarpPair(torch.autograd.Function):
    """flow = prev + scale * trilinear_upsample(delta, factor); (w0, w1) = warp pair with that
    flow -- one kernel forward (fs_upsample_warp3d_pair_fwd), and backward one warp launch that folds in the
    gradient reaching `flow` from its other consumers plus the separable up-sampling adjoint."""

    @staticmethod
    def forward(ctx, img0, img1, delta, prev, factor, scale):
        delta = _need_cuda_f32("delta", delta, 5)
        img0 = _need_cuda_f32("img0", img0, 5)
        img1 = _need_cuda_f32("img1", img1, 5)
        B, C = img0.shape[:2]
        if delta.shape[0] != B or delta.shape[1] != 6 or img0.shape != img1.shape:
            raise ValueError("delta must be [B,6,d,h,w] for images %s / %s, got %s" %
                             (tuple(img0.shape), tuple(img1.shape), tuple(delta.shape)))
        Ds, Hs, Ws = delta.shape[2:]
        full = (B, 6, Ds * factor, Hs * factor, Ws * factor)
        if prev is not None:
            prev = _need_cuda_f32("prev", prev, 5)
            if tuple(prev.shape) != full:
                raise ValueError("prev %s must have the up-sampled shape %s" % (tuple(prev.shape), full))
        flow = delta.new_empty(full)
        out0, out1 = img0.new_empty((B, C) + full[2:]), img0.new_empty((B, C) + full[2:])
        with torch.cuda.device(delta.device):
            _call("fs_upsample_warp3d_pair_fwd", img0.data_ptr(), img1.data_ptr(), delta.data_ptr(), _ptr(prev),
                  flow.data_ptr(), out0.data_ptr(), out1.data_ptr(), B, C, _in_dhw(img0, flow), Ds, Hs, Ws,
                  int(factor), float(scale), _stream(delta),
                  algo_bytes=4 * (delta.numel() + flow.numel() * (2 if prev is not None else 1)) + 8 * out0.numel() * 2)
        ctx.save_for_backward(img0, img1, flow)
        ctx.cfg = (tuple(delta.shape), int(factor), float(scale), prev is not None)
        ctx.set_materialize_grads(False)
        return flow, flow.view_as(flow), flow.view_as(flow), out0, out1

    @staticmethod
    def backward(ctx, gfa, gfb, gfc, g0, g1):
        img0, img1, flow = ctx.saved_tensors
        dshape, factor, scale, has_prev = ctx.cfg
        need_delta, need_prev = ctx.needs_input_grad[2], has_prev and ctx.needs_input_grad[3]
        need_img = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        if not (need_delta or need_prev or need_img):
            return (None,) * 6
        B, C = img0.shape[:2]
        Ds, Hs, Ws = dshape[2:]
        D, H, W = flow.shape[2:]
        adds = [g for g in (gfa, gfb, gfc) if g is not None]
        if g0 is None and g1 is None and not adds:
            return (None,) * 6
        oshape = (B, C, D, H, W)
        gi0 = gi1 = None
        if need_img and not (g0 is None and g1 is None):
            # the frames carry no gradient in IFNet, so the fused launch below has no image-gradient output; a caller
            # whose frames do require grad gets them from the plain pair backward (scatter with float atomics)
            gi0, gi1, _ = _pair_backward(img0, img1, flow, flow.new_zeros(oshape) if g0 is None else g0,
                                         flow.new_zeros(oshape) if g1 is None else g1, True, False)
            gi0 = gi0 if ctx.needs_input_grad[0] else None
            gi1 = gi1 if ctx.needs_input_grad[1] else None
        if not (need_delta or need_prev):
            return gi0, gi1, None, None, None, None
        g0, s0 = (flow.new_zeros(oshape), 0) if g0 is None else _gout_strided(g0, flow)
        g1, s1 = (flow.new_zeros(oshape), 0) if g1 is None else _gout_strided(g1, flow)
        aargs, keep, abytes = _flow_addends(adds, flow)
        gtot = torch.empty_like(flow)
        gdelta = flow.new_empty(dshape)
        ws = flow.new_empty(B * 6 * (D * H * Ws + D * Hs * Ws))
        with torch.cuda.device(flow.device):
            _call("fs_upsample_warp3d_pair_bwd3", img0.data_ptr(), img1.data_ptr(), flow.data_ptr(), g0.data_ptr(),
                  s0, g1.data_ptr(), s1, *aargs, gtot.data_ptr(), gdelta.data_ptr(), ws.data_ptr(), B, C,
                  _in_dhw(img0, flow), Ds, Hs, Ws, factor, scale, _stream(flow),
                  algo_bytes=8 * flow.numel() + 8 * g0.numel() * 2 + abytes + 4 * (flow.numel() + gdelta.numel()))
        del keep
        return gi0, gi1, (gdelta if need_delta else None), (gtot if need_prev else None), None, None


def upsample_warp_pair(img0, img1, delta, prev, factor, scale=None):
    """((flow_a, flow_b, flow_c), w0, w1) with flow = prev + scale * F.interpolate(delta, scale_factor=factor,
    trilinear, align_corners=False) (prev may be None; scale defaults to factor, Flow-3D/model/IFNet.py:118) and
    (w0, w1) = warp_pair(img0, img1, flow): IFBlock's flow up-scaling, the running-flow accumulation and the
    two backward warps in one launch.  flow_* are three aliases of the flow, one per downstream consumer (see
    _WarpPairAcc): their gradients are summed inside the backward warp launch."""
    fa, fb, fc, w0, w1 = _UpsampleWarpPair.apply(img0, img1, delta, prev, int(factor),
                                                 float(factor if scale is None else scale))
    return (fa, fb, fc), w0, w1


# --------------------------------------------------------------------------------------------
# a11: photometric term of Flow-2D Model.update (Flow-2D/model/RIFE.py:190-191, 244-279)
# --------------------------------------------------------------------------------------------
def rife2d_photometric(flow4, merged, img0, img1):
    """loss_photo = mean over the two directions of sum_pixels ((warp(merged) - frame)^2 + eps^2)^0.25
    / 3 / B, with `backwrd_warp`'s half-pixel-shifted zero-padded sampling done by the HIP kernel.
    The reference's two F.interpolate calls (:248 merged -> the flow's extent, align_corners=True; :269
    frame -> the warped extent, align_corners=False) are exact identities when the extents agree -- every
    size that is a multiple of 16 -- and are skipped then; for extents where IFNet crops its outputs below
    the input (floor(n/4) % 4 == 0 and n % 4 != 0, e.g. H = 145..147) they are real bilinear resizes."""
    import torch.nn.functional as F
    if merged.shape[2:] != flow4.shape[2:]:
        merged = F.interpolate(merged, size=tuple(flow4.shape[2:]), mode='bilinear', align_corners=True)

    def term(flow2, frame):
        w = warp2d_photo(merged, flow2)
        if frame.shape[2:] != w.shape[2:]:
            frame = F.interpolate(frame, size=tuple(w.shape[2:]), mode='bilinear', align_corners=False)
        # charbonnier(x, 0.25, 1e-9) = (x^2 + 1e-18)^0.25, summed, / 3 / B: one fused penalty+reduction
        return robust_loss(w, frame, None, PEN_CHARBONNIER, 0.25, 1.e-9 ** 2, form="sum") / 3 / frame.size(0)

    return (term(flow4[:, 2:4], img0) + term(flow4[:, :2], img1)) / 2


# --------------------------------------------------------------------------------------------
# a3/a4: local-window correlation (UPFlow/model/correlation_package/correlation.py:8-45)
# --------------------------------------------------------------------------------------------
def corr2d_forward_into(input1, input2, output, max_displacement):
    """correlation_cuda.forward semantics: `output` is resized and filled in place."""
    input1 = _need_cuda_f32("input1", input1, 4)
    input2 = _need_cuda_f32("input2", input2, 4)
    if input1.shape != input2.shape or input1.device != input2.device:
        raise ValueError("input1 %s and input2 %s must match" %
                         (tuple(input1.shape), tuple(input2.shape)))
    B, C, H, W = input1.shape
    nd = 2 * max_displacement + 1
    output.resize_(B, nd * nd, H, W)
    with torch.cuda.device(input1.device):
        _call("fs_corr2d_fwd", input1.data_ptr(), input2.data_ptr(), output.data_ptr(), B, C, H, W,
              int(max_displacement), _stream(input1), algo_bytes=4 * (2 * input1.numel() + output.numel()),
              algo_flops=2 * output.numel() * C)
    return output


def corr2d_backward_into(input1, input2, grad_output, grad_input1, grad_input2, max_displacement):
    """correlation_cuda.backward semantics: grad tensors are resized and filled in place
    (pass None to skip one of them)."""
    input1 = _need_cuda_f32("input1", input1, 4)
    input2 = _need_cuda_f32("input2", input2, 4)
    grad_output = _need_cuda_f32("grad_output", grad_output, 4)
    B, C, H, W = input1.shape
    nd = 2 * max_displacement + 1
    if tuple(grad_output.shape) != (B, nd * nd, H, W):
        raise ValueError("grad_output must be %s, got %s" % ((B, nd * nd, H, W),
                                                             tuple(grad_output.shape)))
    for g in (grad_input1, grad_input2):
        if g is not None:
            g.resize_(B, C, H, W)
    with torch.cuda.device(input1.device):
        _call("fs_corr2d_bwd", input1.data_ptr(), input2.data_ptr(), grad_output.data_ptr(),
              _ptr(grad_input1), _ptr(grad_input2), B, C, H, W, int(max_displacement),
              _stream(input1), algo_bytes=4 * (2 * input1.numel() + grad_output.numel()) + 4 * input1.numel() * sum(
                  1 for g in (grad_input1, grad_input2) if g is not None),
              algo_flops=2 * grad_output.numel() * C * sum(1 for g in (grad_input1, grad_input2) if g is not None))


class _Corr2D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f1, f2, md):
        out = corr2d_forward_into(f1, f2, f1.new_empty(0), md)
        ctx.save_for_backward(f1.contiguous(), f2.contiguous())
        ctx.md = md
        return out

    @staticmethod
    def backward(ctx, gout):
        f1, f2 = ctx.saved_tensors
        g1 = torch.empty_like(f1) if ctx.needs_input_grad[0] else None
        g2 = torch.empty_like(f2) if ctx.needs_input_grad[1] else None
        if g1 is None and g2 is None:
            return None, None, None
        corr2d_backward_into(f1, f2, gout, g1, g2, ctx.md)
        return g1, g2, None


def corr2d(f1, f2, max_displacement=4):
    """Cost volume [B,(2md+1)^2,H,W] = channel-mean of f1 * shifted f2 (zero padded)."""
    if f1.dim() == 4 and f2.dim() == 4 and f1.shape[0] == 0 and f2.shape[0] == 0:
        nd = 2 * int(max_displacement) + 1
        return _empty_like_graph((0, nd * nd) + tuple(f1.shape[2:]), f1, f2)
    return _Corr2D.apply(f1, f2, int(max_displacement))


class _Corr2DNorm(torch.autograd.Function):
    """§8f.4: corr2d(normalize(f1), normalize(f2)) with per-(b,c)-plane moments, normalisation folded
    into the correlation kernels' tile loads (upflow.py:96-138 + correlation.py:26)."""

    @staticmethod
    def forward(ctx, f1, f2, md):
        f1 = _need_cuda_f32("f1", f1, 4)
        f2 = _need_cuda_f32("f2", f2, 4)
        if f1.shape != f2.shape:
            raise ValueError("f1 %s and f2 %s must match" % (tuple(f1.shape), tuple(f2.shape)))
        B, C, H, W = f1.shape
        if H * W < 2:
            raise ValueError("per-plane variance needs at least two pixels")
        st1, st2 = f1.new_empty(B * C, 2), f1.new_empty(B * C, 2)
        nd = 2 * md + 1
        out = f1.new_empty(B, nd * nd, H, W)
        with torch.cuda.device(f1.device):
            s = _stream(f1)
            _call("fs_plane_moments", f1.data_ptr(), st1.data_ptr(), B * C, H * W, s, algo_bytes=4 * f1.numel())
            _call("fs_plane_moments", f2.data_ptr(), st2.data_ptr(), B * C, H * W, s, algo_bytes=4 * f2.numel())
            _call("fs_corr2d_norm_fwd", f1.data_ptr(), f2.data_ptr(), st1.data_ptr(), st2.data_ptr(),
                  out.data_ptr(), B, C, H, W, md, s, algo_bytes=4 * (2 * f1.numel() + out.numel()))
        ctx.save_for_backward(f1, f2, st1, st2)
        ctx.md = md
        return out

    @staticmethod
    def backward(ctx, gout):
        f1, f2, st1, st2 = ctx.saved_tensors
        B, C, H, W = f1.shape
        gout = _need_cuda_f32("grad_output", gout, 4)
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need1 or need2):
            return None, None, None
        gn1 = torch.empty_like(f1) if need1 else None
        gn2 = torch.empty_like(f2) if need2 else None
        g1 = g2 = None
        with torch.cuda.device(f1.device):
            s = _stream(f1)
            _call("fs_corr2d_norm_bwd", f1.data_ptr(), f2.data_ptr(), st1.data_ptr(), st2.data_ptr(),
                  gout.data_ptr(), _ptr(gn1), _ptr(gn2), B, C, H, W, ctx.md, s,
                  algo_bytes=4 * (2 * f1.numel() + gout.numel()))
            if need1:
                g1 = torch.empty_like(f1)
                _call("fs_plane_norm_bwd", f1.data_ptr(), st1.data_ptr(), gn1.data_ptr(), g1.data_ptr(),
                      B * C, H * W, s, algo_bytes=12 * f1.numel())
            if need2:
                g2 = torch.empty_like(f2)
                _call("fs_plane_norm_bwd", f2.data_ptr(), st2.data_ptr(), gn2.data_ptr(), g2.data_ptr(),
                      B * C, H * W, s, algo_bytes=12 * f2.numel())
        return g1, g2, None


def corr2d_normalized(f1, f2, max_displacement=4):
    """Cost volume of the per-plane centred / scaled feature maps (UPFlow `if_norm_before_cost_volume`
    with moments per channel and per image), without materialising the normalised maps."""
    return _Corr2DNorm.apply(f1, f2, int(max_displacement))


class _Corr2DPair(torch.autograd.Function):
    """Both directions of one UPFlow pyramid level (upflow.py:649, 652) in one launch each way:
    (corr(f1a, f2a), corr(f1b, f2b)); with `normalize` the per-(b,c)-plane normalisation of
    normalize_features (upflow.py:96-138) is folded into the tile loads (§8f.4) and the four moment /
    adjoint passes run as one launch each."""

    @staticmethod
    def forward(ctx, f1a, f2a, f1b, f2b, md, normalize):
        ts = [_need_cuda_f32(n, t, 4) for n, t in (("f1a", f1a), ("f2a", f2a), ("f1b", f1b), ("f2b", f2b))]
        f1a, f2a, f1b, f2b = ts
        if not (f1a.shape == f2a.shape == f1b.shape == f2b.shape):
            raise ValueError("the four feature maps must share one shape, got %s" % ([tuple(t.shape) for t in ts],))
        B, C, H, W = f1a.shape
        nd = 2 * md + 1
        outa, outb = f1a.new_empty(B, nd * nd, H, W), f1a.new_empty(B, nd * nd, H, W)
        stats = None
        with torch.cuda.device(f1a.device):
            s = _stream(f1a)
            if normalize:
                if H * W < 2:
                    raise ValueError("per-plane variance needs at least two pixels")
                stats = f1a.new_empty(4, B * C, 2)
                _call("fs_plane_moments4", f1a.data_ptr(), f2a.data_ptr(), f1b.data_ptr(), f2b.data_ptr(),
                      stats.data_ptr(), B * C, H * W, s, algo_bytes=16 * f1a.numel(), record_as="fs_plane_moments")
            _call("fs_corr2d_pair_fwd", f1a.data_ptr(), f2a.data_ptr(), f1b.data_ptr(), f2b.data_ptr(), _ptr(stats),
                  outa.data_ptr(), outb.data_ptr(), B, C, H, W, md, s,
                  algo_bytes=8 * (2 * f1a.numel() + outa.numel()), algo_flops=4 * outa.numel() * C)
        ctx.save_for_backward(f1a, f2a, f1b, f2b, stats)
        ctx.md = md
        return outa, outb

    @staticmethod
    def backward(ctx, ga, gb):
        f1a, f2a, f1b, f2b, stats = ctx.saved_tensors
        need = ctx.needs_input_grad[:4]
        if not any(need):
            return (None,) * 6
        B, C, H, W = f1a.shape
        nd = 2 * ctx.md + 1
        ga = torch.zeros(B, nd * nd, H, W, device=f1a.device) if ga is None else _need_cuda_f32("grad_output", ga, 4)
        gb = torch.zeros_like(ga) if gb is None else _need_cuda_f32("grad_output", gb, 4)
        gn = [torch.empty_like(f1a) if n else None for n in need]
        with torch.cuda.device(f1a.device):
            s = _stream(f1a)
            _call("fs_corr2d_pair_bwd", f1a.data_ptr(), f2a.data_ptr(), f1b.data_ptr(), f2b.data_ptr(), _ptr(stats),
                  ga.data_ptr(), gb.data_ptr(), _ptr(gn[0]), _ptr(gn[1]), _ptr(gn[2]), _ptr(gn[3]), B, C, H, W,
                  ctx.md, s, algo_bytes=8 * (2 * f1a.numel() + ga.numel()) + 4 * sum(f1a.numel() for n in need if n),
                  algo_flops=4 * ga.numel() * C * sum(1 for n in need[:2] if n))
            if stats is None:
                return tuple(gn) + (None, None)
            gf = [torch.empty_like(f1a) if n else None for n in need]
            _call("fs_plane_norm_bwd4", f1a.data_ptr(), f2a.data_ptr(), f1b.data_ptr(), f2b.data_ptr(),
                  stats.data_ptr(), _ptr(gn[0]), _ptr(gn[1]), _ptr(gn[2]), _ptr(gn[3]), _ptr(gf[0]), _ptr(gf[1]),
                  _ptr(gf[2]), _ptr(gf[3]), B * C, H * W, s, algo_bytes=12 * f1a.numel() * sum(1 for n in need if n),
                  record_as="fs_plane_norm_bwd")
        return tuple(gf) + (None, None)


def corr2d_pair(f1a, f2a, f1b, f2b, max_displacement=4, normalize=False):
    """(corr2d(f1a, f2a), corr2d(f1b, f2b)) -- or their corr2d_normalized forms -- in one launch per pass."""
    return _Corr2DPair.apply(f1a, f2a, f1b, f2b, int(max_displacement), bool(normalize))


# --------------------------------------------------------------------------------------------
# a9/a10: robust penalty + masked reduction (UPFlow/utils/loss.py:17-48, upflow.py:267-289)
# --------------------------------------------------------------------------------------------
PEN_ABS_ROBUST, PEN_CHARBONNIER, PEN_L1_EPS, PEN_L1 = 0, 1, 2, 3
_REDUCE_BLOCKS = 1024


def _flat3(t):
    """[B,C,*spatial] -> (B, C, S) sizes of a contiguous tensor."""
    return t.shape[0], t.shape[1], int(t[0, 0].numel())


class _RobustLoss(torch.autograd.Function):
    """loss = form(S1, S2);  S1 = sum pen(x - y) * w,  S2 = sum w  (one HIP pass)."""

    @staticmethod
    def forward(ctx, x, y, w, mode, q, eps, border, form):
        x = _need_cuda_f32("x", x, x.dim())
        if y is not None:
            y = _need_cuda_f32("y", y, x.dim())
            if y.shape != x.shape:
                raise ValueError("x %s and y %s differ" % (tuple(x.shape), tuple(y.shape)))
        B, C, S = _flat3(x)
        if w is not None:
            w = _need_cuda_f32("mask", w, x.dim())
            if tuple(w.shape) != (B, 1) + tuple(x.shape[2:]):
                raise ValueError("mask must be [B,1,...] matching x %s, got %s" %
                                 (tuple(x.shape), tuple(w.shape)))
        H, W = (x.shape[2], x.shape[3]) if x.dim() == 4 else (1, S)
        sums = x.new_empty(2)
        ws = x.new_empty(2 * _REDUCE_BLOCKS)
        with torch.cuda.device(x.device):
            _call("fs_robust_sum", x.data_ptr(), _ptr(y), _ptr(w), sums.data_ptr(), ws.data_ptr(),
                  B, C, S, H, W, border, mode, float(q), float(eps), _stream(x),
                  algo_bytes=4 * x.numel() * (2 if y is not None else 1))
        S1, S2 = sums[0], sums[1]
        n = float(B * C * S)
        if form == "mean":
            loss, dS1 = S1 / n, torch.full_like(S1, 1.0 / n)  # fill kernel: graph-capturable
        elif form == "sum":
            loss, dS1 = S1 * 1.0, torch.ones_like(S1)
        elif form == "ratio":      # sum(l * w) / (sum(w) + 1e-6)          upflow.py:287
            dS1 = 1.0 / (S2 + 1e-6)
            loss = S1 * dS1
        elif form == "ratio2":     # sum(l * w) / (sum(w) * 2 + 1e-6)      loss.py:39-42
            dS1 = 1.0 / (S2 * 2 + 1e-6)
            loss = S1 * dS1
        elif form == "mean_ratio2":  # mean(l * w) / (mean(w) * 2 + 1e-6)  loss.py:20-29
            dS1 = (1.0 / n) / ((S2 / float(B * S)) * 2 + 1e-6)
            loss = S1 * dS1
        else:
            raise ValueError("unknown reduction form %r" % form)
        ctx.save_for_backward(x, y, w, dS1)
        ctx.cfg = (mode, float(q), float(eps), border)
        return loss

    @staticmethod
    def backward(ctx, gout):
        x, y, w, dS1 = ctx.saved_tensors
        need_x = ctx.needs_input_grad[0]
        need_y = y is not None and ctx.needs_input_grad[1]
        if not (need_x or need_y):
            return (None,) * 8
        mode, q, eps, border = ctx.cfg
        B, C, S = _flat3(x)
        H, W = (x.shape[2], x.shape[3]) if x.dim() == 4 else (1, S)
        coef = (gout * dS1).reshape(1).contiguous()
        gx = torch.empty_like(x) if need_x else None
        gy = torch.empty_like(x) if need_y else None
        with torch.cuda.device(x.device):
            _call("fs_robust_sum_bwd", x.data_ptr(), _ptr(y), _ptr(w), coef.data_ptr(), _ptr(gx),
                  _ptr(gy), B, C, S, H, W, border, mode, q, eps, _stream(x),
                  algo_bytes=4 * x.numel() * ((2 if y is not None else 1) + int(need_x) + int(need_y)))
        return gx, gy, None, None, None, None, None, None


def robust_loss(x, y, mask, mode, q=1.0, eps=0.0, border=0, form="mean"):
    return _RobustLoss.apply(x, y, mask, mode, q, eps, border, form)


def l1_loss(a, b):
    """torch.nn.functional.l1_loss(a, b) (mean) in one fused pass (Flow-3D/model/RIFE.py:132-134)."""
    return robust_loss(a, b, None, PEN_L1, form="mean")


def photo_loss_function(diff, mask, q, charbonnier_or_abs_robust, if_use_occ, averge=True):
    """UPFlow/utils/loss.py:17-48."""
    if charbonnier_or_abs_robust:
        if if_use_occ:
            return robust_loss(diff, None, mask, PEN_CHARBONNIER, q, 1e-6,
                               form="mean_ratio2" if averge else "ratio2")
        return robust_loss(diff, None, None, PEN_CHARBONNIER, q, 1e-8, form="mean" if averge else "sum")
    if if_use_occ:
        return robust_loss(diff, None, mask, PEN_ABS_ROBUST, q, form="ratio2")
    return robust_loss(diff, None, None, PEN_ABS_ROBUST, q, form="mean" if averge else "sum")


def photo_loss_multi_type(x, y, occ_mask, photo_loss_type='abs_robust', photo_loss_delta=0.4,
                          photo_loss_use_occ=False):
    """UPFlow/model/upflow.py:267-289: abs_robust / charbonnier / L1 through fs_robust_sum, SSIM through
    fs_wssim."""
    if photo_loss_type == 'SSIM':
        return weighted_ssim_loss(x, y, occ_mask, photo_loss_use_occ)
    mode, q, eps = {"abs_robust": (PEN_ABS_ROBUST, photo_loss_delta, 0.0),
                    "charbonnier": (PEN_CHARBONNIER, photo_loss_delta, 1e-6),
                    "L1": (PEN_L1_EPS, 1.0, 0.0)}[photo_loss_type]
    if photo_loss_use_occ:
        return robust_loss(x, y, occ_mask, mode, q, eps, form="ratio")
    return robust_loss(x, y, None, mode, q, eps, form="mean")


# --------------------------------------------------------------------------------------------
# a8: census loss (UPFlow/utils/loss.py:51-91)
# --------------------------------------------------------------------------------------------
class _CensusDist(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, max_distance):
        img1 = _need_cuda_f32("img1", img1, 4)
        img2 = _need_cuda_f32("img1_warp", img2, 4)
        if img1.shape != img2.shape or img1.shape[1] != 3:
            raise ValueError("census needs two [B,3,H,W] images, got %s and %s" %
                             (tuple(img1.shape), tuple(img2.shape)))
        B, _, H, W = img1.shape
        dist = img1.new_empty(B, 1, H, W)
        with torch.cuda.device(img1.device):
            _call("fs_census_dist_fwd", img1.data_ptr(), img2.data_ptr(), dist.data_ptr(), B, H, W,
                  int(max_distance), _stream(img1), algo_bytes=4 * (2 * img1.numel() + dist.numel()))
        ctx.save_for_backward(img1, img2)
        ctx.md = int(max_distance)
        return dist

    @staticmethod
    def backward(ctx, gdist):
        img1, img2 = ctx.saved_tensors
        n1, n2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (n1 or n2):
            return None, None, None
        gdist = gdist.contiguous()
        B, _, H, W = img1.shape
        g1 = torch.empty_like(img1) if n1 else None
        g2 = torch.empty_like(img2) if n2 else None
        with torch.cuda.device(img1.device):
            _call("fs_census_dist_bwd", img1.data_ptr(), img2.data_ptr(), gdist.data_ptr(), _ptr(g1),
                  _ptr(g2), B, H, W, ctx.md, _stream(img1),
                  algo_bytes=4 * (2 * img1.numel() + gdist.numel()) + 4 * img1.numel() * (int(n1) + int(n2)))
        return g1, g2, None


def census_dist(img1, img1_warp, max_distance=3):
    return _CensusDist.apply(img1, img1_warp, max_distance)


def census_loss(img1, img1_warp, mask, q, charbonnier_or_abs_robust, if_use_occ, averge=True,
                max_distance=3):
    """loss_functions.census_loss_torch.  The occ x inner-border mask only matters when
    if_use_occ is set (reference quirk, loss.py:42-47: otherwise it is built and ignored)."""
    dist = census_dist(img1, img1_warp, max_distance)
    if charbonnier_or_abs_robust:
        if if_use_occ:
            return robust_loss(dist, None, mask, PEN_CHARBONNIER, q, 1e-6, border=max_distance,
                               form="mean_ratio2" if averge else "ratio2")
        return robust_loss(dist, None, None, PEN_CHARBONNIER, q, 1e-8, form="mean" if averge else "sum")
    if if_use_occ:
        return robust_loss(dist, None, mask, PEN_ABS_ROBUST, q, border=max_distance, form="ratio2")
    return robust_loss(dist, None, None, PEN_ABS_ROBUST, q, form="mean" if averge else "sum")


# --------------------------------------------------------------------------------------------
# a12: IFNet epilogues
# --------------------------------------------------------------------------------------------
class _Merge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w0, w1, m):
        w0 = _need_cuda_f32("warped_img0", w0, w0.dim())
        w1 = _need_cuda_f32("warped_img1", w1, w0.dim())
        m = _need_cuda_f32("mask", m, w0.dim())
        B, C, S = _flat3(w0)
        if w1.shape != w0.shape or tuple(m.shape) != (B, 1) + tuple(w0.shape[2:]):
            raise ValueError("merge operands mismatch: %s %s %s" %
                             (tuple(w0.shape), tuple(w1.shape), tuple(m.shape)))
        merged, sig = torch.empty_like(w0), torch.empty_like(m)
        with torch.cuda.device(w0.device):
            _call("fs_merge_fwd", w0.data_ptr(), w1.data_ptr(), m.data_ptr(), merged.data_ptr(),
                  sig.data_ptr(), B, C, S, _stream(w0), algo_bytes=4 * (3 * w0.numel() + 2 * m.numel()))
        ctx.save_for_backward(w0, w1, m)
        # the sigmoid output usually has no consumer with a gradient (Flow-3D returns it, no loss reads it): an
        # undefined gradient stays None instead of a zero-filled tensor the kernel would then read
        ctx.set_materialize_grads(False)
        return merged, sig

    @staticmethod
    def backward(ctx, gmerged, gsig):
        w0, w1, m = ctx.saved_tensors
        n0, n1, nm = ctx.needs_input_grad
        if not (n0 or n1 or nm) or (gmerged is None and gsig is None):
            return None, None, None
        B, C, S = _flat3(w0)
        gmerged = torch.zeros_like(w0) if gmerged is None else gmerged.contiguous()
        gsig = gsig.contiguous() if gsig is not None else None
        g0 = torch.empty_like(w0) if n0 else None
        g1 = torch.empty_like(w1) if n1 else None
        gm = torch.empty_like(m) if nm else None
        with torch.cuda.device(w0.device):
            _call("fs_merge_bwd", w0.data_ptr(), w1.data_ptr(), m.data_ptr(), gmerged.data_ptr(),
                  _ptr(gsig), _ptr(g0), _ptr(g1), _ptr(gm), B, C, S, _stream(w0),
                  algo_bytes=4 * ((3 + int(n0) + int(n1)) * w0.numel() + (1 + int(gsig is not None) + int(nm)) * m.numel()))
        return g0, g1, gm


def merge(warped_img0, warped_img1, mask_logit):
    """(merged, sigmoid(mask)) with merged = w0 * sigmoid(m) + w1 * (1 - sigmoid(m))."""
    return _Merge.apply(warped_img0, warped_img1, mask_logit)


class _Distill(torch.autograd.Function):
    @staticmethod
    def forward(ctx, merged_i, merged_tea, gt, flow_i, flow_tea):
        d = merged_i.dim()
        ts = [_need_cuda_f32(n, t, d) for n, t in (("merged", merged_i), ("merged_teacher", merged_tea),
                                                    ("gt", gt), ("flow", flow_i),
                                                    ("flow_teacher", flow_tea))]
        merged_i, merged_tea, gt, flow_i, flow_tea = ts
        B, C, S = _flat3(merged_i)
        F_ = flow_i.shape[1]
        if not (merged_tea.shape == merged_i.shape == gt.shape and flow_tea.shape == flow_i.shape
                and flow_i.shape[2:] == merged_i.shape[2:]):
            raise ValueError("distill operands mismatch")
        sums = merged_i.new_empty(2)
        ws = merged_i.new_empty(2 * _REDUCE_BLOCKS)
        with torch.cuda.device(merged_i.device):
            _call("fs_distill_fwd", merged_i.data_ptr(), merged_tea.data_ptr(), gt.data_ptr(),
                  flow_i.data_ptr(), flow_tea.data_ptr(), sums.data_ptr(), ws.data_ptr(), B, C, F_, S,
                  _stream(merged_i), algo_bytes=4 * (3 * merged_i.numel() + 2 * flow_i.numel()))
        ctx.save_for_backward(merged_i, merged_tea, gt, flow_i, flow_tea)
        return sums[0] / float(B * S)

    @staticmethod
    def backward(ctx, gout):
        merged_i, merged_tea, gt, flow_i, flow_tea = ctx.saved_tensors
        if not ctx.needs_input_grad[3]:
            return (None,) * 5
        B, C, S = _flat3(merged_i)
        F_ = flow_i.shape[1]
        coef = (gout / float(B * S)).reshape(1).contiguous()
        gf = torch.empty_like(flow_i)
        with torch.cuda.device(merged_i.device):
            _call("fs_distill_bwd", merged_i.data_ptr(), merged_tea.data_ptr(), gt.data_ptr(),
                  flow_i.data_ptr(), flow_tea.data_ptr(), coef.data_ptr(), gf.data_ptr(), B, C, F_, S,
                  _stream(merged_i), algo_bytes=4 * (3 * merged_i.numel() + 3 * flow_i.numel()))
        return None, None, None, gf, None


class _Distill3(torch.autograd.Function):
    """Sum of the three student blocks' distillation terms in one launch each way (fs_distill3_*)."""

    @staticmethod
    def forward(ctx, m0, m1, m2, merged_tea, gt, f0, f1, f2, flow_tea):
        d = m0.dim()
        ms = [_need_cuda_f32("merged", t, d) for t in (m0, m1, m2)]
        fl = [_need_cuda_f32("flow", t, d) for t in (f0, f1, f2)]
        merged_tea, gt = _need_cuda_f32("merged_teacher", merged_tea, d), _need_cuda_f32("gt", gt, d)
        flow_tea = _need_cuda_f32("flow_teacher", flow_tea, d)
        B, C, S = _flat3(ms[0])
        F_ = fl[0].shape[1]
        if not (all(t.shape == ms[0].shape for t in ms + [merged_tea, gt]) and
                all(t.shape == fl[0].shape for t in fl + [flow_tea]) and fl[0].shape[2:] == ms[0].shape[2:]):
            raise ValueError("distill operands mismatch")
        sums = ms[0].new_empty(4)
        ws = ms[0].new_empty(4 * _REDUCE_BLOCKS)
        with torch.cuda.device(ms[0].device):
            _call("fs_distill3_fwd", ms[0].data_ptr(), ms[1].data_ptr(), ms[2].data_ptr(), merged_tea.data_ptr(),
                  gt.data_ptr(), fl[0].data_ptr(), fl[1].data_ptr(), fl[2].data_ptr(), flow_tea.data_ptr(),
                  sums.data_ptr(), ws.data_ptr(), B, C, F_, S, _stream(ms[0]),
                  algo_bytes=4 * (5 * ms[0].numel() + 4 * fl[0].numel()), record_as="fs_distill_fwd")
        ctx.save_for_backward(*ms, merged_tea, gt, *fl, flow_tea)
        n = float(B * S)
        return (0 + sums[0] / n) + sums[1] / n + sums[2] / n  # the order of IFNet.forward's running sum

    @staticmethod
    def backward(ctx, gout):
        m0, m1, m2, merged_tea, gt, f0, f1, f2, flow_tea = ctx.saved_tensors
        if not any(ctx.needs_input_grad[5:8]):
            return (None,) * 9
        B, C, S = _flat3(m0)
        F_ = f0.shape[1]
        coef = (gout / float(B * S)).reshape(1).contiguous()
        g = [torch.empty_like(f0) for _ in range(3)]
        with torch.cuda.device(m0.device):
            _call("fs_distill3_bwd", m0.data_ptr(), m1.data_ptr(), m2.data_ptr(), merged_tea.data_ptr(),
                  gt.data_ptr(), f0.data_ptr(), f1.data_ptr(), f2.data_ptr(), flow_tea.data_ptr(), coef.data_ptr(),
                  g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), B, C, F_, S, _stream(m0),
                  algo_bytes=4 * (5 * m0.numel() + 7 * f0.numel()), record_as="fs_distill_bwd")
        return (None, None, None, None, None) + tuple(g[i] if ctx.needs_input_grad[5 + i] else None
                                                      for i in range(3)) + (None,)


def distill_terms3(merged, merged_teacher, gt, flows, flow_teacher):
    """sum_i distill_term(merged[i], merged_teacher, gt, flows[i], flow_teacher) for the three student blocks
    in one launch each way; the operands of the three terms must share their shapes."""
    return _Distill3.apply(merged[0], merged[1], merged[2], merged_teacher, gt, flows[0], flows[1], flows[2],
                           flow_teacher)


def distill_term(merged_i, merged_teacher, gt, flow_i, flow_teacher):
    """One block's term of loss_distill; gradient reaches flow_i only (the reference detaches the
    teacher flow and the loss mask)."""
    return _Distill.apply(merged_i, merged_teacher, gt, flow_i, flow_teacher)


# --------------------------------------------------------------------------------------------
# §8f.1: trilinear resize of IFBlock with a gather-formulated HIP backward
# --------------------------------------------------------------------------------------------
# --------------------------------------------------------------------------------------------
# §8f.3: Flow-2D/model/laplacian.py:49-88 LapLoss as one pyramid of (input - target)
# --------------------------------------------------------------------------------------------
def _lap_sizes(N, H, W, levels):
    import ctypes
    a, b, c = ctypes.c_longlong(0), ctypes.c_longlong(0), ctypes.c_longlong(0)
    code = _lib.lib().fs_laploss2d_sizes(N, H, W, levels, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    if code != 0:
        raise ValueError("LapLoss needs every pyramid level >= 3 px per side (reflect padding by 2) and "
                         "1 <= levels <= 8; got [%d,%d,%d], %d levels" % (N, H, W, levels))
    return a.value, b.value, c.value


class _LapLoss2D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, target, levels):
        inp = _need_cuda_f32("input", inp, 4)
        target = _need_cuda_f32("target", target, 4)
        if inp.shape != target.shape:
            raise ValueError("input %s and target %s differ in shape" % (tuple(inp.shape), tuple(target.shape)))
        B, C, H, W = inp.shape
        n_sgn, n_fwd, n_bwd = _lap_sizes(B * C, H, W, levels)
        sgn = inp.new_empty(n_sgn)
        ws = inp.new_empty(n_fwd)
        loss = inp.new_empty(2)
        with torch.cuda.device(inp.device):
            _call("fs_laploss2d_fwd", inp.data_ptr(), target.data_ptr(), sgn.data_ptr(), ws.data_ptr(),
                  loss.data_ptr(), B * C, H, W, levels, _stream(inp), algo_bytes=8 * inp.numel())
        ctx.save_for_backward(sgn)
        ctx.cfg = (B, C, H, W, levels, n_bwd)
        return loss[0]

    @staticmethod
    def backward(ctx, gloss):
        (sgn,) = ctx.saved_tensors
        B, C, H, W, levels, n_bwd = ctx.cfg
        gl = gloss.detach().to(torch.float32).reshape(1).contiguous()
        ws = sgn.new_empty(n_bwd)
        gd = sgn.new_empty((B, C, H, W))
        with torch.cuda.device(sgn.device):
            _call("fs_laploss2d_bwd", sgn.data_ptr(), gl.data_ptr(), ws.data_ptr(), gd.data_ptr(), B * C, H, W,
                  levels, _stream(sgn), algo_bytes=8 * gd.numel())
        gi = gd if ctx.needs_input_grad[0] else None
        gt = -gd if ctx.needs_input_grad[1] else None
        return gi, gt, None


def laploss2d(inp, target, max_levels=5):
    """LapLoss(max_levels)(input, target) of Flow-2D/model/laplacian.py:76-88 (scalar)."""
    return _LapLoss2D.apply(inp, target, int(max_levels))


class _LapLoss3D(torch.autograd.Function):
    """3-D Laplacian-pyramid L1 loss (csrc/laplacian3d.hip): the 3-D analogue of Flow-2D's LapLoss.  PARITY
    UNPINNED -- the reference's Flow-3D/model/laplacian.py is dead code with a CPU scipy round trip and no
    gradient through the pyramid; the oracle is this build's own restatement."""

    @staticmethod
    def forward(ctx, inp, target, levels):
        inp = _need_cuda_f32("input", inp, 5)
        target = _need_cuda_f32("target", target, 5)
        if inp.shape != target.shape:
            raise ValueError("input %s and target %s differ in shape" % (tuple(inp.shape), tuple(target.shape)))
        B, C, D, H, W = inp.shape
        a, b, c = ctypes.c_longlong(0), ctypes.c_longlong(0), ctypes.c_longlong(0)
        if _lib.lib().fs_laploss3d_sizes(B * C, D, H, W, levels, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)):
            raise ValueError("LapLoss needs every pyramid level >= 3 voxels per side (reflect padding by 2) and "
                             "1 <= levels <= 8; got %s, %d levels" % (tuple(inp.shape), levels))
        sgn, ws, loss = inp.new_empty(a.value), inp.new_empty(b.value), inp.new_empty(2)
        with torch.cuda.device(inp.device):
            _call("fs_laploss3d_fwd", inp.data_ptr(), target.data_ptr(), sgn.data_ptr(), ws.data_ptr(),
                  loss.data_ptr(), B * C, D, H, W, levels, _stream(inp), algo_bytes=8 * inp.numel())
        ctx.save_for_backward(sgn)
        ctx.cfg = (tuple(inp.shape), levels, c.value)
        return loss[0]

    @staticmethod
    def backward(ctx, gloss):
        (sgn,) = ctx.saved_tensors
        shape, levels, n_bwd = ctx.cfg
        B, C, D, H, W = shape
        gl = gloss.detach().to(torch.float32).reshape(1).contiguous()
        ws, gd = sgn.new_empty(n_bwd), sgn.new_empty(shape)
        with torch.cuda.device(sgn.device):
            _call("fs_laploss3d_bwd", sgn.data_ptr(), gl.data_ptr(), ws.data_ptr(), gd.data_ptr(), B * C, D, H, W,
                  levels, _stream(sgn), algo_bytes=8 * gd.numel())
        return (gd if ctx.needs_input_grad[0] else None), (-gd if ctx.needs_input_grad[1] else None), None


def laploss3d(inp, target, max_levels=5):
    """3-D LapLoss(max_levels)(input, target) (scalar); see _LapLoss3D for its parity status."""
    return _LapLoss3D.apply(inp, target, int(max_levels))


class _Interp3D(torch.autograd.Function):
    """mul * F.interpolate(x, scale_factor = factor or 1/factor, trilinear, align_corners=False): HIP forward
    (ATen's arithmetic, bit-identical) and gather-form HIP backward."""

    @staticmethod
    def forward(ctx, x, factor, up, mul):
        x = _need_cuda_f32("x", x, 5)
        B, C, Di, Hi, Wi = x.shape
        with torch.cuda.device(x.device):
            if up:
                y = x.new_empty((B, C, Di * factor, Hi * factor, Wi * factor))
                _call("fs_upsample3d_scale_add", x.data_ptr(), 0, y.data_ptr(), B, C, Di, Hi, Wi, int(factor),
                      float(mul), _stream(x), algo_bytes=4 * (x.numel() + y.numel()))
            else:
                if min(Di, Hi, Wi) // factor < 1:
                    raise ValueError("input %s is too small to be down-sampled by %d" % (tuple(x.shape), factor))
                y = x.new_empty((B, C, Di // factor, Hi // factor, Wi // factor))
                _call("fs_downsample3d_fwd", x.data_ptr(), y.data_ptr(), B, C, Di, Hi, Wi, int(factor), float(mul),
                      _stream(x), algo_bytes=4 * (x.numel() + y.numel()))
        ctx.cfg = (tuple(x.shape), int(factor), bool(up), float(mul))
        return y

    @staticmethod
    def backward(ctx, gy):
        shape, factor, up, mul = ctx.cfg
        gy = _need_cuda_f32("grad_output", gy, 5)
        gx = gy.new_empty(shape)
        B, C, Di, Hi, Wi = shape
        Do, Ho, Wo = gy.shape[2:]
        ws = gy.new_empty(B * C * (Do * Ho * Wi + Do * Hi * Wi)) if up else None
        with torch.cuda.device(gy.device):
            if up:
                _call("fs_interp3d_bwd_scaled", gy.data_ptr(), gx.data_ptr(), ws.data_ptr(), B, C, Di, Hi, Wi, Do,
                      Ho, Wo, factor, 1, mul, _stream(gy), algo_bytes=4 * (gy.numel() + gx.numel()),
                      record_as="fs_interp3d_bwd")
            else:
                # the exact float4 form takes the scale itself; the general forms need a separate pass
                fold = (mul != 1.0 and (Di, Hi, Wi) == (Do * factor, Ho * factor, Wo * factor) and Wi % 4 == 0
                        and gx.data_ptr() % 16 == 0)
                _call("fs_interp3d_bwd_scaled", gy.data_ptr(), gx.data_ptr(), 0, B, C, Di, Hi, Wi, Do, Ho, Wo,
                      factor, 0, float(mul) if fold else 1.0, _stream(gy), algo_bytes=4 * (gy.numel() + gx.numel()),
                      record_as="fs_interp3d_bwd")
                if mul != 1.0 and not fold:
                    gx.mul_(mul)
        return gx, None, None, None


class _DownsampleCat(torch.autograd.Function):
    """F.interpolate(torch.cat(pieces, 1), scale_factor=1/factor, trilinear) without the concatenation
    (fs_downsample3d_fwd_ms reads every channel where it lies); backward: the adjoint over the concatenated
    gradient, handed out as channel slices as torch.cat's backward would."""

    @staticmethod
    def forward(ctx, factor, *pieces):
        planes = _channel_planes(pieces)
        x0 = pieces[0]
        B, (Di, Hi, Wi) = x0.shape[0], x0.shape[2:]
        if planes is None or min(Di, Hi, Wi) // factor < 1:
            raise ValueError("pieces cannot be read in place")  # (interpolate3d_cat checks first)
        pv, sv, C = planes
        y = x0.new_empty((B, C, Di // factor, Hi // factor, Wi // factor))
        with torch.cuda.device(x0.device):
            _call("fs_downsample3d_fwd_ms", pv, sv, y.data_ptr(), B, C, Di, Hi, Wi, int(factor), 1.0, _stream(x0),
                  algo_bytes=4 * (B * C * Di * Hi * Wi + y.numel()), record_as="fs_downsample3d_fwd")
        ctx.cfg = ((B, C, Di, Hi, Wi), int(factor), [int(t.shape[1]) for t in pieces])
        return y

    @staticmethod
    def backward(ctx, gy):
        shape, factor, splits = ctx.cfg
        gy = _need_cuda_f32("grad_output", gy, 5)
        gx = gy.new_empty(shape)
        B, C, Di, Hi, Wi = shape
        Do, Ho, Wo = gy.shape[2:]
        with torch.cuda.device(gy.device):
            _call("fs_interp3d_bwd_scaled", gy.data_ptr(), gx.data_ptr(), 0, B, C, Di, Hi, Wi, Do, Ho, Wo, factor, 0, 1.0,
                  _stream(gy), algo_bytes=4 * (gy.numel() + gx.numel()), record_as="fs_interp3d_bwd")
        need = ctx.needs_input_grad[1:]
        return (None,) + tuple(g if n else None for g, n in zip(gx.split(splits, 1), need))


def interpolate3d_cat(pieces, factor):
    """F.interpolate(torch.cat(pieces, 1), scale_factor=1/factor, mode="trilinear", align_corners=False) for
    factor 2 / 4 with the pieces read in place, or None when they cannot be (the caller concatenates)."""
    if factor not in (2, 4) or _channel_planes(pieces) is None or min(pieces[0].shape[2:]) // factor < 1:
        return None
    return _DownsampleCat.apply(int(factor), *pieces)


class _UpsampleScaleAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, small, prev, factor, scale):
        small = _need_cuda_f32("small", small, 5)
        B, C, Di, Hi, Wi = small.shape
        out_shape = (B, C, Di * factor, Hi * factor, Wi * factor)
        if prev is not None:
            prev = _need_cuda_f32("prev", prev, 5)
            if tuple(prev.shape) != out_shape:
                raise ValueError("prev %s must have the up-sampled shape %s" % (tuple(prev.shape), out_shape))
        out = small.new_empty(out_shape)
        with torch.cuda.device(small.device):
            _call("fs_upsample3d_scale_add", small.data_ptr(), _ptr(prev), out.data_ptr(), B, C, Di, Hi, Wi,
                  int(factor), float(scale), _stream(small),
                  algo_bytes=4 * (small.numel() + out.numel() * (2 if prev is not None else 1)))
        ctx.cfg = (tuple(small.shape), int(factor), float(scale), prev is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        shape, factor, scale, has_prev = ctx.cfg
        gout = _need_cuda_f32("grad_output", gout, 5)
        gs = None
        if ctx.needs_input_grad[0]:
            gs = gout.new_empty(shape)
            B, C, Di, Hi, Wi = shape
            Do, Ho, Wo = gout.shape[2:]
            ws = gout.new_empty(B * C * (Do * Ho * Wi + Do * Hi * Wi))
            with torch.cuda.device(gout.device):
                _call("fs_interp3d_bwd_scaled", gout.data_ptr(), gs.data_ptr(), ws.data_ptr(), B, C, Di, Hi, Wi, Do,
                      Ho, Wo, factor, 1, scale, _stream(gout), algo_bytes=4 * (gout.numel() + gs.numel()),
                      record_as="fs_interp3d_bwd")
        gp = gout if (has_prev and ctx.needs_input_grad[1]) else None
        return gs, gp, None, None


def upsample3d_scale_add(small, prev, factor, scale=1.0):
    """prev + scale * F.interpolate(small, scale_factor=factor, trilinear, align_corners=False) in one pass
    (IFBlock's flow / mask accumulation); prev may be None."""
    return _UpsampleScaleAdd.apply(small, prev, int(factor), float(scale))


def _int_factor(scale_factor):
    for f in (2, 4):
        if scale_factor == f:
            return f, True
        if scale_factor == 1.0 / f:
            return f, False
    return None, None


def interpolate3d(x, scale_factor, mul=1.0):
    """mul * F.interpolate(x, scale_factor, mode="trilinear", align_corners=False) for the IFBlock factors
    (4, 2, 1/2, 1/4) on the HIP kernels (forward and backward); other factors / dtypes: stock autograd."""
    f, up = _int_factor(scale_factor)
    if f is not None and x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and \
            (up or min(x.shape[2:]) // f >= 1):
        return _Interp3D.apply(x, f, up, float(mul))
    y = torch.nn.functional.interpolate(x, scale_factor=scale_factor, mode="trilinear",
                                        align_corners=False, recompute_scale_factor=False)
    return y if mul == 1.0 else y * mul


class _Interp2D(torch.autograd.Function):
    """mul * F.interpolate(x, scale_factor = factor or 1/factor, bilinear, align_corners=False) of the 2-D
    IFBlock (Flow-2D/model/IFNet.py:89, 92, 115-116): fs_resize2d_fwd / fs_resize2d_bwd."""

    @staticmethod
    def forward(ctx, x, factor, up, mul):
        x = _need_cuda_f32("x", x, 4)
        B, C, Hi, Wi = x.shape
        Ho, Wo = (Hi * factor, Wi * factor) if up else (Hi // factor, Wi // factor)
        if min(Ho, Wo) < 1:
            raise ValueError("input %s is too small to be down-sampled by %d" % (tuple(x.shape), factor))
        y = x.new_empty((B, C, Ho, Wo))
        with torch.cuda.device(x.device):
            _call("fs_resize2d_fwd", x.data_ptr(), y.data_ptr(), B, C, Hi, Wi, Ho, Wo, int(factor), int(up),
                  float(mul), _stream(x), algo_bytes=4 * (x.numel() + y.numel()))
        ctx.cfg = (tuple(x.shape), int(factor), bool(up), float(mul))
        return y

    @staticmethod
    def backward(ctx, gy):
        shape, factor, up, mul = ctx.cfg
        gy = _need_cuda_f32("grad_output", gy, 4)
        B, C, Hi, Wi = shape
        gx = gy.new_empty(shape)
        with torch.cuda.device(gy.device):
            _call("fs_resize2d_bwd", gy.data_ptr(), gx.data_ptr(), B, C, Hi, Wi, gy.shape[2], gy.shape[3], factor,
                  int(up), mul, _stream(gy), algo_bytes=4 * (gy.numel() + gx.numel()))
        return gx, None, None, None


def interpolate2d(x, scale_factor, mul=1.0):
    """mul * F.interpolate(x, scale_factor, mode="bilinear", align_corners=False) for the IFBlock factors on
    the HIP kernels; other factors / dtypes: stock autograd."""
    f, up = _int_factor(scale_factor)
    if f is not None and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and \
            (up or min(x.shape[2:]) // f >= 1):
        return _Interp2D.apply(x, f, up, float(mul))
    y = torch.nn.functional.interpolate(x, scale_factor=scale_factor, mode="bilinear", align_corners=False,
                                        recompute_scale_factor=False)
    return y if mul == 1.0 else y * mul


# --------------------------------------------------------------------------------------------
# PReLU after every IFNet convolution: ATen forward, one-pass HIP backward
# --------------------------------------------------------------------------------------------
_PRELU_MAX_CHUNKS = 64


def prelu_backward(x, gy, weight, want_bias_grad=False):
    """fs_prelu_bwd: (grad_x, grad_weight, grad_bias or None) of y = prelu(x, weight) in one pass over
    [B,C,*]; grad_bias = per-channel sum of grad_x (the producing convolution's bias gradient)."""
    x = _need_cuda_f32("x", x, x.dim())
    gy = _need_cuda_f32("grad_output", gy, x.dim())
    B, C, S = _flat3(x)
    gx = torch.empty_like(x)
    gw = torch.empty_like(weight)
    gb = x.new_empty(C) if want_bias_grad else None
    ws = x.new_empty((2 if want_bias_grad else 1) * B * C * _PRELU_MAX_CHUNKS)
    with torch.cuda.device(x.device):
        _call("fs_prelu_bwd", x.data_ptr(), gy.data_ptr(), weight.data_ptr(), gx.data_ptr(),
              gw.data_ptr(), _ptr(gb), ws.data_ptr(), B, C, S, weight.numel(), _stream(x),
              algo_bytes=12 * x.numel())
    return gx, gw, gb


class _PReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        y = torch.nn.functional.prelu(x, weight)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx, gw, _ = prelu_backward(x, gy, weight)
        return gx, gw


def prelu(x, weight):
    return _PReLU.apply(x, weight)


# --------------------------------------------------------------------------------------------
# 3-D correlation (new capability; no reference implementation exists)
# --------------------------------------------------------------------------------------------
class _Corr3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f1, f2, md):
        f1 = _need_cuda_f32("f1", f1, 5)
        f2 = _need_cuda_f32("f2", f2, 5)
        if f1.shape != f2.shape or f1.device != f2.device:
            raise ValueError("f1 %s and f2 %s must match" % (tuple(f1.shape), tuple(f2.shape)))
        B, C, D, H, W = f1.shape
        nd = 2 * md + 1
        out = f1.new_empty(B, nd ** 3, D, H, W)
        with torch.cuda.device(f1.device):
            _call("fs_corr3d_fwd", f1.data_ptr(), f2.data_ptr(), out.data_ptr(), B, C, D, H, W, md,
                  _stream(f1), algo_bytes=4 * (2 * f1.numel() + out.numel()))
        ctx.save_for_backward(f1, f2)
        ctx.md = md
        return out

    @staticmethod
    def backward(ctx, gout):
        f1, f2 = ctx.saved_tensors
        g1 = torch.empty_like(f1) if ctx.needs_input_grad[0] else None
        g2 = torch.empty_like(f2) if ctx.needs_input_grad[1] else None
        if g1 is None and g2 is None:
            return None, None, None
        gout = _need_cuda_f32("grad_output", gout, 5)
        B, C, D, H, W = f1.shape
        with torch.cuda.device(f1.device):
            _call("fs_corr3d_bwd", f1.data_ptr(), f2.data_ptr(), gout.data_ptr(), _ptr(g1), _ptr(g2), B, C,
                  D, H, W, ctx.md, _stream(f1))
        return g1, g2, None


def corr3d(f1, f2, max_displacement=4):
    """Volume cost volume [B,(2md+1)^3,D,H,W]: channel-mean of f1 * shifted f2, zero padded, dz-major."""
    return _Corr3D.apply(f1, f2, int(max_displacement))


# --------------------------------------------------------------------------------------------
# IFNet-3D convolution weight gradient: implicit GEMM on the fp32 matrix cores
# --------------------------------------------------------------------------------------------
# --------------------------------------------------------------------------------------------
# Prepared convolution weights: one re-layout launch per optimiser step instead of one per convolution
# --------------------------------------------------------------------------------------------
# fs_conv3d_fwd* / fs_conv3d_tr* re-lay their weights into a slab (`ws`) with a ~5 us launch in front of every
# convolution: ~110 launches + dispatch gaps per 256^3 Flow-3D step.  Inside a `with prepared_weights():` block the
# slabs of weights that are autograd LEAVES (a model's parameters) are kept, keyed by (storage address, layer
# geometry), and the first convolution that meets a stale slab re-lays ALL registered weights of the device with ONE
# fs_conv3d_wprep_batch launch.  The block is a promise by its owner -- `Model.update` / `Model.inference` -- that the
# weights change only at its end (the optimiser step): leaving it makes every slab stale.  That explicit epoch is the
# rule; tensor version counters are checked as well, but they cannot be the rule: torch's fused AdamW updates
# parameters without bumping them.  Outside such a block (a bare IFNet, the convgrad modules in someone else's
# model) every convolution prepares its weights itself, as before.  FLOWSCI_WPREP_PER_LAUNCH=1 (ablation mode): never keep slabs.
import contextlib as _contextlib
import os as _os
import weakref as _weakref

_PREP_ON = _lib.ablation_env("FLOWSCI_WPREP_PER_LAUNCH") != "1"
_prep_epoch = 0
_prep_depth = 0
_prep_tables = {}


def invalidate_prepared_weights():
    """Every cached weight slab is stale from now on."""
    global _prep_epoch
    _prep_epoch += 1


@_contextlib.contextmanager
def prepared_weights():
    """Weights are constant inside this block except for an optimiser step at its very end."""
    global _prep_depth
    _prep_depth += 1
    try:
        yield
    finally:
        _prep_depth -= 1
        invalidate_prepared_weights()
        if _prep_depth == 0 and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            for tab in _prep_tables.values():  # layers met for the first time in this block join the batch table,
                tab._evict()                   # slabs no block has used for a while leave it
                if tab.dirty:
                    tab.upload()


class _PrepEntry:
    __slots__ = ("wref", "ws", "jobs", "stamp", "used", "pinned")


_PREP_KEEP_EPOCHS = 8      # a slab not used for this many weight epochs (optimiser steps / inference calls) is dropped
_PREP_MAX_JOBS = 65535     # fs_conv3d_wprep_batch's grid limit


class _PrepTable:
    def __init__(self, device):
        self.device, self.entries, self.dirty, self.dev_jobs, self.njobs = device, {}, False, None, 0
        # what a HIP-graph capture has baked in as raw pointers -- job tables (the captured fs_conv3d_wprep_batch
        # launch reads `dev_jobs.data_ptr()`) and slabs (the captured convolutions read `ws`) -- must never go back
        # to the caching allocator while the graph may be replayed; a graph has no destructor hook here, so they are
        # kept for the life of the process (a table is 48 B per job, a slab 1-2 MB per layer and shape)
        self.captured = []

    def _live(self):
        live = []
        for k, e in list(self.entries.items()):
            w = e.wref()
            if w is None:  # the weight is gone (its address may be reused): forget the slab
                del self.entries[k]
                self.dirty = True
            elif e.jobs:
                live.append((e, w))
        return live

    def _evict(self):
        """Variable-size inference / evaluation meets a new geometry per call: slabs that no block has used for
        _PREP_KEEP_EPOCHS epochs leave the table (and the batch launch) unless a captured graph reads them."""
        for k, e in list(self.entries.items()):
            if not e.pinned and _prep_epoch - e.used > _PREP_KEEP_EPOCHS:
                del self.entries[k]
                self.dirty = True

    def upload(self):
        """(Re)build the device copy of the job table -- a pinned host buffer and an H2D copy: not capturable, so
        prepared_weights() does it when a block ends, never inside a HIP-graph capture.  The previous table is
        simply dropped unless a capture referenced it (`captured`)."""
        self._evict()
        live = self._live()
        jobs = [j for e, _ in live for j in e.jobs]
        if len(jobs) > _PREP_MAX_JOBS:  # more than one launch can take: every convolution prepares its own weights
            jobs = []
        if jobs:
            arr = (_lib.FsWprepJob * len(jobs))(*jobs)
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).pin_memory()
            self.dev_jobs, self.njobs = host.to(self.device, non_blocking=True), len(jobs)
        else:
            self.dev_jobs, self.njobs = None, 0
        self.dirty = False

    def refresh(self):
        """Re-lay every registered weight with one launch.  False when that is not possible right now (the table
        changed and the stream is capturing): the caller prepares its own weights, as the classic path does."""
        if self.dirty or self.dev_jobs is None:
            if torch.cuda.is_current_stream_capturing():
                return False
            self.upload()
        live = self._live()
        if self.dirty:  # an entry died between the upload and now
            return False
        if not self.njobs:
            return False  # nothing registered, or too many jobs for one launch: per-launch preparation
        if torch.cuda.is_current_stream_capturing():
            self.captured.append(self.dev_jobs)       # the graph replays this launch with this table's address
            for e, _ in live:
                e.pinned = True
        with torch.cuda.device(self.device):
            _call("fs_conv3d_wprep_batch", self.dev_jobs.data_ptr(), self.njobs,
                  torch.cuda.current_stream(self.device).cuda_stream,
                  algo_bytes=8 * sum(j.total for e, _ in live for j in e.jobs))
        for e, w in live:
            e.stamp = (w._version, _prep_epoch)
        return True


def _prepared(w, nfloats, key, plan):
    """(pointer to pass as `w`, slab tensor to pass as `ws`).  plan(jobs, cap, ws) -> number of FsWprepJob records
    written (the library's own dispatch decides the layout).  0 as the pointer means "the slab is prepared"."""
    if not (_PREP_ON and _prep_depth > 0 and w.is_leaf and w.requires_grad):
        return w.data_ptr(), w.new_empty(max(int(nfloats), 1))
    tab = _prep_tables.get((w.device.type, w.device.index))
    if tab is None:
        tab = _prep_tables[(w.device.type, w.device.index)] = _PrepTable(w.device)
    global _last_prep
    k = (w.data_ptr(), tuple(w.shape)) + key
    e = tab.entries.get(k)
    capturing = torch.cuda.is_current_stream_capturing()
    if e is not None and e.wref() is not None:
        e.used = _prep_epoch
        if capturing and not e.pinned:  # a captured convolution reads this slab by address from now on
            e.pinned = True
            tab.captured.append(e.ws)
    _last_prep = None  # (only a first-use stamp can describe a slab nobody has written: _prep_not_written)
    if e is None or e.wref() is None:
        e = _last_prep = _PrepEntry()
        e.wref, e.stamp, e.used, e.pinned = _weakref.ref(w), None, _prep_epoch, capturing
        e.ws = torch.empty(max(int(nfloats), 1), device=w.device, dtype=torch.float32)
        buf = (_lib.FsWprepJob * 4)()
        n = int(plan(buf, 4, e.ws))
        if n < 0 or n > 4:
            raise _lib.FlowsciKernelError("weight re-layout plan failed (%d)" % n)
        e.jobs = [_lib.FsWprepJob.from_buffer_copy(buf[i]) for i in range(n)]
        tab.entries[k] = e
        tab.dirty = True
        if capturing:
            tab.captured.append(e.ws)
        # first use: the convolution re-lays the weights into the (now persistent) slab itself, as the classic path does;
        # from the next stale epoch on the slab is part of the device's one batch launch
        e.stamp = (w._version, _prep_epoch)
        return w.data_ptr(), e.ws
    if not e.jobs:  # this shape's kernel reads the weights as stored
        return w.data_ptr(), e.ws
    if e.stamp != (w._version, _prep_epoch) and not tab.refresh():
        e.stamp = (w._version, _prep_epoch)
        return w.data_ptr(), e.ws
    return 0, e.ws


_last_prep = None


def _prep_not_written():
    """The entry point that was handed a FIRST-USE slab (stamped current on the promise that the call itself lays the
    weights out) answered FS_ERR_UNSUPPORTED: it may have returned before its re-layout ran (csrc/convfwd.hip rejects
    > 12 source planes first), so the stamp does not describe the slab -- the fallback call with the same key must not
    pass w = NULL.  Slabs written by an earlier call or by the batch launch are not touched."""
    if _last_prep is not None:
        _last_prep.stamp = None


_wino_seen = {}


def _fwd_k3_macs(xptr, B, Cin, Cout, dhw, wmode):
    """Multiply-adds per output and input channel that fs_conv3d_fwd* EXECUTES for this k3 s1 p1 call: 27 as a direct
    implicit GEMM, 18 as the 1-D Winograd F(2,3) kernel (csrc/convwino.hpp), 13.5 as F(4,3) (csrc/convwino4.hpp), 9 as the
    2-D F(2,3) x F(4,3) kernel (csrc/convwino2d.hpp).
    Asked of the library's own dispatch (its re-layout plan), cached per geometry; only the flop accounting of the
    timing records depends on it."""
    key = (xptr % 16, B, Cin, Cout) + tuple(int(v) for v in dhw) + (int(wmode),)
    if key not in _wino_seen:
        buf = (_lib.FsWprepJob * 4)()
        n = _lib.lib().fs_conv3d_fwd_wprep_jobs(buf, 4, 0x1000 + xptr % 16, 0x1000, 0x1000, B, Cin, Cout, *key[4:7],
                                                *key[4:7], 3, 1, 1, int(wmode))
        _wino_seen[key] = {4: 18.0, 5: 13.5, 6: 9.0}.get(buf[0].kind, 27.0) if n == 1 else 27.0
    return _wino_seen[key]


def _fwd_k4_symbol(xptr, B, Cin, Cout, in_dhw, out_dhw):
    """Kernel symbol of a k4 s2 p1 fs_conv3d_fwd* call where ops.py can name it: the round-5 kernel that runs the layer
    with fp32 accuracy on the bf16 matrix rate (csrc/convfwd_s3.hpp; the library's re-layout plan says slab kind 7).  Asked
    only while launches are being timed; None = the fp32-MFMA kernels (several instantiations)."""
    if _timing is None:
        return None
    key = ("k4", xptr % 16, B, Cin, Cout) + tuple(int(v) for v in in_dhw)
    if key not in _wino_seen:
        buf = (_lib.FsWprepJob * 4)()
        n = _lib.lib().fs_conv3d_fwd_wprep_jobs(buf, 4, 0x1000 + xptr % 16, 0x1000, 0x1000, B, Cin, Cout, *key[5:8],
                                                *[int(v) for v in out_dhw], 4, 2, 1, 0)
        _wino_seen[key] = (n == 1 and buf[0].kind == 7)
    return ("conv3d_fwd_s3_kernel<%d, 8, 4>" % (1 if Cout <= 32 else 2)) if _wino_seen[key] else None


def _tr_symbol(xptr, B, Cin, Cout, in_dhw, out_dhw, has_z):
    """Kernel symbol of an fs_conv3d_tr* call where ops.py can name it: the round-5 split-bf16 kernels (csrc/convtr_s3.hpp;
    the library's re-layout plan says slab kind 8 = the 32-row form, 9 = the 16-row form).  Asked only while launches are
    being timed; None = the fp32-MFMA kernels."""
    if _timing is None:
        return None
    key = ("tr", xptr % 16, B, Cin, Cout, int(has_z)) + tuple(int(v) for v in in_dhw) + tuple(int(v) for v in out_dhw)
    if key not in _wino_seen:
        buf = (_lib.FsWprepJob * 8)()
        n = _lib.lib().fs_conv3d_tr_wprep_jobs(buf, 8, 0x1000 + xptr % 16, 0x1000, 0x1000, B, Cin, Cout, *key[6:12], int(has_z))
        kinds = set(buf[i].kind for i in range(max(n, 0)))
        _wino_seen[key] = "convtr_s3_kernel<false>" if kinds == {8} else ("convtr_s3_kernel<true>" if kinds == {9} else None)
    return _wino_seen[key]


def _fwd_k3_symbol(macs, W):
    """Kernel symbol of a k3 s1 p1 fs_conv3d_fwd* call whose dispatch executes `macs` multiply-adds per (output, input
    channel) -- the names `rocprofv3 --kernel-trace --stats` prints (csrc/convwino2d.hpp::launch_wino2d: 16 x-tiles per
    row on rows of 64 voxels, 8 on rows of 32); None for the direct kernels (several instantiations by shape)."""
    return "conv3d_wino2d_ps_kernel<0, %d>" % (16 if int(W) % 64 == 0 else 8) if macs == 9.0 else None


def _prepared_fwd(w, xptr, B, Cin, Cout, in_dhw, out_dhw, k, stride, pad, wmode):
    """Slab of fs_conv3d_fwd* for this call: which layout (direct taps, or the Winograd-transformed filter of the
    64-channel k3 trunk layers) is the library's decision for the call's geometry."""
    L = _lib.lib()
    geo = (B, Cin, Cout) + tuple(int(v) for v in in_dhw) + tuple(int(v) for v in out_dhw) + (int(k), int(stride), int(pad),
                                                                                          int(wmode))
    return _prepared(w, L.fs_conv3d_fwd_ws_floats(Cin, Cout, int(k)), ("fwd", xptr % 16) + geo,
                     lambda jobs, cap, ws: L.fs_conv3d_fwd_wprep_jobs(jobs, cap, xptr, w.data_ptr(), ws.data_ptr(), *geo))


def conv3d_wrw_supported(k, stride, padding):
    return (len(k) == 3 and k[0] == k[1] == k[2] and stride[0] == stride[1] == stride[2] and
            padding[0] == padding[1] == padding[2] and (k[0], stride[0]) in ((3, 1), (4, 2)) and
            0 <= padding[0] < k[0])


def _vol(dhw):
    return int(dhw[0]) * int(dhw[1]) * int(dhw[2])


# The convolution kernels address one staged channel chunk with 32-bit byte offsets; these predicates
# mirror the FS_ERR_SHAPE limits of csrc/conv{fwd,tr,wrw}.hip so that callers can route an oversize layer
# (e.g. the first / last layers of a 512^3 volume) to the torch.nn base class instead of raising.
def conv3d_fwd_fits(in_dhw, out_dhw, k):
    return (4 if k == 3 else 2) * _vol(in_dhw) * 4 < (1 << 32) and _vol(out_dhw) < (1 << 31)


def conv3d_tr_fits(in_dhw):
    return 4 * _vol(in_dhw) * 4 < (1 << 32) and 8 * _vol(in_dhw) < (1 << 31)


def conv3d_wrw_fits(src_dhw, g_dhw):
    return 8 * _vol(src_dhw) * 4 < (1 << 32) and 64 * _vol(g_dhw) * 4 < (1 << 32)


# The weight-gradient kernels finish with float atomics into a zero-filled dW: ~56 fill launches of a few KB .. 1.7 MB per
# Flow-3D step (4 us each plus their dispatch gaps).  Inside a prepared_weights() block -- one optimiser step -- the dW
# tensors are slices of ONE arena zeroed by a single memset; the arena is sized from what the previous step asked for and
# lives as long as any gradient that points into it (a new one per step: nothing is ever re-zeroed in place).
_dw_arena = {}   # device -> [epoch, tensor, offset (floats), floats asked for so far in this epoch]
_dw_need = {}    # device -> floats the last complete epoch asked for


def _dw_zeros(like, shape):
    n = 1
    for v in shape:
        n *= int(v)
    if _prep_depth == 0 or not _PREP_ON or torch.cuda.is_current_stream_capturing():
        return like.new_zeros(shape)
    dev = like.device
    st = _dw_arena.get(dev)
    if st is None or st[0] != _prep_epoch:
        if st is not None:
            _dw_need[dev] = st[3]
        need = _dw_need.get(dev, 0)
        st = _dw_arena[dev] = [_prep_epoch, like.new_zeros(need) if need else None, 0, 0]
    pad = (n + 3) // 4 * 4  # slices stay 16-byte aligned
    st[3] += pad
    if st[1] is None or st[2] + pad > st[1].numel():
        return like.new_zeros(shape)  # first step, or a step that asks for more than the last one did
    out = st[1].narrow(0, st[2], n).view(shape)
    st[2] += pad
    return out


def _conv3d_wrw_det(g, src_ptr, pv, sv, geo, nbytes, flops, equiv, allow_unsupported=False):
    """fs_conv3d_wrw_det: the same kernels with every run of positions storing its partial tile into its own copy of dW
    (a workspace of runs x |dW| floats) and one more launch adding the copies in run order -- no float atomics, bitwise
    reproducible.  Taken when torch.are_deterministic_algorithms_enabled()."""
    L = _lib.lib()
    need = int(L.fs_conv3d_wrw_det_ws_floats(g.data_ptr(), src_ptr, pv, sv, *geo))
    if need == -FS_ERR_UNSUPPORTED and allow_unsupported:
        return None
    if need < 0:
        _lib.check(-need, "fs_conv3d_wrw_det_ws_floats")
    B, Cg, Cs = geo[:3]
    k = geo[9]
    dw = g.new_empty(Cg, Cs, k, k, k)
    ws = g.new_empty(max(need, 1))
    with torch.cuda.device(g.device):
        _call("fs_conv3d_wrw_det", g.data_ptr(), src_ptr, pv, sv, dw.data_ptr(), ws.data_ptr(), need, *geo, _stream(g),
              algo_bytes=nbytes, algo_flops=flops, equiv_flops=equiv, record_as="fs_conv3d_wrw")
    return dw


def conv3d_wrw(g, src, k, stride, pad):
    """dW[Cg, Cs, k,k,k] = sum_{b,o} g[b,:,o] (x) src[b,:,o*stride + koff - pad]  (fs_conv3d_wrw)."""
    g = _need_cuda_f32("g", g, 5)
    src = _need_cuda_f32("src", src, 5)
    B, Cg = g.shape[:2]
    Cs = src.shape[1]
    if src.shape[0] != B:
        raise ValueError("batch mismatch")
    fq = 2 * g.numel() * Cs * int(k) ** 3
    kid = conv3d_wrw_kernel_id(g.data_ptr(), src.data_ptr(), B, Cg, Cs, g.shape[2:], src.shape[2:], k, stride, pad)
    geo = (B, Cg, Cs, g.shape[2], g.shape[3], g.shape[4], src.shape[2], src.shape[3], src.shape[4], int(k), int(stride), int(pad))
    if torch.are_deterministic_algorithms_enabled():
        return _conv3d_wrw_det(g, src.data_ptr(), None, None, geo, 4 * (g.numel() + src.numel()),
                               {WRW_KERNEL_WINO43: fq // 2, WRW_KERNEL_WINO23: fq * 2 // 3}.get(kid, fq), fq)
    dw = _dw_zeros(g, (Cg, Cs, k, k, k))
    with torch.cuda.device(g.device):
        _call("fs_conv3d_wrw", g.data_ptr(), src.data_ptr(), dw.data_ptr(), B, Cg, Cs, g.shape[2],
              g.shape[3], g.shape[4], src.shape[2], src.shape[3], src.shape[4], int(k), int(stride),
              int(pad), _stream(g), algo_bytes=4 * (g.numel() + src.numel()),
              algo_flops={WRW_KERNEL_WINO43: fq // 2, WRW_KERNEL_WINO23: fq * 2 // 3}.get(kid, fq), equiv_flops=fq,
              kernel="conv3d_wrw_wino4_kernel<0>" if kid == WRW_KERNEL_WINO43 else None)
    return dw


WRW_KERNEL_BRICK, WRW_KERNEL_DMA, WRW_KERNEL_WINO23, WRW_KERNEL_WINO43 = 0, 1, 2, 3  # include/flowsci_hip.h FS_WRW_KERNEL_*


def conv3d_wrw_kernel_id(g_ptr, src_ptr, B, Cg, Cs, g_dhw, src_dhw, k, stride, pad):
    """The kernel fs_conv3d_wrw dispatches this call to (WRW_KERNEL_*): the library's own answer
    (fs_conv3d_wrw_kernel_id; nothing is launched, the pointers are inspected for alignment only)."""
    kid = int(_lib.lib().fs_conv3d_wrw_kernel_id(int(g_ptr), int(src_ptr), int(B), int(Cg), int(Cs),
                                                 *(int(v) for v in g_dhw), *(int(v) for v in src_dhw), int(k), int(stride),
                                                 int(pad)))
    if kid < 0:
        _lib.check(-kid, "fs_conv3d_wrw_kernel_id")
    return kid


def conv3d_wrw_takes_winograd(B, Cg, Cs, g_dhw, src_dhw, k, stride, pad, g_misalign=0, src_misalign=0):
    """Does fs_conv3d_wrw run this call in a Winograd domain (F(4,3), csrc/convwrwwino4.hpp: half the direct form's
    multiply-adds)?  Asked of the library's dispatch: the 64 -> 64 k3 s1 p1 layers with rows of 64 x, an even number
    of y rows, >= 1024 position bricks, 16-byte aligned operands."""
    return conv3d_wrw_kernel_id(0x1000 + g_misalign, 0x1000 + src_misalign, B, Cg, Cs, g_dhw, src_dhw, k, stride,
                                pad) in (WRW_KERNEL_WINO23, WRW_KERNEL_WINO43)


FS_ERR_UNSUPPORTED = 5


def _channel_planes(pieces):
    """(pointer array, batch-stride array, Cin) for fs_*_ms -- two ctypes HOST arrays of Cin entries, read by the entry
    point at launch (the caller keeps `pieces` alive across the launch) -- every channel of every piece as the
    plane it already is.  A piece is a [B, Ci, D,H,W] tensor, contiguous or a channel slice of a wider contiguous
    one; None when a piece has other strides / alignment (the caller concatenates instead)."""
    B, dhw = pieces[0].shape[0], tuple(pieces[0].shape[2:])
    vol = dhw[0] * dhw[1] * dhw[2]
    inner = (vol, dhw[1] * dhw[2], dhw[2], 1)
    ptrs, strides = [], []
    for t in pieces:
        if (not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32 or t.dim() != 5 or
                t.shape[0] != B or tuple(t.shape[2:]) != dhw or tuple(t.stride()[1:]) != inner or
                t.stride(0) < t.shape[1] * vol or t.stride(0) % 4 or t.data_ptr() % 16 or vol % 4):
            return None
        for c in range(t.shape[1]):
            ptrs.append(t.data_ptr() + 4 * c * vol)
            strides.append(t.stride(0))
    n = len(ptrs)
    if n > 12:
        return None
    return (ctypes.c_void_p * n)(*ptrs), (ctypes.c_longlong * n)(*strides), n


def conv3d_fwd_prelu_ms(pieces, w, bias, prelu_weight, k, stride, pad):
    """(y, prelu(y)) of conv3d(torch.cat(pieces, 1), w, bias, stride, pad) without the concatenation
    (fs_conv3d_fwd_prelu_ms: the loader waves read every channel where it lies), or None when no such kernel
    covers the shape -- the caller then concatenates."""
    planes = _channel_planes(pieces)
    if planes is None:
        return None
    pv, sv, Cin = planes
    w = _need_cuda_f32("w", w, 5)
    a = _need_cuda_f32("prelu_weight", prelu_weight, 1)
    Cout = w.shape[0]
    if w.shape[1] != Cin or tuple(w.shape[2:]) != (k, k, k):
        raise ValueError("weight %s does not fit %d input channels" % (tuple(w.shape), Cin))
    if bias is not None:
        bias = _need_cuda_f32("bias", bias, 1)
    x0 = pieces[0]
    B = x0.shape[0]
    Di, Hi, Wi = x0.shape[2:]
    Do, Ho, Wo = [(n + 2 * pad - k) // stride + 1 for n in (Di, Hi, Wi)]
    if min(Do, Ho, Wo) < 1:
        return None
    y = x0.new_empty((B, Cout, Do, Ho, Wo))
    z = torch.empty_like(y)
    wp, ws = _prepared_fwd(w, x0.data_ptr(), B, Cin, Cout, (Di, Hi, Wi), (Do, Ho, Wo), k, stride, pad, 0)
    xbytes = 4 * B * Cin * Di * Hi * Wi
    with torch.cuda.device(x0.device):
        rc = _call_rc("fs_conv3d_fwd_prelu_ms", pv, sv, wp, _ptr(bias), a.data_ptr(), y.data_ptr(), z.data_ptr(),
                   ws.data_ptr(), B, Cin, Cout, Di, Hi, Wi, Do, Ho, Wo, int(k), int(stride), int(pad), int(a.numel()),
                      _stream(x0), algo_bytes=xbytes + 8 * y.numel(), algo_flops=2 * y.numel() * Cin * int(k) ** 3,
                      record_as="fs_conv3d_fwd", allow=(FS_ERR_UNSUPPORTED,))
    if rc == FS_ERR_UNSUPPORTED:
        _prep_not_written()
        return None
    return y, z


def conv3d_wrw_ms(g, pieces, k, stride, pad):
    """conv3d_wrw(g, torch.cat(pieces, 1), ...) without the concatenation (fs_conv3d_wrw_ms), or None."""
    planes = _channel_planes(pieces)
    if planes is None:
        return None
    pv, sv, Cs = planes
    g = _need_cuda_f32("g", g, 5)
    B, Cg = g.shape[:2]
    Di, Hi, Wi = pieces[0].shape[2:]
    if torch.are_deterministic_algorithms_enabled():
        geo = (B, Cg, Cs, g.shape[2], g.shape[3], g.shape[4], Di, Hi, Wi, int(k), int(stride), int(pad))
        fq = 2 * g.numel() * Cs * int(k) ** 3
        return _conv3d_wrw_det(g, 0, pv, sv, geo, 4 * (g.numel() + B * Cs * Di * Hi * Wi), fq, fq, allow_unsupported=True)
    dw = _dw_zeros(g, (Cg, Cs, k, k, k))
    with torch.cuda.device(g.device):
        rc = _call_rc("fs_conv3d_wrw_ms", g.data_ptr(), pv, sv, dw.data_ptr(), B, Cg, Cs, g.shape[2], g.shape[3], g.shape[4],
                   Di, Hi, Wi, int(k), int(stride), int(pad), _stream(g),
                      algo_bytes=4 * (g.numel() + B * Cs * Di * Hi * Wi), algo_flops=2 * g.numel() * Cs * int(k) ** 3,
                      record_as="fs_conv3d_wrw", allow=(FS_ERR_UNSUPPORTED,))
    if rc == FS_ERR_UNSUPPORTED:
        return None
    return dw


def conv3d_deconv_grad_input_dprelu(gy, w, act_y, prelu_weight):
    """Input gradient of ConvTranspose3d(4, 2, 1) with weight w [Cin, Cout, 4,4,4] whose INPUT was z = prelu(act_y):
    (grad_act_y, grad_prelu_weight, grad_bias_of_the_layer_that_produced_act_y) in one launch
    (fs_conv3d_fwd_dprelu: the PReLU backward is the convolution's epilogue), or None when that fused kernel does not
    cover the shape -- the caller then runs conv3d_fwd + prelu_backward."""
    gy = _need_cuda_f32("grad_output", gy, 5)
    w = _need_cuda_f32("w", w, 5)
    act_y = _need_cuda_f32("act_y", act_y, 5)
    a = _need_cuda_f32("prelu_weight", prelu_weight, 1)
    B, Cg = gy.shape[:2]                      # the strided convolution's input = grad_out: Cg = deconv Cout
    Cin_t = w.shape[0]                        # its output channels = the deconvolution's input channels
    if w.shape[1] != Cg or tuple(w.shape[2:]) != (4, 4, 4) or act_y.shape[1] != Cin_t or a.numel() not in (1, Cin_t):
        raise ValueError("shapes do not describe a k4 s2 deconvolution: gy %s w %s act_y %s" %
                         (tuple(gy.shape), tuple(w.shape), tuple(act_y.shape)))
    Di, Hi, Wi = gy.shape[2:]
    Do, Ho, Wo = act_y.shape[2:]
    if any((n + 2 - 4) // 2 + 1 != m for n, m in zip((Di, Hi, Wi), (Do, Ho, Wo))) or Cin_t > 32:
        return None
    L = _lib.lib()
    npart = int(L.fs_conv3d_fwd_dprelu_part_floats(B, Cin_t, Do, Ho, Wo))
    if npart < 0:
        return None
    out = torch.empty_like(act_y)
    ga, gb = torch.empty_like(a), act_y.new_empty(Cin_t)
    part = act_y.new_empty(npart)
    wp, ws = _prepared_fwd(w, gy.data_ptr(), B, Cg, Cin_t, (Di, Hi, Wi), (Do, Ho, Wo), 4, 2, 1, 0)
    nb = 4 * (gy.numel() + 2 * out.numel())
    fl = 2 * out.numel() * Cg * 64
    with torch.cuda.device(gy.device):
        args = (gy.data_ptr(), wp, act_y.data_ptr(), a.data_ptr(), a.numel(), out.data_ptr(), ga.data_ptr(),
                gb.data_ptr(), part.data_ptr(), ws.data_ptr(), B, Cg, Cin_t, Di, Hi, Wi, Do, Ho, Wo, 4, 2, 1, _stream(gy))
        if _timing is None or (_timing_only is not None and "fs_conv3d_fwd" not in _timing_only):
            rc = L.fs_conv3d_fwd_dprelu(*args)
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = L.fs_conv3d_fwd_dprelu(*args)
            e1.record()
            if rc == 0:
                _timing.setdefault("fs_conv3d_fwd", []).append((e0, e1, nb, fl, fl, None))
    if rc == FS_ERR_UNSUPPORTED:
        _prep_not_written()
        return None
    _lib.check(rc, "fs_conv3d_fwd_dprelu")
    return out, ga, gb


def conv3d_k3_grad_input_dprelu(gy, w, act_y, prelu_weight):
    """Input gradient of a stride-1 'same' Conv3d (weight w [Cout, Cin, 3,3,3]) whose INPUT was z = prelu(act_y) -- the
    inner PReLU of an IFBlock residual unit: (grad_act_y, grad_prelu_weight, grad_bias of the layer that produced act_y)
    in one launch (fs_conv3d_fwd_dprelu, kernel 3: the PReLU backward is the convolution's epilogue), or None when the
    fused kernel does not cover the shape (the caller then runs conv3d_fwd wmode 1 + prelu_backward)."""
    gy = _need_cuda_f32("grad_output", gy, 5)
    w = _need_cuda_f32("w", w, 5)
    act_y = _need_cuda_f32("act_y", act_y, 5)
    a = _need_cuda_f32("prelu_weight", prelu_weight, 1)
    B, Cg = gy.shape[:2]
    Cx = w.shape[1]                           # the gradient's channels = the convolution's input channels
    if w.shape[0] != Cg or tuple(w.shape[2:]) != (3, 3, 3) or tuple(act_y.shape) != (B, Cx) + tuple(gy.shape[2:]):
        raise ValueError("shapes do not belong to one convolution: %s %s %s" % (tuple(gy.shape), tuple(w.shape), tuple(act_y.shape)))
    if a.numel() not in (1, Cx):
        raise ValueError("prelu_weight must have 1 or %d elements" % Cx)
    D, H, W = gy.shape[2:]
    L = _lib.lib()
    npart = int(L.fs_conv3d_fwd_dprelu_part_floats_k3(B, Cx, D, H, W))
    if npart < 0:
        return None
    out = torch.empty_like(act_y)
    ga, gb = torch.empty_like(a), act_y.new_empty(Cx)
    part = act_y.new_empty(npart)
    wp, ws = _prepared_fwd(w, gy.data_ptr(), B, Cg, Cx, (D, H, W), (D, H, W), 3, 1, 1, 1)
    macs = _fwd_k3_macs(gy.data_ptr(), B, Cg, Cx, (D, H, W), 1)
    with torch.cuda.device(gy.device):
        rc = _call_rc("fs_conv3d_fwd_dprelu", gy.data_ptr(), wp, act_y.data_ptr(), a.data_ptr(), a.numel(),
                      out.data_ptr(), ga.data_ptr(), gb.data_ptr(), part.data_ptr(), ws.data_ptr(), B, Cg, Cx, D, H, W,
                      D, H, W, 3, 1, 1, _stream(gy), algo_bytes=4 * (gy.numel() + 2 * out.numel()),
                      algo_flops=int(2 * out.numel() * Cg * macs), kernel=_fwd_k3_symbol(macs, W),
                      equiv_flops=2 * out.numel() * Cg * 27, record_as="fs_conv3d_fwd", allow=(FS_ERR_UNSUPPORTED,))
    if rc == FS_ERR_UNSUPPORTED:
        _prep_not_written()
        return None
    return out, ga, gb


def conv3d_fwd_workgroups(B, Cout, out_dhw, k):
    """Workgroups fs_conv3d_fwd launches for this layer (mirrors its brick choice, csrc/convfwd.hip)."""
    Do, Ho, Wo = out_dhw
    cd = lambda a, b: -(-a // b)
    tw = 32 if Wo > 16 else 16
    r = 32 // tw
    if k == 3:
        mg = cd(Cout, 64)
        big = B * cd(Do, 2) * cd(Ho, 8 * r) * cd(Wo, tw) * mg
        if big >= 512:
            return big
        small = B * Do * cd(Ho, 4 * r) * cd(Wo, tw) * mg
        return small if small >= 256 else 2 * small  # 32-channel workgroups when 64-channel ones cannot fill the chip
    if Cout <= 32:
        return B * Do * cd(Ho, 8 * r) * cd(Wo, tw)
    mg = cd(Cout, 64)
    big = B * cd(Do, 2) * cd(Ho, 8 * r) * cd(Wo, tw) * mg
    if big >= 256:
        return big
    small = B * Do * cd(Ho, 4 * r) * cd(Wo, tw) * mg  # quarter-size bricks, then 32-channel workgroups
    return small if small >= 256 else 2 * small


def conv3d_fwd(x, w, bias, k, stride, pad, wmode=0, prelu_weight=None, addend=None):
    """fs_conv3d_fwd: y = conv3d(x, W, bias, stride, pad) with W = w (wmode 0, [Cout,Cin,k,k,k]) or the
    flipped transpose of w (wmode 1, w [Cin,Cout,k,k,k]: input gradient of a stride-1 same conv).
    With `prelu_weight` (wmode 0): returns (y, prelu(y) [+ addend]), both written by the convolution's
    epilogue.  `addend` without `prelu_weight`: y = conv + bias + addend."""
    x = _need_cuda_f32("x", x, 5)
    w = _need_cuda_f32("w", w, 5)
    B, Cin = x.shape[:2]
    Cout = w.shape[1] if wmode else w.shape[0]
    if (w.shape[0] if wmode else w.shape[1]) != Cin or tuple(w.shape[2:]) != (k, k, k):
        raise ValueError("weight %s does not fit input %s (wmode %d)" % (tuple(w.shape), tuple(x.shape), wmode))
    if bias is not None:
        bias = _need_cuda_f32("bias", bias, 1)
        if bias.numel() != Cout:
            raise ValueError("bias must have %d elements" % Cout)
    Di, Hi, Wi = x.shape[2:]
    Do, Ho, Wo = [(n + 2 * pad - k) // stride + 1 for n in (Di, Hi, Wi)]
    if min(Do, Ho, Wo) < 1:
        raise ValueError("convolution output is empty for input %s" % (tuple(x.shape),))
    y = x.new_empty((B, Cout, Do, Ho, Wo))
    wp, ws = _prepared_fwd(w, x.data_ptr(), B, Cin, Cout, (Di, Hi, Wi), (Do, Ho, Wo), k, stride, pad, wmode)
    nb, fq = 4 * (x.numel() + y.numel()), 2 * y.numel() * Cin * int(k) ** 3
    fl, sym = fq, None
    if int(k) == 3 and int(stride) == 1 and int(pad) == 1:  # the Winograd forms execute fewer multiply-adds
        macs = _fwd_k3_macs(x.data_ptr(), B, Cin, Cout, (Di, Hi, Wi), wmode)
        fl, sym = int(2 * y.numel() * Cin * macs), _fwd_k3_symbol(macs, Wi)
    elif int(k) == 4 and int(stride) == 2 and int(pad) == 1 and not wmode:
        sym = _fwd_k4_symbol(x.data_ptr(), B, Cin, Cout, (Di, Hi, Wi), (Do, Ho, Wo))
        if sym is not None:
            fl = 6 * fq  # matrix-core flops EXECUTED: six bf16 products per fp32 multiply-add (priced against the bf16 peak)
    if addend is not None:
        addend = _need_cuda_f32("addend", addend, 5)
        if addend.shape != y.shape:
            raise ValueError("addend %s must have the output shape %s" % (tuple(addend.shape), tuple(y.shape)))
        nb += 4 * y.numel()
    with torch.cuda.device(x.device):
        if prelu_weight is None and addend is not None:
            _call("fs_conv3d_fwd_add", x.data_ptr(), wp, _ptr(bias), addend.data_ptr(), y.data_ptr(),
                  ws.data_ptr(), B, Cin, Cout, Di, Hi, Wi, Do, Ho, Wo, int(k), int(stride), int(pad), int(wmode),
                  _stream(x), algo_bytes=nb, algo_flops=fl, equiv_flops=fq, record_as="fs_conv3d_fwd", kernel=sym)
            return y
        if prelu_weight is None:
            _call("fs_conv3d_fwd", x.data_ptr(), wp, _ptr(bias), y.data_ptr(), ws.data_ptr(), B, Cin,
                  Cout, Di, Hi, Wi, Do, Ho, Wo, int(k), int(stride), int(pad), int(wmode), _stream(x),
                  algo_bytes=nb, algo_flops=fl, equiv_flops=fq, kernel=sym)
            return y
        if wmode:
            raise ValueError("the fused PReLU epilogue is forward-only (wmode 0)")
        a = _need_cuda_f32("prelu_weight", prelu_weight, 1)
        if a.numel() not in (1, Cout):
            raise ValueError("prelu_weight must have 1 or %d elements" % Cout)
        z = torch.empty_like(y)
        _call("fs_conv3d_fwd_prelu", x.data_ptr(), wp, _ptr(bias), a.data_ptr(), _ptr(addend),
              y.data_ptr(), z.data_ptr(), ws.data_ptr(), B, Cin, Cout, Di, Hi, Wi, Do, Ho, Wo, int(k), int(stride),
              int(pad),
              a.numel(), _stream(x), algo_bytes=nb + 4 * y.numel(), algo_flops=fl, equiv_flops=fq, record_as="fs_conv3d_fwd",
              kernel=sym)
    return y, z


def conv3d_tr_supported(cout, k, stride, padding):
    return (tuple(k) == (4, 4, 4) and tuple(stride) == (2, 2, 2) and tuple(padding) == (1, 1, 1) and
            (cout <= 32 or (cout % 32 == 0 and cout <= 128)))


def conv3d_tr(x, w, bias, out_dhw=None, prelu_weight=None, addend=None):
    """fs_conv3d_tr: ConvTranspose3d(4, 2, 1)(x) with weight w [Cin,Cout,4,4,4]; with out_dhw = the
    input extent of a Conv3d(4, 2, 1) layer and w = that layer's weight, its input gradient.
    With `prelu_weight`: returns (y, prelu(y)), both written by the epilogue.  With `addend` (y's shape):
    y = conv_transpose(x) + bias + addend."""
    x = _need_cuda_f32("x", x, 5)
    w = _need_cuda_f32("w", w, 5)
    B, Cin = x.shape[:2]
    Cout = w.shape[1]
    if w.shape[0] != Cin or tuple(w.shape[2:]) != (4, 4, 4):
        raise ValueError("weight %s does not fit input %s" % (tuple(w.shape), tuple(x.shape)))
    if bias is not None:
        bias = _need_cuda_f32("bias", bias, 1)
        if bias.numel() != Cout:
            raise ValueError("bias must have %d elements" % Cout)
    Di, Hi, Wi = x.shape[2:]
    Do, Ho, Wo = (2 * Di, 2 * Hi, 2 * Wi) if out_dhw is None else tuple(int(v) for v in out_dhw)
    y = x.new_empty((B, Cout, Do, Ho, Wo))
    nws = int(_lib.lib().fs_conv3d_tr_ws_floats(Cin, Cout))
    if nws < 0:
        raise ValueError("fs_conv3d_tr supports up to 32 output channels or 64 / 96 / 128, got %d" % Cout)
    L = _lib.lib()
    has_z = int(prelu_weight is not None)
    wp, ws = _prepared(w, nws, ("tr", B, Cin, Cout, Di, Hi, Wi, Do, Ho, Wo, has_z, x.data_ptr() % 16),
                       lambda jobs, cap, slab: L.fs_conv3d_tr_wprep_jobs(jobs, cap, x.data_ptr(), w.data_ptr(),
                                                                        slab.data_ptr(), B, Cin, Cout, Di, Hi, Wi, Do, Ho,
                                                                        Wo, has_z))
    nb, fq = 4 * (x.numel() + y.numel()), 2 * x.numel() * Cout * 64
    sym = _tr_symbol(x.data_ptr(), B, Cin, Cout, (Di, Hi, Wi), (Do, Ho, Wo), has_z)
    # matrix-core flops EXECUTED: six bf16 products per fp32 multiply-add on the split-bf16 kernels (32 / 16 channel rows)
    fl = fq if sym is None else 6 * 2 * x.numel() * (32 * ((Cout + 31) // 32) if sym.endswith("<false>") else 16) * 64
    with torch.cuda.device(x.device):
        if addend is not None:
            if prelu_weight is not None:
                raise ValueError("addend and prelu_weight are mutually exclusive")
            addend = _need_cuda_f32("addend", addend, 5)
            if addend.shape != y.shape:
                raise ValueError("addend %s must have the output shape %s" % (tuple(addend.shape), tuple(y.shape)))
            _call("fs_conv3d_tr_add", x.data_ptr(), wp, _ptr(bias), addend.data_ptr(), y.data_ptr(),
                  ws.data_ptr(), B, Cin, Cout, Di, Hi, Wi, Do, Ho, Wo, _stream(x), algo_bytes=nb + 4 * y.numel(),
                  algo_flops=fl, equiv_flops=fq, record_as="fs_conv3d_tr", kernel=sym)
            return y
        if prelu_weight is None:
            _call("fs_conv3d_tr", x.data_ptr(), wp, _ptr(bias), y.data_ptr(), ws.data_ptr(), B, Cin,
                  Cout, Di, Hi, Wi, Do, Ho, Wo, _stream(x), algo_bytes=nb, algo_flops=fl, equiv_flops=fq, kernel=sym)
            return y
        a = _need_cuda_f32("prelu_weight", prelu_weight, 1)
        if a.numel() not in (1, Cout):
            raise ValueError("prelu_weight must have 1 or %d elements" % Cout)
        z = torch.empty_like(y)
        _call("fs_conv3d_tr_prelu", x.data_ptr(), wp, _ptr(bias), a.data_ptr(), y.data_ptr(),
              z.data_ptr(), ws.data_ptr(), B, Cin, Cout, Di, Hi, Wi, Do, Ho, Wo, a.numel(), _stream(x),
              algo_bytes=nb + 4 * y.numel(), algo_flops=fl, equiv_flops=fq, record_as="fs_conv3d_tr", kernel=sym)
    return y, z


# --------------------------------------------------------------------------------------------
# a9 'SSIM' branch: weighted SSIM + masked reduction, fused (upflow.py:141-196, 285-289)
# --------------------------------------------------------------------------------------------
class _WSSIMLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, weight, use_occ):
        x = _need_cuda_f32("x", x, 4)
        y = _need_cuda_f32("y", y, 4)
        weight = _need_cuda_f32("occ_mask", weight, 4)
        B, C, H, W = x.shape
        if y.shape != x.shape or tuple(weight.shape) != (B, 1, H, W):
            raise ValueError("weighted SSIM operands mismatch: %s %s %s" %
                             (tuple(x.shape), tuple(y.shape), tuple(weight.shape)))
        sums = x.new_empty(2)
        ws = x.new_empty(2 * _REDUCE_BLOCKS)
        with torch.cuda.device(x.device):
            _call("fs_wssim_fwd", x.data_ptr(), y.data_ptr(), weight.data_ptr(), sums.data_ptr(),
                  ws.data_ptr(), B, C, H, W, int(bool(use_occ)), _stream(x),
                  algo_bytes=4 * (2 * x.numel() + weight.numel()))
        if use_occ:
            dS1 = 1.0 / (sums[1] + 1e-6)
        else:
            dS1 = torch.full_like(sums[0], 1.0 / float(B * C * (H - 2) * (W - 2)))
        ctx.save_for_backward(x, y, weight, dS1)
        ctx.use_occ = int(bool(use_occ))
        return sums[0] * dS1

    @staticmethod
    def backward(ctx, gout):
        x, y, weight, dS1 = ctx.saved_tensors
        nx, ny = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (nx or ny):
            return None, None, None, None
        B, C, H, W = x.shape
        coef = (gout * dS1).reshape(1).contiguous()
        gx = torch.empty_like(x) if nx else None
        gy = torch.empty_like(y) if ny else None
        with torch.cuda.device(x.device):
            _call("fs_wssim_bwd", x.data_ptr(), y.data_ptr(), weight.data_ptr(), coef.data_ptr(), _ptr(gx),
                  _ptr(gy), B, C, H, W, ctx.use_occ, _stream(x))
        return gx, gy, None, None


def weighted_ssim_loss(x, y, occ_mask, photo_loss_use_occ):
    """photo_loss_multi_type(..., photo_loss_type='SSIM') in one fused pass."""
    return _WSSIMLoss.apply(x, y, occ_mask, bool(photo_loss_use_occ))
