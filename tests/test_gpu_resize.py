"""-m gpu: SURVEY §8f.1 -- the resize kernels (trilinear down / up forward, the 2-D bilinear pair) against
ATen's F.interpolate, the fused "up-sample flow x scale -> accumulate -> warp" launch against its unfused
composition and against the CPU oracle, and the warp backward that folds in the gradient reaching the flow
from its other consumers.  Reference: Flow-3D/model/IFNet.py:85,88,118-119,190-191, Flow-2D/model/IFNet.py:
89,92,115-116."""
import pytest
import torch
import torch.nn.functional as F

from oracle import warps as owarps

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from opticalflowscivis_amd import ops as o
    return o


def _rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("shape,sf,mul", [((2, 5, 16, 24, 32), 0.5, 1.0), ((1, 6, 16, 8, 24), 0.25, 0.25),
                                          ((1, 2, 9, 11, 14), 0.5, 0.5), ((1, 3, 13, 10, 9), 0.25, 1.0),
                                          ((2, 6, 4, 6, 8), 2.0, 2.0), ((1, 1, 3, 5, 7), 4.0, 4.0),
                                          ((1, 2, 5, 4, 6), 4.0, 1.0)])
def test_interpolate3d_forward_is_atens(ops, shape, sf, mul):
    """mul * F.interpolate(trilinear, align_corners=False): the index / lambda arithmetic and summation order
    of ATen's GPU kernel (what the reference runs), also on odd extents.  Down-scaling (every lambda is 1/2,
    every product exact) is bit-identical to ATen on the GPU; everything agrees with ATen's CPU result to an
    ulp (its builds may contract multiply-adds, this kernel never does).  Backward == autograd's."""
    x = _rnd(shape, 11)
    a = x.clone().requires_grad_()
    ref = F.interpolate(a, scale_factor=sf, mode="trilinear", align_corners=False, recompute_scale_factor=False) * mul
    b = x.to(DEV).requires_grad_()
    out = ops.interpolate3d(b, sf, mul)
    assert out.shape == ref.shape
    if sf < 1:
        gref = F.interpolate(x.to(DEV), scale_factor=sf, mode="trilinear", align_corners=False,
                             recompute_scale_factor=False) * mul
        assert torch.equal(out.detach(), gref)
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= 2.5e-7 * max(1.0, float(ref.abs().max()))
    G = _rnd(ref.shape, 12)
    (ga,) = torch.autograd.grad((ref * G).sum(), [a])
    (gb,) = torch.autograd.grad((out * G.to(DEV)).sum(), [b])
    assert float((gb.cpu() - ga).abs().max()) < 1e-5 * max(1.0, float(ga.abs().max()))


@pytest.mark.parametrize("shape,sf,mul", [((2, 5, 40, 56), 0.25, 1.0), ((2, 4, 40, 56), 0.25, 0.25),
                                          ((1, 9, 80, 112), 0.5, 0.5), ((1, 2, 37, 51), 0.25, 1.0),
                                          ((1, 3, 37, 51), 0.5, 1.0), ((2, 4, 10, 14), 4.0, 4.0),
                                          ((1, 1, 20, 28), 2.0, 1.0), ((1, 4, 9, 13), 2.0, 2.0)])
def test_interpolate2d_vs_aten(ops, shape, sf, mul):
    """The 2-D bilinear pair of Flow-2D's IFBlock: forward == ATen's GPU kernel bit for bit when down-scaling
    (ATen's CPU kernel sums the four weighted corners in a different order: an ulp), to an ulp when
    up-scaling; backward == autograd's."""
    x = _rnd(shape, 21)
    a = x.clone().requires_grad_()
    ref = F.interpolate(a, scale_factor=sf, mode="bilinear", align_corners=False, recompute_scale_factor=False) * mul
    b = x.to(DEV).requires_grad_()
    out = ops.interpolate2d(b, sf, mul)
    assert out.shape == ref.shape
    if sf < 1:
        gref = F.interpolate(x.to(DEV), scale_factor=sf, mode="bilinear", align_corners=False,
                             recompute_scale_factor=False) * mul
        assert torch.equal(out.detach(), gref)
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= 2.5e-7 * max(1.0, float(ref.abs().max()))
    G = _rnd(ref.shape, 22)
    (ga,) = torch.autograd.grad((ref * G).sum(), [a])
    (gb,) = torch.autograd.grad((out * G.to(DEV)).sum(), [b])
    assert float((gb.cpu() - ga).abs().max()) < 1e-5 * max(1.0, float(ga.abs().max()))


def test_resize_entry_points_reject_bad_arguments(ops):
    from opticalflowscivis_amd import _lib
    L = _lib.lib()
    x = torch.zeros(1, 2, 8, 8, 8, device=DEV)
    y = torch.zeros(1, 2, 16, 16, 16, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert L.fs_downsample3d_fwd(x.data_ptr(), y.data_ptr(), 1, 2, 8, 8, 8, 3, 1.0, st) == 3       # factor
    assert L.fs_downsample3d_fwd(x.data_ptr(), y.data_ptr(), 1, 2, 8, 8, 3, 4, 1.0, st) == 2       # too small
    assert L.fs_downsample3d_fwd(None, y.data_ptr(), 1, 2, 8, 8, 8, 2, 1.0, st) == 1
    assert L.fs_resize2d_fwd(x.data_ptr(), y.data_ptr(), 1, 2, 8, 8, 17, 16, 2, 1, 1.0, st) == 2   # out != 2 in
    assert L.fs_resize2d_bwd(x.data_ptr(), y.data_ptr(), 1, 2, 8, 8, 4, 4, 5, 0, 1.0, st) == 3
    assert L.fs_interp3d_bwd_scaled(y.data_ptr(), x.data_ptr(), None, 1, 2, 8, 8, 8, 16, 16, 16, 2, 1, 2.0, st) == 3
    assert L.fs_upsample_warp3d_pair_fwd(x.data_ptr(), x.data_ptr(), x.data_ptr(), None, y.data_ptr(),
                                         y.data_ptr(), y.data_ptr(), 1, 1, None, 8, 8, 8, 3, 2.0, st) == 3
    with pytest.raises(ValueError):
        ops.upsample_warp_pair(y[:, :1], y[:, :1], x, None, 2)  # delta must have 6 channels


def _composed(ops, img0, img1, delta, prev, factor):
    flow = ops.upsample3d_scale_add(delta, prev, factor, float(factor))
    w0, w1 = ops.warp_pair(img0, img1, flow)
    return flow, w0, w1


@pytest.mark.parametrize("small,factor,with_prev,img_extra", [((6, 10, 8), 4, True, 0), ((12, 16, 20), 2, True, 0),
                                                              ((5, 7, 9), 2, False, 0), ((8, 8, 8), 4, True, 8),
                                                              ((3, 18, 33), 2, True, 0)])
def test_upsample_warp_pair_vs_composition_and_oracle(ops, small, factor, with_prev, img_extra):
    """One launch == fs_upsample3d_scale_add followed by fs_warp3d_pair_fwd, bit for bit (flow and both warps);
    == the oracle (CPU F.interpolate + warp3d_ref) at fp32 tolerance; gradients w.r.t. delta and the running
    flow == the composition's, including a second consumer of the flow (the path the fused backward folds
    in).  img_extra > 0: frames larger than the flow (extents that are not multiples of 16)."""
    B = 2
    full = tuple(factor * n for n in small)
    iext = tuple(n + img_extra for n in full)
    delta = _rnd((B, 6) + small, 31, 0.8)
    prev = _rnd((B, 6) + full, 32, 1.5) if with_prev else None
    img0, img1 = torch.rand((B, 1) + iext, generator=torch.Generator().manual_seed(33)), \
        torch.rand((B, 1) + iext, generator=torch.Generator().manual_seed(34))
    d1, d2 = delta.to(DEV).requires_grad_(), delta.to(DEV).requires_grad_()
    p1 = prev.to(DEV).requires_grad_() if with_prev else None
    p2 = prev.to(DEV).requires_grad_() if with_prev else None
    i0, i1 = img0.to(DEV), img1.to(DEV)
    (f_f, f_f2, f_f3), a0, a1 = ops.upsample_warp_pair(i0, i1, d1, p1, factor)
    f_c, c0, c1 = _composed(ops, i0, i1, d2, p2, factor)
    assert torch.equal(f_f2, f_c) and torch.equal(f_f3, f_c)
    assert torch.equal(f_f, f_c) and torch.equal(a0, c0) and torch.equal(a1, c1)
    # oracle
    fo = F.interpolate(delta, scale_factor=factor, mode="trilinear", align_corners=False,
                       recompute_scale_factor=False) * factor
    fo = prev + fo if with_prev else fo
    assert float((f_f.detach().cpu() - fo).abs().max()) < 1e-5
    o0, o1 = owarps.warp3d_ref(img0, fo[:, :3]), owarps.warp3d_ref(img1, fo[:, 3:6])
    assert float((a0.detach().cpu() - o0).abs().max()) < 2e-5 and float((a1.detach().cpu() - o1).abs().max()) < 2e-5
    # gradients: warps + a second consumer of the flow
    G0, G1 = _rnd(a0.shape, 35).to(DEV), _rnd(a1.shape, 36).to(DEV)
    Gf = _rnd(f_f.shape, 37, 0.3).to(DEV)
    ins1 = [d1] + ([p1] if with_prev else [])
    ins2 = [d2] + ([p2] if with_prev else [])
    other = _rnd((B, 5) + full, 38).to(DEV)
    Gc = _rnd((B, 11) + full, 39, 0.2).to(DEV)

    def consumers(fa, fb, fc):  # next block's input concatenation, its accumulation, the distillation term
        return (torch.cat((other, fa), 1) * Gc).sum() + (fb * Gf).sum() + (fc.square() * 0.25).sum()

    for use_flow in (True, False):
        l1 = (a0 * G0).sum() + (a1 * G1).sum() + (consumers(f_f, f_f2, f_f3) if use_flow else 0)
        l2 = (c0 * G0).sum() + (c1 * G1).sum() + (consumers(f_c, f_c, f_c) if use_flow else 0)
        g1 = torch.autograd.grad(l1, ins1, retain_graph=True)
        g2 = torch.autograd.grad(l2, ins2, retain_graph=True)
        for x, y in zip(g1, g2):
            assert float((x - y).abs().max()) < 2e-5 * max(1.0, float(y.abs().max()))
    # the warped frames consumed through torch.cat (the next block's input): their gradients arrive as channel
    # slices of a wider tensor and are read in place, batch stride 3 * D*H*W
    Gw = _rnd((B, 3) + tuple(a0.shape[2:]), 40).to(DEV)
    filler = _rnd(tuple(a0.shape), 41).to(DEV)
    l1 = (torch.cat((filler, a0, a1), 1) * Gw).sum() + consumers(f_f, f_f2, f_f3)
    l2 = (torch.cat((filler, c0, c1), 1) * Gw).sum() + consumers(f_c, f_c, f_c)
    for x, y in zip(torch.autograd.grad(l1, ins1, retain_graph=True), torch.autograd.grad(l2, ins2, retain_graph=True)):
        assert float((x - y).abs().max()) < 2e-5 * max(1.0, float(y.abs().max()))
    # only the flow consumer (no gradient reaches the warps)
    g1 = torch.autograd.grad((f_f * Gf).sum(), ins1)
    g2 = torch.autograd.grad((f_c * Gf).sum(), ins2)
    for x, y in zip(g1, g2):
        assert float((x - y).abs().max()) < 2e-5 * max(1.0, float(y.abs().max()))


@pytest.mark.parametrize("only_images", [False, True])
def test_upsample_warp_pair_image_gradients(ops, only_images):
    """frames that require grad get their gradient from the fused node too (IFNet never asks, a caller of
    the public op may): == warp_pair(upsample3d_scale_add(...))'s, also when ONLY the frames require grad."""
    B, small, factor = 2, (5, 6, 8), 2
    full = tuple(factor * n for n in small)
    delta, prev = _rnd((B, 6) + small, 51, 0.8), _rnd((B, 6) + full, 52, 1.5)
    img0 = torch.rand((B, 2) + full, generator=torch.Generator().manual_seed(53))
    img1 = torch.rand((B, 2) + full, generator=torch.Generator().manual_seed(54))
    leaves = []
    for _ in range(2):
        leaves.append([img0.to(DEV).requires_grad_(), img1.to(DEV).requires_grad_(),
                       delta.to(DEV).requires_grad_(not only_images), prev.to(DEV).requires_grad_(not only_images)])
    (fa, fb, fc), a0, a1 = ops.upsample_warp_pair(leaves[0][0], leaves[0][1], leaves[0][2], leaves[0][3], factor)
    f_c, c0, c1 = _composed(ops, leaves[1][0], leaves[1][1], leaves[1][2], leaves[1][3], factor)
    G0, G1 = _rnd(a0.shape, 55).to(DEV), _rnd(a1.shape, 56).to(DEV)
    Gf = _rnd(fa.shape, 57, 0.3).to(DEV)
    l1 = (a0 * G0).sum() + (a1 * G1).sum() + ((fb * Gf).sum() if not only_images else 0)
    l2 = (c0 * G0).sum() + (c1 * G1).sum() + ((f_c * Gf).sum() if not only_images else 0)
    n = 2 if only_images else 4
    g1 = torch.autograd.grad(l1, leaves[0][:n])
    g2 = torch.autograd.grad(l2, leaves[1][:n])
    for x, y in zip(g1, g2):
        assert float(y.abs().max()) > 0
        assert float((x - y).abs().max()) < 2e-5 * max(1.0, float(y.abs().max()))
    # one frame only
    (fa, fb, fc), a0, a1 = ops.upsample_warp_pair(leaves[0][0], img1.to(DEV), delta.to(DEV), prev.to(DEV), factor)
    (g,) = torch.autograd.grad((a0 * G0).sum() + (a1 * G1).sum(), [leaves[0][0]])
    assert float((g - g2[0]).abs().max()) < 2e-5 * max(1.0, float(g2[0].abs().max()))


def test_warp_pair_acc_folds_in_the_other_consumers_gradient(ops):
    g = torch.Generator().manual_seed(41)
    B, D, H, W = 2, 8, 70, 37
    img0, img1 = torch.rand(B, 1, D, H, W, generator=g).to(DEV), torch.rand(B, 1, D, H, W, generator=g).to(DEV)
    flow = ((torch.rand(B, 6, D, H, W, generator=g) * 2 - 1) * 2.0)
    fa, fb = flow.to(DEV).requires_grad_(), flow.to(DEV).requires_grad_()
    w0, w1, (fout, fout2, fout3) = ops.warp_pair_acc(img0, img1, fa)
    r0, r1 = ops.warp_pair(img0, img1, fb)
    assert torch.equal(w0, r0) and torch.equal(w1, r1) and torch.equal(fout, fa) and torch.equal(fout3, fa)
    G0, G1 = torch.randn(w0.shape, generator=g).to(DEV), torch.randn(w1.shape, generator=g).to(DEV)
    Gf = torch.randn(flow.shape, generator=g).to(DEV)
    (ga,) = torch.autograd.grad((w0 * G0).sum() + (w1 * G1).sum() + (fout * Gf).sum() + (fout.square() * 0.5).sum(),
                                [fa], retain_graph=True)
    (gb,) = torch.autograd.grad((r0 * G0).sum() + (r1 * G1).sum() + (fb * Gf).sum() + (fb.square() * 0.5).sum(),
                                [fb], retain_graph=True)
    assert float((ga - gb).abs().max()) < 1e-5 * max(1.0, float(gb.abs().max()))
    # three consumers through the three aliases; one through torch.cat: its gradient is a channel slice of an
    # 11-channel tensor and is read in place (batch stride 11 * D*H*W)
    other = torch.randn(B, 5, D, H, W, generator=g).to(DEV)
    Gc = torch.randn(B, 11, D, H, W, generator=g).to(DEV)
    (ga,) = torch.autograd.grad((w0 * G0).sum() + (w1 * G1).sum() + (torch.cat((other, fout), 1) * Gc).sum() +
                                (fout2 * Gf).sum() + (fout3.square() * 0.5).sum(), [fa], retain_graph=True)
    (gb,) = torch.autograd.grad((r0 * G0).sum() + (r1 * G1).sum() + (torch.cat((other, fb), 1) * Gc).sum() +
                                (fb * Gf).sum() + (fb.square() * 0.5).sum(), [fb], retain_graph=True)
    assert float((ga - gb).abs().max()) < 1e-5 * max(1.0, float(gb.abs().max()))
    # the warped frames through torch.cat as well (gradients = channel slices, batch stride 3 * D*H*W)
    Gw = torch.randn(B, 3, D, H, W, generator=g).to(DEV)
    (ga,) = torch.autograd.grad((torch.cat((img0, w0, w1), 1) * Gw).sum() + (fout2 * Gf).sum(), [fa], retain_graph=True)
    (gb,) = torch.autograd.grad((torch.cat((img0, r0, r1), 1) * Gw).sum() + (fb * Gf).sum(), [fb], retain_graph=True)
    assert float((ga - gb).abs().max()) < 1e-5 * max(1.0, float(gb.abs().max()))
    # each path alone
    (ga,) = torch.autograd.grad((w0 * G0).sum(), [fa], retain_graph=True)
    (gb,) = torch.autograd.grad((r0 * G0).sum(), [fb], retain_graph=True)
    assert float((ga - gb).abs().max()) < 1e-5 * max(1.0, float(gb.abs().max()))
    (ga,) = torch.autograd.grad((fout * Gf).sum(), [fa], retain_graph=True)
    assert torch.equal(ga, Gf)
    # in place at the C-ABI: grad_flow_add may be grad_flow6 itself
    from opticalflowscivis_amd import _lib
    acc = Gf.clone()
    st = torch.cuda.current_stream().cuda_stream
    rc = _lib.lib().fs_warp3d_pair_bwd_acc(img0.data_ptr(), img1.data_ptr(), fb.data_ptr(), G0.data_ptr(),
                                           G1.data_ptr(), None, None, acc.data_ptr(), acc.data_ptr(), B, 1, None,
                                           D, H, W, st)
    assert rc == 0
    (gw,) = torch.autograd.grad((r0 * G0).sum() + (r1 * G1).sum(), [fb])
    assert float((acc - (gw + Gf)).abs().max()) < 1e-5 * max(1.0, float(gw.abs().max()))


def test_warp2d_rife_input_extent_differs_from_flow_extent(ops):
    """Flow-2D/model/warplayer.py builds its grid from the FLOW's shape and normalises by the INPUT's: IFNet-2D
    calls it with frames of 146 rows and a 144-row flow.  Single warp and the pair launch vs the oracle."""
    g = torch.Generator().manual_seed(51)
    x = torch.rand(2, 1, 26, 40, generator=g)
    y = torch.rand(2, 1, 26, 40, generator=g)
    f = ((torch.rand(2, 4, 24, 36, generator=g) * 2 - 1) * 2.5)
    xo, fo = x.clone().requires_grad_(), f.clone().requires_grad_()
    ref0 = owarps.warp2d_rife_ref(xo, fo[:, :2])
    ref1 = owarps.warp2d_rife_ref(y, fo[:, 2:4])
    G0, G1 = torch.randn(ref0.shape, generator=g), torch.randn(ref1.shape, generator=g)
    gx_ref, gf_ref = torch.autograd.grad((ref0 * G0).sum() + (ref1 * G1).sum(), [xo, fo])
    xd, fd = x.to(DEV).requires_grad_(), f.to(DEV).requires_grad_()
    w0, w1 = ops.warp_pair(xd, y.to(DEV), fd)
    assert w0.shape == ref0.shape == (2, 1, 24, 36)
    assert float((w0.detach().cpu() - ref0.detach()).abs().max()) < 2e-5
    assert float((w1.detach().cpu() - ref1.detach()).abs().max()) < 2e-5
    gx, gf = torch.autograd.grad((w0 * G0.to(DEV)).sum() + (w1 * G1.to(DEV)).sum(), [xd, fd])
    assert float((gx.cpu() - gx_ref).abs().max()) < 2e-4 and float((gf.cpu() - gf_ref).abs().max()) < 2e-4
    s = ops.warp2d(x.to(DEV), f[:, :2].contiguous().to(DEV))
    assert float((s.cpu() - ref0.detach()).abs().max()) < 2e-5
    with pytest.raises(ValueError):  # only the RIFE warp defines it
        ops.warp2d_pwc(x.to(DEV), f[:, :2].contiguous().to(DEV), with_mask=False)


def test_downsample_reads_its_pieces_in_place(ops):
    """ops.interpolate3d_cat == F.interpolate(torch.cat(pieces, 1)) bit for bit (same kernel, channels read where they
    lie: contiguous frames and channel slices of a wider tensor), gradients equal to the concatenated path's."""
    g = torch.Generator().manual_seed(17)
    B, D, H, W = 2, 16, 24, 40
    wide = torch.randn(B, 4, D, H, W, generator=g).to(DEV)
    base = [torch.randn(B, 1, D, H, W, generator=g).to(DEV), wide[:, 1:2], wide[:, 2:4], torch.randn(B, 1, D, H, W, generator=g).to(DEV)]
    for factor in (2, 4):
        pa = [t.clone().requires_grad_(i != 0) for i, t in enumerate(base)]
        pb = [t.clone().requires_grad_(i != 0) for i, t in enumerate(base)]
        ya = ops.interpolate3d_cat(tuple(pa), factor)
        yb = ops.interpolate3d(torch.cat(pb, 1), 1.0 / factor)
        assert ya is not None and torch.equal(ya, yb)
        ref = F.interpolate(torch.cat([t.cpu() for t in base], 1), scale_factor=1.0 / factor, mode="trilinear",
                            align_corners=False, recompute_scale_factor=False)
        assert float((ya.detach().cpu() - ref).abs().max()) < 1e-6
        G = torch.randn(ya.shape, generator=g).to(DEV)
        ga = torch.autograd.grad((ya * G).sum(), pa[1:])
        gb = torch.autograd.grad((yb * G).sum(), pb[1:])
        for a, b in zip(ga, gb):
            assert torch.equal(a, b)
    assert ops.interpolate3d_cat(tuple(base), 3) is None
