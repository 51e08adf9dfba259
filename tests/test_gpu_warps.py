"""-m gpu: HIP warps (through the C-ABI) against the golden vectors and the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import warps as owarps

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
# fp32 tolerances.  Coordinates agree with the reference to ~1e-5 px (different but equivalent
# fp32 evaluation order of linspace/unnormalise), images are in [0,1] with O(1) gradients.
OUT_ATOL = 2e-5
GRAD_ATOL = 2e-4


def T(a, grad=False):
    t = torch.from_numpy(np.asarray(a)).clone().to(DEV)
    return t.requires_grad_() if grad else t


def maxerr(a, b):
    return float((a.detach().cpu() - torch.as_tensor(b)).abs().max())


def frac_bad(a, b, atol, rtol=1e-4):
    b = torch.as_tensor(b)
    err = (a.detach().cpu() - b).abs()
    return float((err > atol + rtol * b.abs()).float().mean())


@pytest.fixture(scope="module")
def ops():
    from opticalflowscivis_amd import ops as o
    return o


def _golden_case(fn, g, pre, xn="x", fnm="f", gxn="gx", gfn="gf", **kw):
    x, f = T(g[pre + xn], True), T(g[pre + fnm], True)
    out = fn(x, f, **kw)
    assert maxerr(out, g[pre + "out"]) < OUT_ATOL
    gx, gf = torch.autograd.grad((out * T(g[pre + "G"])).sum(), [x, f])
    assert maxerr(gx, g[pre + gxn]) < GRAD_ATOL
    assert frac_bad(gf, g[pre + gfn], GRAD_ATOL) == 0.0


def test_warp3d_golden(ops, golden):
    g = golden("rife_ops")
    for tag in ("nc", "cu", "tile", "mixed"):
        _golden_case(ops.warp3d, g, "w3_%s_" % tag)
    out = ops.warp3d(T(g["w3_zero_x"]), torch.zeros(1, 3, 5, 6, 7, device=DEV))
    assert maxerr(out, g["w3_zero_out"]) < OUT_ATOL


def test_warp2d_rife_golden(ops, golden):
    g = golden("rife_ops")
    for tag in ("a", "b"):
        _golden_case(ops.warp2d, g, "w2_%s_" % tag)
    out = ops.warp2d(T(g["w2_zero_x"]), torch.zeros(1, 2, 8, 12, device=DEV))
    assert maxerr(out, g["w2_zero_out"]) < OUT_ATOL


def test_warp2d_pwc_golden(ops, golden):
    g = golden("upflow_ops")
    x, f = T(g["pwcmask_x"], True), T(g["pwcmask_f"], True)
    out = ops.warp2d_pwc(x, f, with_mask=False)
    assert maxerr(out, g["pwc_out"]) < OUT_ATOL
    gx, gf = torch.autograd.grad((out * T(g["pwc_G"])).sum(), [x, f])
    assert maxerr(gx, g["pwc_gx"]) < GRAD_ATOL
    assert maxerr(gf, g["pwc_gf"]) < GRAD_ATOL
    # validity mask: parity is defined away from the fp32-borderline pixels (weight sum == 1 +- ulp)
    xc, fc = torch.from_numpy(g["pwcmask_x"]), torch.from_numpy(g["pwcmask_f"])
    sure = ~owarps.pwc_mask_borderline(xc, fc)  # [B,1,H,W]
    out = ops.warp2d_pwc(x, f, with_mask=True)
    ref = torch.from_numpy(g["pwcmask_out"])
    err = (out.detach().cpu() - ref).abs()
    assert float((err * sure).max()) < OUT_ATOL
    # on borderline pixels the result must be one of the two legal values: 0 or the unmasked sample
    unm = torch.from_numpy(g["pwc_out"])
    o = out.detach().cpu()
    legal = ((o - unm).abs() < OUT_ATOL) | (o.abs() < OUT_ATOL)
    assert bool(legal.all())
    gx, gf = torch.autograd.grad((out * T(g["pwcmask_G"])).sum(), [x, f])
    gerr = (gf.cpu() - torch.from_numpy(g["pwcmask_gf"])).abs()
    assert float((gerr * sure).max()) < GRAD_ATOL
    # grad_in is a scatter: a borderline pixel whose mask flipped adds / removes its four taps, so the
    # golden total is only comparable after taking those pixels out of the upstream gradient on BOTH
    # sides.  The mask has no derivative, so with G * sure the two gradients must agree to fp32 noise;
    # the oracle side is the golden-pinned restatement (tests/test_oracle_golden.py).
    Gs = torch.from_numpy(g["pwcmask_G"]) * sure
    xo, fo = xc.clone().requires_grad_(), fc.clone().requires_grad_()
    gx_ref, gf_ref = torch.autograd.grad((owarps.warp2d_pwc_ref(xo, fo, True) * Gs).sum(), [xo, fo])
    gx2, gf2 = torch.autograd.grad((ops.warp2d_pwc(x, f, with_mask=True) * Gs.to(DEV)).sum(), [x, f])
    assert maxerr(gx2, gx_ref) < GRAD_ATOL
    assert maxerr(gf2, gf_ref) < GRAD_ATOL
    # and the golden total itself: what differs is bounded by the borderline pixels' own contributions
    assert float((gx.cpu() - torch.from_numpy(g["pwcmask_gx"])).abs().sum()) <= \
        float((torch.from_numpy(g["pwcmask_G"]).abs() * (~sure)).sum()) + GRAD_ATOL * gx.numel()


def test_warp2d_dilated_golden(ops, golden):
    g = golden("upflow_ops")
    for tag in ("s0", "s1"):
        pre = "dil_%s_" % tag
        I, f = T(g[pre + "I"], True), T(g[pre + "f"], True)
        out = ops.warp2d_dilated(I, f, T(g[pre + "start"]))
        assert maxerr(out, g[pre + "out"]) < OUT_ATOL
        gI, gf = torch.autograd.grad((out * T(g[pre + "G"])).sum(), [I, f])
        assert maxerr(gI, g[pre + "gI"]) < GRAD_ATOL
        assert maxerr(gf, g[pre + "gf"]) < GRAD_ATOL


@pytest.mark.parametrize("shape", [(2, 1, 24, 70, 50), (1, 3, 17, 33, 65), (1, 1, 64, 64, 64)])
def test_warp3d_vs_oracle(ops, shape):
    B, C, D, H, W = shape
    g = torch.Generator().manual_seed(D * 1000 + H)
    x = torch.rand(shape, generator=g)
    f = (torch.rand(B, 3, D, H, W, generator=g) * 2 - 1) * 3.0
    G = torch.randn(shape, generator=g)
    xr, fr = x.clone().requires_grad_(), f.clone().requires_grad_()
    ref = owarps.warp3d_ref(xr, fr)
    gxr, gfr = torch.autograd.grad((ref * G).sum(), [xr, fr])
    xd, fd = x.to(DEV).requires_grad_(), f.to(DEV).requires_grad_()
    out = ops.warp3d(xd, fd)
    gx, gf = torch.autograd.grad((out * G.to(DEV)).sum(), [xd, fd])
    assert maxerr(out, ref) < OUT_ATOL
    assert frac_bad(gx, gxr, GRAD_ATOL) == 0.0
    # a coordinate within fp32 noise of a cell boundary may pick the neighbouring cell: the sample
    # is continuous there but d/dflow is not, so allow a vanishing fraction of such voxels
    assert frac_bad(gf, gfr, GRAD_ATOL) < 1e-5
    # flow-only backward (the training path: images carry no grad) takes the no-atomics kernel
    fd2 = f.to(DEV).requires_grad_()
    out2 = ops.warp3d(x.to(DEV), fd2)
    (gf2,) = torch.autograd.grad((out2 * G.to(DEV)).sum(), [fd2])
    assert torch.equal(out2, out)
    assert maxerr(gf2, gf.cpu()) < 1e-6


@pytest.mark.parametrize("mode", ["rife", "pwc", "photo", "dilated"])
def test_warp2d_vs_oracle(ops, mode):
    B, C, H, W = 3, 2, 37, 130
    g = torch.Generator().manual_seed(11)
    x = torch.rand(B, C, H, W, generator=g)
    f = (torch.rand(B, 2, H, W, generator=g) * 2 - 1) * 4.0
    G = torch.randn(B, C, H, W, generator=g)
    hip = {"rife": ops.warp2d, "pwc": lambda a, b: ops.warp2d_pwc(a, b, False),
           "photo": ops.warp2d_photo, "dilated": ops.warp2d_dilated}[mode]
    ora = {"rife": owarps.warp2d_rife_ref, "pwc": lambda a, b: owarps.warp2d_pwc_ref(a, b, False),
           "photo": owarps.warp2d_photo_ref, "dilated": owarps.warp2d_dilated_ref}[mode]
    xr, fr = x.clone().requires_grad_(), f.clone().requires_grad_()
    ref = ora(xr, fr)
    gxr, gfr = torch.autograd.grad((ref * G).sum(), [xr, fr])
    xd, fd = x.to(DEV).requires_grad_(), f.to(DEV).requires_grad_()
    out = hip(xd, fd)
    gx, gf = torch.autograd.grad((out * G.to(DEV)).sum(), [xd, fd])
    assert maxerr(out, ref) < OUT_ATOL
    assert frac_bad(gx, gxr, GRAD_ATOL) == 0.0
    assert frac_bad(gf, gfr, GRAD_ATOL) < 1e-4


def test_warp3d_full_size_properties(ops):
    """BASELINE C4 size (B=2, 256^3): size-independent properties instead of an oracle run."""
    S = 256
    x = torch.rand(2, 1, S, S, S, device=DEV)
    z = torch.zeros(2, 3, S, S, S, device=DEV)
    out = ops.warp3d(x, z)
    # zero flow = the reference's axis rotation out[d,h,w] = in[w,d,h].  Not exact in the reference
    # either: its fp32 linspace/unnormalise chain lands within ~3e-5 px of the integer at size 256,
    # and white-noise data has O(1) slope per voxel.
    assert float((out - x.permute(0, 1, 3, 4, 2)).abs().max()) < 1e-4
    # integer flow = rotated, shifted and border-clamped copy
    z[:, 0] = 3.0
    z[:, 1] = -2.0
    z[:, 2] = 1.0
    out = ops.warp3d(x, z)
    idx = torch.arange(S, device=DEV)
    iw = (idx + 1).clamp(0, S - 1)   # w + F2 -> input D index
    idd = (idx - 2).clamp(0, S - 1)  # d + F1 -> input H index
    ih = (idx + 3).clamp(0, S - 1)   # h + F0 -> input W index
    exp = x[:, :, iw][:, :, :, idd][:, :, :, :, ih].permute(0, 1, 3, 4, 2)
    assert float((out - exp).abs().max()) < 1e-4
    # linearity in the input
    y = torch.rand_like(x)
    f = (torch.rand(2, 3, S, S, S, device=DEV) * 2 - 1) * 4
    lhs = ops.warp3d(2.0 * x - 3.0 * y, f)
    rhs = 2.0 * ops.warp3d(x, f) - 3.0 * ops.warp3d(y, f)
    assert float((lhs - rhs).abs().max()) < 1e-4


def test_warp3d_pair_matches_singles(ops):
    """The pair launch (IFNet call site) == two single warps on the flow halves, bit for bit."""
    g = torch.Generator().manual_seed(4)
    B, D, H, W = 2, 10, 70, 40
    i0, i1 = torch.rand(B, 1, D, H, W, generator=g).to(DEV), torch.rand(B, 1, D, H, W, generator=g).to(DEV)
    f = ((torch.rand(B, 6, D, H, W, generator=g) * 2 - 1) * 2).to(DEV).requires_grad_()
    G0, G1 = torch.randn(B, 1, D, H, W, generator=g).to(DEV), torch.randn(B, 1, D, H, W, generator=g).to(DEV)
    a0, a1 = ops.warp_pair(i0, i1, f)
    (gp,) = torch.autograd.grad((a0 * G0).sum() + (a1 * G1).sum(), [f])
    f2 = f.detach().clone().requires_grad_()
    b0, b1 = ops.warp3d(i0, f2[:, :3]), ops.warp3d(i1, f2[:, 3:6])
    (gs,) = torch.autograd.grad((b0 * G0).sum() + (b1 * G1).sum(), [f2])
    assert torch.equal(a0, b0) and torch.equal(a1, b1)
    assert torch.equal(gp, gs)
    # 2-D
    i0, i1 = torch.rand(B, 2, H, W, generator=g).to(DEV), torch.rand(B, 2, H, W, generator=g).to(DEV)
    f = ((torch.rand(B, 4, H, W, generator=g) * 2 - 1) * 2).to(DEV).requires_grad_()
    a0, a1 = ops.warp_pair(i0, i1, f)
    (gp,) = torch.autograd.grad(a0.sum() - a1.sum(), [f])
    f2 = f.detach().clone().requires_grad_()
    b0, b1 = ops.warp2d(i0, f2[:, :2]), ops.warp2d(i1, f2[:, 2:4])
    (gs,) = torch.autograd.grad(b0.sum() - b1.sum(), [f2])
    assert torch.equal(a0, b0) and torch.equal(a1, b1) and torch.equal(gp, gs)


def test_operand_validation(ops):
    x = torch.rand(1, 1, 8, 8, 8, device=DEV)
    with pytest.raises(ValueError):
        ops.warp3d(x, torch.zeros(1, 2, 8, 8, 8, device=DEV))
    with pytest.raises(ValueError):
        ops.warp3d(x.double(), torch.zeros(1, 3, 8, 8, 8, device=DEV))
    with pytest.raises(ValueError):
        ops.warp3d(x.cpu(), torch.zeros(1, 3, 8, 8, 8))
    with pytest.raises(ValueError):  # batch mismatch (a different EXTENT is legal for the RIFE warp only)
        ops.warp2d(torch.rand(1, 1, 8, 8, device=DEV), torch.zeros(2, 2, 8, 8, device=DEV))
    with pytest.raises(ValueError):
        ops.warp2d_pwc(torch.rand(1, 1, 8, 8, device=DEV), torch.zeros(1, 2, 8, 9, device=DEV), with_mask=True)


def test_wild_flows_do_not_fault(ops):
    """NaN / Inf / 1e30 flow vectors must stay inside the tensors (a faulting kernel can reset the
    node): every warp, forward and backward, must complete; finite flows next to them stay exact."""
    g = torch.Generator().manual_seed(0)
    bad = torch.tensor([float("nan"), float("inf"), -float("inf"), 1e30, -1e30, 3e9, -3e9, 0.0])

    def poison(f):
        f = f.clone()
        flat = f.view(-1)
        idx = torch.randperm(flat.numel(), generator=g)[:64]
        flat[idx] = bad.repeat(8)
        return f

    x3 = torch.rand(1, 2, 9, 70, 40, generator=g).to(DEV).requires_grad_()
    f3 = poison(torch.randn(1, 3, 9, 70, 40, generator=g)).to(DEV).requires_grad_()
    out = ops.warp3d(x3, f3)
    out.nan_to_num().sum().backward()
    i0 = torch.rand(1, 1, 9, 70, 40, generator=g).to(DEV)
    f6 = poison(torch.randn(1, 6, 9, 70, 40, generator=g)).to(DEV).requires_grad_()
    a, b = ops.warp_pair(i0, i0, f6)
    (a.nan_to_num().sum() + b.nan_to_num().sum()).backward()
    x2 = torch.rand(2, 3, 33, 47, generator=g).to(DEV).requires_grad_()
    for fn in (ops.warp2d, lambda p, q: ops.warp2d_pwc(p, q, True), lambda p, q: ops.warp2d_pwc(p, q, False),
               ops.warp2d_photo, ops.warp2d_dilated):
        f2 = poison(torch.randn(2, 2, 33, 47, generator=g)).to(DEV).requires_grad_()
        o = fn(x2, f2)
        o.nan_to_num().sum().backward()
    torch.cuda.synchronize()
    # voxels with finite flow are unaffected by poisoned neighbours
    f_ok = torch.randn(1, 3, 9, 70, 40, generator=g)
    f_bad = poison(f_ok)
    same = (f_ok == f_bad).all(dim=1, keepdim=True).to(DEV)
    o1, o2 = ops.warp3d(x3.detach()[:, :1], f_ok.to(DEV)), ops.warp3d(x3.detach()[:, :1], f_bad.to(DEV))
    assert torch.equal(o1[same], o2[same])


@pytest.mark.parametrize("shape", [(32, 196, 3, 8), (32, 128, 5, 15), (2, 96, 10, 29), (3, 17, 7, 9), (2, 5, 6, 7),
                                   (1, 64, 19, 57)])
def test_warp2d_channel_sliced_launches_vs_oracle(ops, shape):
    """The coarse UPFlow pyramid levels (few pixels, many channels: SURVEY Appendix B) run the 2-D warp with 4 or 16
    channel slices per pixel; forward, grad_in (atomics) and grad_flow (LDS reduction over the slices) against the
    oracle, masked (a5) and unmasked (a6), at the C3 level shapes and at shapes whose channel count does not divide
    into the slices / whose pixel count does not fill the last workgroup."""
    B, C, H, W = shape
    g = torch.Generator().manual_seed(C * 7 + H)
    x = torch.rand(shape, generator=g)
    f = 1.5 * torch.randn(B, 2, 1, 1, generator=g) + 0.6 * torch.randn(B, 2, H, W, generator=g)
    G = torch.randn(shape, generator=g)
    sure = ~owarps.pwc_mask_borderline(x, f)
    for with_mask in (True, False):
        Gs = G * sure if with_mask else G
        xo, fo = x.clone().requires_grad_(), f.clone().requires_grad_()
        ro = owarps.warp2d_pwc_ref(xo, fo, with_mask)
        rgx, rgf = torch.autograd.grad((ro * Gs).sum(), [xo, fo])
        xd, fd = x.to(DEV).requires_grad_(), f.to(DEV).requires_grad_()
        out = ops.warp2d_pwc(xd, fd, with_mask=with_mask)
        gx, gf = torch.autograd.grad((out * Gs.to(DEV)).sum(), [xd, fd])
        err = (out.detach().cpu() - ro.detach()).abs()
        assert float((err * sure).max() if with_mask else err.max()) < OUT_ATOL
        assert maxerr(gx, rgx) < GRAD_ATOL
        assert float((gf.cpu() - rgf).abs().max()) < GRAD_ATOL * max(1.0, C / 8)  # a sum over C channels
        # flow-only backward (no grad_in): the same grad_flow, bit for bit (same reduction order)
        fd2 = f.to(DEV).requires_grad_()
        (gf2,) = torch.autograd.grad((ops.warp2d_pwc(x.to(DEV), fd2, with_mask=with_mask) * Gs.to(DEV)).sum(), [fd2])
        assert torch.equal(gf2, gf)
    # the RIFE convention through the same sliced kernels
    xo, fo = x.clone().requires_grad_(), f.clone().requires_grad_()
    if H >= 2 and W >= 2:
        ro = owarps.warp2d_rife_ref(xo, fo)
        rgx, rgf = torch.autograd.grad((ro * G).sum(), [xo, fo])
        xd, fd = x.to(DEV).requires_grad_(), f.to(DEV).requires_grad_()
        out = ops.warp2d(xd, fd)
        gx, gf = torch.autograd.grad((out * G.to(DEV)).sum(), [xd, fd])
        assert maxerr(out, ro) < OUT_ATOL and maxerr(gx, rgx) < GRAD_ATOL
        assert frac_bad(gf, rgf, GRAD_ATOL * max(1.0, C / 8)) < 1e-3  # border-clamp kinks at tiny extents


def _occ_compare(ops, ff, fb, a1, a2, scale, mode, eps=2e-4):
    """HIP masks vs oracle masks: identical except where the tested quantity is within `eps` of the
    threshold (the comparison is discontinuous; fp32 evaluation order decides those pixels)."""
    of, ob = ops.occ_check2d(ff.to(DEV), fb.to(DEV), a1, a2, scale, mode)
    rf, rb = owarps.occ_check_ref(ff, fb, a1, a2, scale, mode)
    assert of.shape == rf.shape and set(np.unique(of.cpu().numpy())) <= {0.0, 1.0}
    if mode == "out":
        assert torch.equal(of.cpu(), rf) and torch.equal(ob.cpu(), rb)
        return 0.0
    lf, lb, th = owarps.occ_fb_lhs_thresh(ff, fb, a1, a2, scale)
    bad = 0
    for o, r, l in ((of, rf, lf), (ob, rb, lb)):
        diff = o.cpu() != r
        assert bool(((l - th).abs()[diff] < eps).all()), "mask differs away from the threshold"
        bad += int(diff.sum())
    return bad / (2.0 * rf.numel())


def test_occ_check_golden(ops, golden):
    """§8f.2: fs_occ_check2d against the masks the reference itself produced."""
    g = golden("upflow_next")
    ff, fb = torch.from_numpy(g["occ_ff"]), torch.from_numpy(g["occ_fb"])
    for mode in ("all", "obj", "out"):
        for scale in (1, 4):
            of, ob = ops.occ_check2d(ff.to(DEV), fb.to(DEV), 0.1, 0.5, scale, mode)
            nf = int((of.cpu() != torch.from_numpy(g["occ_%s_s%d_f" % (mode, scale)])).sum())
            nb = int((ob.cpu() != torch.from_numpy(g["occ_%s_s%d_b" % (mode, scale)])).sum())
            assert nf + nb <= 2, (mode, scale, nf, nb)  # 3840 pixels; threshold ties only
            assert _occ_compare(ops, ff, fb, 0.1, 0.5, scale, mode) < 1e-3


def test_occ_check_vs_oracle_c3_shape(ops):
    """C3 shape [32,2,150,450] with the UPFlow defaults (alpha 0.1 / 0.5, 'obj')."""
    g = torch.Generator().manual_seed(5)
    ff = 2.0 * torch.randn(32, 2, 1, 1, generator=g) + torch.randn(32, 2, 150, 450, generator=g)
    fb = -ff + 0.5 * torch.randn(32, 2, 150, 450, generator=g)
    for mode in ("obj", "all", "out"):
        assert _occ_compare(ops, ff, fb, 0.1, 0.5, 1, mode) < 1e-4
    # consistent flows (b = -f, constant) are visible everywhere except where they leave the frame
    c = torch.tensor([3.0, -2.0]).view(1, 2, 1, 1).expand(1, 2, 40, 60).contiguous()
    of, ob = ops.occ_check2d(c.to(DEV), (-c).to(DEV), 0.1, 0.5, 1, "all")
    assert float(of[:, :, 4:-4, 4:-4].min()) == 1.0 and float(ob[:, :, 4:-4, 4:-4].min()) == 1.0
    with pytest.raises(ValueError):
        ops.occ_check2d(c.to(DEV), c[:, :1].to(DEV), 0.1, 0.5)
    # wild flows stay inside the tensors
    w = c.clone(); w[0, 0, 3, 3] = float("nan"); w[0, 1, 5, 5] = float("inf"); w[0, 0, 7, 7] = -1e30
    ops.occ_check2d(w.to(DEV), (-w).to(DEV), 0.1, 0.5, 1, "obj")
    torch.cuda.synchronize()


def test_empty_batch_is_a_no_op(ops):
    """B = 0 (a ragged last shard): empty outputs of the right shape, gradients flow (as empty tensors),
    no kernel launch -- what F.grid_sample / Corr_pyTorch do for an empty batch."""
    x3 = torch.zeros(0, 2, 4, 5, 6, device=DEV, requires_grad=True)
    f3 = torch.zeros(0, 3, 4, 5, 6, device=DEV, requires_grad=True)
    o = ops.warp3d(x3, f3)
    assert tuple(o.shape) == (0, 2, 4, 5, 6)
    o.sum().backward()
    assert x3.grad.shape == x3.shape and f3.grad.shape == f3.shape
    x2 = torch.zeros(0, 3, 8, 9, device=DEV, requires_grad=True)
    f2 = torch.zeros(0, 2, 8, 9, device=DEV)
    assert tuple(ops.warp2d(x2, f2).shape) == (0, 3, 8, 9)
    assert tuple(ops.warp2d_pwc(x2, f2, True).shape) == (0, 3, 8, 9)
    a, b = ops.warp_pair(torch.zeros(0, 1, 8, 9, device=DEV), torch.zeros(0, 1, 8, 9, device=DEV),
                         torch.zeros(0, 4, 8, 9, device=DEV))
    assert tuple(a.shape) == tuple(b.shape) == (0, 1, 8, 9)
    c = ops.corr2d(torch.zeros(0, 5, 6, 7, device=DEV), torch.zeros(0, 5, 6, 7, device=DEV), 4)
    assert tuple(c.shape) == (0, 81, 6, 7)
    # mismatched batches are still an error, not silently empty
    with pytest.raises(ValueError):
        ops.warp3d(torch.zeros(0, 2, 4, 5, 6, device=DEV), torch.zeros(1, 3, 4, 5, 6, device=DEV))


def test_warp3d_ragged_sweep_vs_oracle(ops):
    """Round 4: the forward pair runs as a ring pipeline (mover waves + LDS-DMA tiles, 16 slices per workgroup, gathers
    one slice ahead) with a 16-byte and a dword form of the tile DMA.  Seeded sweep over extents that hit every corner of
    it: fewer slices than ring stages (D = 2, 3), D not a multiple of 16, H / W below, at and just above one 64 x 32 tile,
    W % 4 != 0 (dword pieces) and W % 4 == 0 (16-byte pieces), several image channels (stages of one pipeline), sampled
    volumes whose extent differs from the flow's.  Forward vs the oracle everywhere; flow gradient on a subset."""
    rng = np.random.RandomState(20260104)
    shapes = [(1, 1, 2, 5, 8), (1, 2, 3, 64, 32), (2, 1, 17, 65, 33), (1, 1, 33, 7, 36), (1, 3, 5, 70, 12),
              (1, 1, 16, 64, 64), (2, 2, 18, 3, 4), (1, 1, 40, 129, 31)]
    for _ in range(6):
        shapes.append((int(rng.randint(1, 3)), int(rng.randint(1, 3)), int(rng.randint(2, 40)), int(rng.randint(2, 140)),
                       int(rng.randint(2, 70))))
    for n, (B, C, D, H, W) in enumerate(shapes):
        g = torch.Generator().manual_seed(n)
        mixed = n % 5 == 4
        ishape = (B, C, D + 2, H + 1, W + 3) if mixed else (B, C, D, H, W)
        x = torch.rand(ishape, generator=g)
        f = (torch.rand(B, 3, D, H, W, generator=g) * 2 - 1) * 3.0
        ref = owarps.warp3d_ref(x, f)
        out = ops.warp3d(x.to(DEV), f.to(DEV))
        assert maxerr(out, ref) < OUT_ATOL, (B, C, D, H, W, mixed)
        # the pair launch (both members in one grid) on the same data
        f6 = torch.cat((f, -f), 1)
        o0, o1 = ops.warp_pair(x.to(DEV), x.flip(0).to(DEV) if B > 1 else x.to(DEV), f6.to(DEV))
        assert torch.equal(o0, out)
        if n % 3 == 0:
            fr = f.clone().requires_grad_()
            G = torch.randn(ref.shape, generator=g)
            (gfr,) = torch.autograd.grad((owarps.warp3d_ref(x, fr) * G).sum(), [fr])
            fd = f.to(DEV).requires_grad_()
            (gf,) = torch.autograd.grad((ops.warp3d(x.to(DEV), fd) * G.to(DEV)).sum(), [fd])
            assert frac_bad(gf, gfr, GRAD_ATOL) < 2e-3, (B, C, D, H, W)  # (small volumes: a few boundary voxels weigh more)


def test_warp2d_plane_owner_grad_in_sweep(ops):
    """Round 4: grad_in of planes that fit LDS comes from plane-owning workgroups (no global atomics), larger planes keep
    the atomic kernel: both against the oracle, at channel counts that do not divide the planes-per-workgroup choice,
    masked and unmasked, and across the 48 KB switch (110 x 110 floats fit, 111 x 111 do not)."""
    cases = [(2, 5, 7, 9, True), (3, 33, 19, 57, True), (1, 7, 110, 110, False), (1, 3, 111, 111, False),
             (32, 9, 3, 8, True), (2, 1, 38, 113, False)]
    for n, (B, C, H, W, mask) in enumerate(cases):
        g = torch.Generator().manual_seed(100 + n)
        x = torch.rand(B, C, H, W, generator=g)
        f = (torch.rand(B, 2, H, W, generator=g) * 2 - 1) * 2.5
        G = torch.randn(B, C, H, W, generator=g)
        if mask:  # borderline pixels (validity decided by fp32 rounding) leave the upstream gradient on both sides
            G = G * (~owarps.pwc_mask_borderline(x, f))
        xr, fr = x.clone().requires_grad_(), f.clone().requires_grad_()
        ref = owarps.warp2d_pwc_ref(xr, fr, mask)
        gxr, gfr = torch.autograd.grad((ref * G).sum(), [xr, fr])
        xd, fd = x.to(DEV).requires_grad_(), f.to(DEV).requires_grad_()
        out = ops.warp2d_pwc(xd, fd, mask)
        gx, gf = torch.autograd.grad((out * G.to(DEV)).sum(), [xd, fd])
        if not mask:
            assert maxerr(out, ref) < OUT_ATOL
        assert frac_bad(gx, gxr, GRAD_ATOL) == 0.0, (B, C, H, W)
        assert frac_bad(gf, gfr, GRAD_ATOL) < 1e-4


def _rc_flows(B, D, H, W, g):
    """Flows that drive the row-cache kernels (csrc/warp3d_rc.hpp) through every path: inside the predicted window
    (smooth, small), a constant shift (window origin far from the tile), white noise (every voxel on the global-gather
    path), a jump along d (the row ring restarts), a ramp along d steeper than one row per slice, poisoned entries."""
    ax = lambda n: torch.linspace(0, 6.28318, n)
    d, h, w = ax(D).view(D, 1, 1), ax(H).view(1, H, 1), ax(W).view(1, 1, W)
    smooth = torch.stack([1.5 * torch.sin(d) * torch.cos(h) + 0 * w, 1.2 * torch.cos(2 * w) + 0 * d + 0 * h,
                          0.8 * torch.sin(h) * torch.sin(w) + 0 * d], 0).expand(B, 3, D, H, W).contiguous()
    shift = torch.empty(B, 3, D, H, W)
    shift[:, 0], shift[:, 1], shift[:, 2] = 9.25, -6.5, 11.75
    noise = (torch.rand(B, 3, D, H, W, generator=g) * 2 - 1) * 3.0
    jump = smooth.clone()
    jump[:, 1, D // 2:] += 7.0
    ramp = smooth.clone()
    ramp[:, 1] += 2.5 * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1)
    wild = smooth.clone()
    flat = wild.view(-1)
    idx = torch.randperm(flat.numel(), generator=g)[:48]
    flat[idx] = torch.tensor([float("nan"), float("inf"), -float("inf"), 1e30, -1e30, 3e9]).repeat(8)
    return {"smooth": smooth, "shift": shift, "noise": noise, "jump": jump, "ramp": ramp, "wild": wild}


@pytest.mark.parametrize("shape,mixed", [((1, 40, 70, 72), False), ((2, 37, 64, 96), False), ((1, 45, 130, 76), True),
                                         ((1, 3, 64, 72), False)])
def test_warp3d_row_cache_kernels_vs_oracle_and_gather_kernels(ops, shape, mixed):
    """Round 5: at C = 1 on volumes that hold the window (W_in >= 72, D_in >= 37, W_in % 4 == 0) forward and flow-gradient
    backward of the trilinear warp run as the ring pipeline with the gather source in an LDS row cache.  Against the oracle
    for every flow kind, and bit for bit against the gather kernels, which the same entry points still take at C = 2
    (forward: a duplicated channel) and when grad_in is asked for (backward)."""
    B, D, H, W = shape
    g = torch.Generator().manual_seed(B * 1000 + D * 10 + H)
    ishape = (B, 1, D + 3, H + 1, W + 4) if mixed else (B, 1, D, H, W)
    x = torch.rand(ishape, generator=g)
    G = torch.randn(B, 1, D, H, W, generator=g)
    for kind, f in _rc_flows(B, D, H, W, g).items():
        xd, fd = x.to(DEV), f.to(DEV).requires_grad_()
        out = ops.warp3d(xd, fd)
        (gf,) = torch.autograd.grad((out * G.to(DEV)).sum(), [fd])
        # the gather kernels on the same operands
        out2 = ops.warp3d(torch.cat((xd, xd), 1), fd.detach())
        xg, fg = xd.clone().requires_grad_(), f.to(DEV).requires_grad_()
        _, gf2 = torch.autograd.grad((ops.warp3d(xg, fg) * G.to(DEV)).sum(), [xg, fg])
        if kind == "wild":
            ok = torch.isfinite(f).all(dim=1, keepdim=True).to(DEV)
            assert torch.equal(out[ok], out2[:, :1][ok]), (kind, shape)
            assert torch.equal(gf.nan_to_num()[ok.expand_as(gf)], gf2.nan_to_num()[ok.expand_as(gf)]), (kind, shape)
            continue
        assert torch.equal(out, out2[:, :1]) and torch.equal(out, out2[:, 1:]), (kind, shape)
        assert torch.equal(gf, gf2), (kind, shape)
        fr = f.clone().requires_grad_()
        ref = owarps.warp3d_ref(x, fr)
        (gfr,) = torch.autograd.grad((ref * G).sum(), [fr])
        assert maxerr(out, ref) < OUT_ATOL, (kind, shape)
        assert frac_bad(gf, gfr, GRAD_ATOL) < 2e-4, (kind, shape)


def test_warp_pair_acc_row_cache_kernel_adds_three_strided_gradients(ops):
    """The three-addend backward (fs_warp3d_pair_bwd_acc3: the dominant hot-path launch of the 256^3 step) on the row-cache
    kernel: addends arrive as dense tensors and as a channel slice of a wider one; equals warp gradient + addends."""
    g = torch.Generator().manual_seed(77)
    B, D, H, W = 2, 38, 66, 80
    i0, i1 = torch.rand(B, 1, D, H, W, generator=g).to(DEV), torch.rand(B, 1, D, H, W, generator=g).to(DEV)
    f = _rc_flows(B, D, H, W, g)["smooth"]
    f6 = torch.cat((f, -0.5 * f), 1).to(DEV).requires_grad_()
    G0, G1 = torch.randn(B, 1, D, H, W, generator=g).to(DEV), torch.randn(B, 1, D, H, W, generator=g).to(DEV)
    wide = torch.randn(B, 11, D, H, W, generator=g).to(DEV)
    A = [torch.randn(B, 6, D, H, W, generator=g).to(DEV), wide[:, 5:11], torch.randn(B, 6, D, H, W, generator=g).to(DEV)]
    w0, w1, (fa, fb, fc) = ops.warp_pair_acc(i0, i1, f6)
    (gt,) = torch.autograd.grad([w0, w1, fa, fb, fc], [f6], [G0, G1] + A)
    f6b = f6.detach().clone().requires_grad_()
    o0, o1 = ops.warp_pair(i0, i1, f6b)
    (gw,) = torch.autograd.grad([o0, o1], [f6b], [G0, G1])
    assert torch.equal(w0, o0) and torch.equal(w1, o1)
    exp = ((gw + A[0]) + A[1]) + A[2]  # (the kernel adds in the order autograd delivers the three gradients)
    assert float((gt - exp).abs().max()) < 1e-5 * max(1.0, float(exp.abs().max()))
