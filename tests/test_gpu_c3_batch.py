"""-m gpu: BACKWARD at BASELINE config C3's full batch (UPFlow, B = 32, 3 x 150 x 450, census on) -- VERDICT r2
"missing" #6.  The B = 32 launch geometry of the backward kernels is checked (a) op by op against the CPU oracle at
the shapes the C3 step launches them with (UPFlow/model/upflow.py:632-652, 508-530; pyramid shapes SURVEY
Appendix B) and (b) end to end: one `backward()` of the B = 32 loss must deliver, per sample and per parameter,
what the B = 2 slices of the batch deliver (every loss term is a batch mean; samples are independent)."""
import numpy as np
import pytest
import torch

from oracle import corr as ocorr
from oracle import losses as olosses
from oracle import warps as owarps

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from opticalflowscivis_amd import ops as o
    return o


@pytest.mark.parametrize("shape", [(32, 32, 38, 113), (32, 64, 19, 57), (32, 96, 10, 29)])
def test_corr2d_pair_backward_at_the_c3_batch(ops, shape):
    """fs_corr2d_pair_{fwd,bwd}: both directions of a pyramid level in one launch, B = 32, vs the oracle's
    shift-by-shift definition (pinned to Corr_pyTorch by tests/test_oracle_golden.py)."""
    g = torch.Generator().manual_seed(shape[1])
    fs = [torch.randn(shape, generator=g) for _ in range(4)]
    Ga, Gb = torch.randn(shape[0], 81, shape[2], shape[3], generator=g), \
        torch.randn(shape[0], 81, shape[2], shape[3], generator=g)
    cpu = [t.clone().requires_grad_() for t in fs]
    ra, rb = ocorr.corr2d_closed(cpu[0], cpu[1], 4), ocorr.corr2d_closed(cpu[2], cpu[3], 4)
    rg = torch.autograd.grad((ra * Ga).sum() + (rb * Gb).sum(), cpu)
    dev = [t.to(DEV).requires_grad_() for t in fs]
    oa, ob = ops.corr2d_pair(dev[0], dev[1], dev[2], dev[3], 4)
    gg = torch.autograd.grad((oa * Ga.to(DEV)).sum() + (ob * Gb.to(DEV)).sum(), dev)
    assert float((oa.detach().cpu() - ra.detach()).abs().max()) < 1e-5
    assert float((ob.detach().cpu() - rb.detach()).abs().max()) < 1e-5
    for a, b in zip(gg, rg):
        assert float((a.cpu() - b).abs().max()) < 5e-5


def test_corr2d_normalized_pair_backward_at_the_c3_batch(ops):
    """The C3 configuration proper: per-plane normalisation folded into the cost volume (fs_corr2d_pair_* with
    moments, fs_plane_norm_bwd4), finest level, B = 32."""
    shape = (32, 32, 38, 113)
    g = torch.Generator().manual_seed(5)
    fs = [1.3 * torch.randn(shape, generator=g) + 0.4 for _ in range(4)]
    Ga, Gb = torch.randn(32, 81, 38, 113, generator=g), torch.randn(32, 81, 38, 113, generator=g)
    cpu = [t.clone().requires_grad_() for t in fs]
    ra, rb = ocorr.corr2d_normalized_ref(cpu[0], cpu[1], 4), ocorr.corr2d_normalized_ref(cpu[2], cpu[3], 4)
    rg = torch.autograd.grad((ra * Ga).sum() + (rb * Gb).sum(), cpu)
    dev = [t.to(DEV).requires_grad_() for t in fs]
    oa, ob = ops.corr2d_pair(dev[0], dev[1], dev[2], dev[3], 4, normalize=True)
    gg = torch.autograd.grad((oa * Ga.to(DEV)).sum() + (ob * Gb.to(DEV)).sum(), dev)
    assert float((oa.detach().cpu() - ra.detach()).abs().max()) < 2e-5
    for a, b in zip(gg, rg):
        assert float((a.cpu() - b).abs().max()) < 1e-4 * max(1.0, float(b.abs().max()))


def test_census_backward_at_the_c3_batch(ops):
    """fs_census_dist_{fwd,bwd} + the masked robust reduction on [32, 3, 150, 450] (upflow.py:527-530)."""
    g = torch.Generator().manual_seed(19)
    im1 = torch.rand(32, 3, 150, 450, generator=g)
    im2 = (im1 + 0.05 * torch.randn(32, 3, 150, 450, generator=g)).clamp(0, 1)
    occ = (torch.rand(32, 1, 150, 450, generator=g) > 0.2).float()
    for use_occ in (False, True):
        a, b = im1.clone().requires_grad_(), im2.clone().requires_grad_()
        ref = olosses.census_loss(a, b, occ, 0.4, False, use_occ)
        r1, r2 = torch.autograd.grad(ref, [a, b])
        c, d = im1.to(DEV).requires_grad_(), im2.to(DEV).requires_grad_()
        loss = ops.census_loss(c, d, occ.to(DEV), 0.4, False, use_occ)
        g1, g2 = torch.autograd.grad(loss, [c, d])
        assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))
        for x, y in ((g1, r1), (g2, r2)):
            assert float((x.cpu() - y).abs().max()) < 2e-4 * float(y.abs().max())


def test_warp_backwards_at_the_c3_batch(ops):
    """The masked feature warp at the finest level [32, 32, 38, 113] (upflow.py:632-633) and the boundary-dilated
    image warp on [32, 3, 150, 450] (upflow.py:508-509), forward and both gradients."""
    g = torch.Generator().manual_seed(23)
    x = torch.rand(32, 32, 38, 113, generator=g)
    f = 2.5 * torch.randn(32, 2, 1, 1, generator=g) + 0.7 * torch.randn(32, 2, 38, 113, generator=g)
    G = torch.randn(x.shape, generator=g)
    sure = ~owarps.pwc_mask_borderline(x, f)
    Gs = G * sure  # the validity mask has no derivative; fp32-borderline pixels are taken out on both sides
    xo, fo = x.clone().requires_grad_(), f.clone().requires_grad_()
    ro = owarps.warp2d_pwc_ref(xo, fo, True)
    rgx, rgf = torch.autograd.grad((ro * Gs).sum(), [xo, fo])
    xd, fd = x.to(DEV).requires_grad_(), f.to(DEV).requires_grad_()
    out = ops.warp2d_pwc(xd, fd, with_mask=True)
    gx, gf = torch.autograd.grad((out * Gs.to(DEV)).sum(), [xd, fd])
    assert float(((out.detach().cpu() - ro.detach()).abs() * sure).max()) < 2e-5
    assert float((gx.cpu() - rgx).abs().max()) < 2e-4 and float((gf.cpu() - rgf).abs().max()) < 2e-4 * 32
    I = torch.rand(32, 3, 150, 450, generator=g)
    f = 2.0 * torch.randn(32, 2, 1, 1, generator=g) + 0.5 * torch.randn(32, 2, 150, 450, generator=g)
    G = torch.randn(I.shape, generator=g)
    Io, fo = I.clone().requires_grad_(), f.clone().requires_grad_()
    ro = owarps.warp2d_dilated_ref(Io, fo, torch.zeros(32, 2, 1, 1))
    rgI, rgf = torch.autograd.grad((ro * G).sum(), [Io, fo])
    Id, fd = I.to(DEV).requires_grad_(), f.to(DEV).requires_grad_()
    out = ops.warp2d_dilated(Id, fd, torch.zeros(32, 2, 1, 1, device=DEV))
    gI, gf = torch.autograd.grad((out * G.to(DEV)).sum(), [Id, fd])
    assert float((out.detach().cpu() - ro.detach()).abs().max()) < 2e-5
    assert float((gI.cpu() - rgI).abs().max()) < 2e-4
    bad = ((gf.cpu() - rgf).abs() > 2e-4).float().mean()  # a coordinate within fp32 noise of a cell boundary
    assert float(bad) < 1e-5


def _c3_net():
    from opticalflowscivis_amd.upflow.model.upflow import UPFlow_net
    conf = UPFlow_net.config()
    conf.update({'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
                 'norm_moments_across_images': False, 'photo_loss_census_weight': 1,
                 'multi_scale_distillation_weight': 1})
    torch.manual_seed(0)
    return conf().to(DEV)


KEYS = ['photo_loss', 'smooth_loss', 'census_loss', 'msd_loss']


def test_upflow_c3_b32_backward_equals_its_b2_slices():
    """One backward() of the C3 step at B = 32 against the 16 B = 2 slices of the same batch: every loss term is a
    mean over the batch, so (i) each parameter's gradient is the mean of the slices' gradients and (ii) the gradient
    w.r.t. the input frames of sample i is 1/16 of what its slice delivers.  The comparison is a band, like the
    forward's (test_gpu_e2e.py::test_upflow_c3_b32_equals_its_b2_slices): MIOpen picks batch-dependent algorithms
    for the stock 2-D convolutions and the validity masks of WarpingLayer_no_div flip on fp32 noise; a B = 32 launch
    geometry error in any HIP backward kernel (a sample dropped, a stride wrong) is an O(1) difference in these
    numbers, not a few per cent."""
    from opticalflowscivis_amd.data import synthetic
    net = _c3_net()
    pairs = synthetic.vortex2d_pairs(32, 150, 450, seed=0, device=DEV)
    im1, im2 = pairs[:, 0].contiguous().requires_grad_(), pairs[:, 1].contiguous().requires_grad_()
    params = [p for p in net.parameters() if p.requires_grad]
    out = net({'im1': im1, 'im2': im2, 'if_loss': True})
    loss = sum(out['loss_dict'][k] for k in KEYS)
    grads = torch.autograd.grad(loss, [im1, im2] + params, allow_unused=True)
    g_im1, g_im2, g32 = grads[0], grads[1], grads[2:]
    assert all(bool(torch.isfinite(g).all()) for g in grads if g is not None)
    acc = [torch.zeros_like(p, dtype=torch.float64) for p in params]
    in_err, in_den = 0.0, 0.0
    worst_sample = 0.0
    for i in range(0, 32, 2):
        a, b = im1.detach()[i:i + 2].clone().requires_grad_(), im2.detach()[i:i + 2].clone().requires_grad_()
        o = net({'im1': a, 'im2': b, 'if_loss': True})
        gs = torch.autograd.grad(sum(o['loss_dict'][k] for k in KEYS), [a, b] + params, allow_unused=True)
        for j, gj in enumerate(gs[2:]):
            if gj is not None:
                acc[j] += gj.double()
        for g_full, g_sl in ((g_im1, gs[0]), (g_im2, gs[1])):
            for s in range(2):
                e = float((g_full[i + s].double() * 16 - g_sl[s].double()).abs().sum())
                d = float(g_sl[s].double().abs().sum())
                in_err, in_den = in_err + e, in_den + d
                worst_sample = max(worst_sample, e / max(d, 1e-30))
    rel = []
    for g, a in zip(g32, acc):
        if g is None:
            continue
        a = a / 16
        rel.append(float((g.double() - a).norm()) / max(float(a.norm()), 1e-30))
    rel = np.array(rel)
    print("C3 B=32 backward vs its B=2 slices: input gradients L1 error %.3e (worst sample %.3e); parameter "
          "gradients relative L2 error median %.3e max %.3e" % (in_err / in_den, worst_sample, np.median(rel),
                                                                rel.max()))
    # measured: parameter gradients median 1.1 % / max 2.6 %; per-pixel input gradients (they pass through the census
    # term and every flipped validity-mask pixel) 13 % in L1, worst sample 30 %.  A dropped or swapped sample would be
    # >= 100 % on that sample and >= 1/32 ... 1/16 of every parameter gradient's norm at once.
    assert in_err / in_den < 0.30 and worst_sample < 0.6
    assert np.median(rel) < 0.03 and rel.max() < 0.08
