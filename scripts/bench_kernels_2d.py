"""Per-kernel timing of the 2-D hot-path rows at the BASELINE C2 / C3 shapes (GPU box only):
algorithmic GB/s of corr2d, census, robust reductions and the 2-D warps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opticalflowscivis_amd import ops


def t(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def row(name, ms, nbytes, flops=0):
    extra = "  %.1f TFLOP/s" % (flops / ms / 1e9) if flops else ""
    print("%-46s %8.4f ms  %7.1f GB/s algorithmic%s" % (name, ms, nbytes / ms / 1e6, extra), flush=True)


dev = "cuda:0"
B = 32
# a3/a4: UPFlow pyramid levels at C3 (150x450 input)
for C, h, w in [(196, 3, 8), (128, 5, 15), (96, 10, 29), (64, 19, 57), (32, 38, 113)]:
    f1 = torch.randn(B, C, h, w, device=dev, requires_grad=True)
    f2 = torch.randn(B, C, h, w, device=dev, requires_grad=True)
    out = ops.corr2d(f1, f2, 4)
    G = torch.randn_like(out)
    fb = 4 * (2 * C + 81) * h * w * B
    row("corr2d fwd  C=%3d %3dx%3d B=32" % (C, h, w), t(lambda: ops.corr2d(f1, f2, 4)), fb, 2 * 81 * C * h * w * B)
    row("corr2d bwd  C=%3d %3dx%3d B=32" % (C, h, w),
        t(lambda: torch.autograd.grad(out, [f1, f2], G, retain_graph=True)), 4 * (4 * C + 81) * h * w * B,
        4 * 81 * C * h * w * B)
# a8: census on the C3 image pair
im1 = torch.rand(B, 3, 150, 450, device=dev, requires_grad=True)
im2 = torch.rand(B, 3, 150, 450, device=dev, requires_grad=True)
occ = (torch.rand(B, 1, 150, 450, device=dev) > 0.3).float()
npx = B * 150 * 450
d = ops.census_dist(im1, im2)
Gd = torch.randn_like(d)
row("census_dist fwd 32x3x150x450", t(lambda: ops.census_dist(im1, im2)), 28 * npx)
row("census_dist bwd", t(lambda: torch.autograd.grad(d, [im1, im2], Gd, retain_graph=True)), (28 + 24) * npx)
row("census loss fwd (dist + robust_sum, occ)", t(lambda: ops.census_loss(im1, im2, occ, 0.4, False, True)), 36 * npx)
# a9: photometric loss
row("photo_loss abs_robust fwd (occ)", t(lambda: ops.photo_loss_multi_type(im1, im2, occ, 'abs_robust', 0.4, True)),
    (24 + 4) * npx)
# a5/a6/a7: feature / image warps
flow = torch.randn(B, 2, 150, 450, device=dev) * 2
row("warp2d dilated fwd 32x3x150x450", t(lambda: ops.warp2d_dilated(im1.detach(), flow)), (8 + 12 + 12) * npx)
f = torch.randn(B, 32, 38, 113, device=dev)
fl = torch.randn(B, 2, 38, 113, device=dev)
row("warp2d pwc+mask fwd 32x32x38x113", t(lambda: ops.warp2d_pwc(f, fl, True)), (8 + 8 * 32) * B * 38 * 113)
# a1: Flow-2D pair warp at C2
i0, i1 = torch.rand(16, 1, 160, 224, device=dev), torch.rand(16, 1, 160, 224, device=dev)
f4 = torch.randn(16, 4, 160, 224, device=dev)
row("warp2d pair fwd 16x1x160x224 (C2)", t(lambda: ops.warp_pair(i0, i1, f4)), 2 * 16 * 16 * 160 * 224)
# §8f.2: occlusion check (one launch) vs the stock-op sequence it replaces
ff = 2.0 * torch.randn(B, 2, 1, 1, device=dev) + torch.randn(B, 2, 150, 450, device=dev)
fbk = -ff + 0.5 * torch.randn(B, 2, 150, 450, device=dev)
row("occ_check2d 'obj' 32x2x150x450 (fused)", t(lambda: ops.occ_check2d(ff, fbk, 0.1, 0.5, 1, "obj")), 24 * npx)
from opticalflowscivis_amd.upflow.utils.tools import tools as _tools
_m = _tools.occ_check_model(occ_alpha_1=0.1, occ_alpha_2=0.5, obj_out_all="obj")


def _occ_stock():
    o1, o2 = _m._forward_backward_occ_check(ff, fbk, 1)
    return (_m.torch_get_obj_occ_check(o1, _m.torch_outgoing_occ_check(ff)),
            _m.torch_get_obj_occ_check(o2, _m.torch_outgoing_occ_check(fbk)))


row("occ_check 'obj' stock ops + HIP warps", t(_occ_stock), 24 * npx)
# §8f.3: Laplacian-pyramid loss at C2 (fwd + bwd), fused vs the stock-op pyramid
import torch.nn.functional as _F


def _stock_gauss(img, kernel):
    return _F.conv2d(_F.pad(img, (2, 2, 2, 2), mode='reflect'), kernel, groups=img.shape[1])


def _stock_pyramid(img, levels):
    """The stock-op Laplacian pyramid (the comparison arm only; the product has no such path)."""
    k1 = torch.tensor([1., 4., 6., 4., 1.], device=img.device)
    kernel = (k1[:, None] * k1[None, :] / 256.).repeat(img.shape[1], 1, 1, 1)
    cur, pyr = img, []
    for _ in range(levels):
        down = _stock_gauss(cur, kernel)[:, :, ::2, ::2]
        up = cur.new_zeros(down.shape[0], down.shape[1], 2 * down.shape[2], 2 * down.shape[3])
        up[:, :, ::2, ::2] = down
        up = _stock_gauss(up, 4 * kernel)
        h, w = min(cur.shape[2], up.shape[2]), min(cur.shape[3], up.shape[3])
        pyr.append(cur[:, :, :h, :w] - up[:, :, :h, :w])
        cur = down
    return pyr


a = torch.rand(16, 1, 160, 224, device=dev, requires_grad=True)
b = torch.rand(16, 1, 160, 224, device=dev)


def _lap_hip():
    (g,) = torch.autograd.grad(ops.laploss2d(a, b, 5), [a])
    return g


def _lap_stock():
    pa, pb = _stock_pyramid(a, 5), _stock_pyramid(b, 5)
    loss = sum(torch.nn.functional.l1_loss(x, y) for x, y in zip(pa, pb))
    (g,) = torch.autograd.grad(loss, [a])
    return g


row("laploss2d fwd+bwd 16x1x160x224 (fused)", t(_lap_hip), 16 * 16 * 160 * 224)
row("LapLoss fwd+bwd stock ops", t(_lap_stock, n=20), 16 * 16 * 160 * 224)
# §8f.4: normalised cost volume at the finest UPFlow level, fused vs normalize_features + corr2d
from opticalflowscivis_amd.upflow.model.upflow import network_tools as _nt
f1 = torch.randn(B, 32, 38, 113, device=dev, requires_grad=True)
f2 = torch.randn(B, 32, 38, 113, device=dev, requires_grad=True)
Gc = torch.randn(B, 81, 38, 113, device=dev)


def _cn_hip():
    return torch.autograd.grad(ops.corr2d_normalized(f1, f2, 4), [f1, f2], Gc)


def _cn_stock():
    n1, n2 = _nt.normalize_features((f1, f2), True, True, False, False)
    return torch.autograd.grad(ops.corr2d(n1, n2, 4), [f1, f2], Gc)


row("corr2d_normalized fwd+bwd C=32 38x113 (fused)", t(_cn_hip), 4 * (6 * 32 + 2 * 81) * 38 * 113 * B)
row("normalize_features + corr2d fwd+bwd (stock)", t(_cn_stock), 4 * (6 * 32 + 2 * 81) * 38 * 113 * B)
