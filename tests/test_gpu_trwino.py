"""-m gpu: the Winograd F(4,2) form of the transposed convolution (csrc/convtrwino.hpp: 17..32 output channels on rows of
64 input positions -- the 64 -> 32 layers between the 64^3 trunk and the 128^3 grid) against fp64: plain, with bias, with
the fused PReLU second output, with the residual addend; channel counts below the 32-row tile,
ragged z / y extents, two x bricks; the last output column comes from the edge kernel.  And through the autograd node
of the layer (forward of a ConvTranspose3d, input gradient of a Conv3d)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 3e-5


@pytest.fixture(scope="module")
def ops():
    from opticalflowscivis_amd import ops as o
    return o


def _kind(B, cin, cout, size, has_z=0):
    from opticalflowscivis_amd import _lib
    L = _lib.lib()
    buf = (_lib.FsWprepJob * 8)()
    out = tuple(2 * n for n in size)
    n = L.fs_conv3d_tr_wprep_jobs(buf, 8, 0x4000, 0x1000, 0x2000, B, cin, cout, *size, *out, has_z)
    return [buf[i].kind for i in range(n)]


@pytest.mark.parametrize("B,cin,cout,size", [(2, 64, 32, (32, 16, 64)), (2, 16, 32, (19, 17, 64)), (2, 12, 20, (32, 16, 64)),
                                             (2, 8, 24, (16, 16, 128))])
def test_trwino_forward_vs_fp64(ops, B, cin, cout, size):
    assert _kind(B, cin, cout, size) == [7]
    g = torch.Generator().manual_seed(cin * 10 + size[1])
    x = torch.randn((B, cin) + size, generator=g)
    w = torch.randn(cin, cout, 4, 4, 4, generator=g) / (cin * 8) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv_transpose3d(x.double(), w.double(), b.double(), 2, 1)
    scale = float(ref.abs().max())
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    got = ops.conv3d_tr(xd, wd, bd)
    err = (got.cpu().double() - ref).abs()
    assert float(err.max()) < TOL * scale, "max error at %s" % (torch.nonzero(err == err.max())[0].tolist(),)
    got = ops.conv3d_tr(xd, wd, None)
    assert float((got.cpu().double() - (ref - b.double().view(1, -1, 1, 1, 1))).abs().max()) < TOL * scale
    # fused PReLU second output (per-channel and shared slope)
    for slope in (torch.rand(cout, generator=g) - 0.3, torch.tensor([0.2])):
        y, z = ops.conv3d_tr(xd, wd, bd, None, slope.to(DEV))
        assert float((y.cpu().double() - ref).abs().max()) < TOL * scale
        assert float((z.cpu().double() - F.prelu(ref, slope.double())).abs().max()) < TOL * scale
    # residual addend
    add = torch.randn(ref.shape, generator=g)
    got = ops.conv3d_tr(xd, wd, bd, None, None, add.to(DEV))
    assert float((got.cpu().double() - (ref + add.double())).abs().max()) < TOL * max(scale, float((ref + add.double()).abs().max()))


def test_trwino_is_the_input_gradient_of_the_strided_convolution(ops):
    """Conv3d(4, 2, 1) 32 -> 64 at 64 x 32 x 128 -> 32 x 16 x 64: its input gradient is this kernel with the layer's
    weight [64][32][64] read as [in = 64][out = 32]."""
    from opticalflowscivis_amd import convgrad
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 32, 64, 32, 128, generator=g)
    w = torch.randn(64, 32, 4, 4, 4, generator=g) / (32 * 64) ** 0.5
    gy = torch.randn(2, 64, 32, 16, 64, generator=g)
    assert _kind(2, 64, 32, (32, 16, 64)) == [7]
    xr = x.double().requires_grad_()
    (F.conv3d(xr, w.double(), None, 2, 1) * gy.double()).sum().backward()
    xd = x.to(DEV).requires_grad_()
    y = convgrad._ConvFn.apply(xd, w.to(DEV), None, (2, 2, 2), (1, 1, 1), False)
    gx, = torch.autograd.grad((y * gy.to(DEV)).sum(), [xd])
    assert float((gx.cpu().double() - xr.grad).abs().max()) < TOL * float(xr.grad.abs().max())


def test_trwino_is_not_taken_where_it_does_not_apply():
    assert _kind(2, 64, 32, (8, 8, 64)) == [1]      # 90 bricks: the class kernel's slab
    assert _kind(2, 64, 32, (32, 32, 32)) == [1]    # rows of 32
    assert _kind(2, 32, 12, (32, 16, 64)) == [2]    # <= 16 output channels: the 16-row kernel
    assert _kind(2, 128, 64, (32, 16, 64)) == [1, 1]  # two 32-channel slices
