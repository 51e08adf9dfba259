"""-m gpu: the Winograd forms of the 64-channel k3 trunk convolutions against fp64 -- forward / input gradient: the 2-D
F(2,3) x F(4,3) kernel (csrc/convwino2d.hpp) that the dispatch takes, and in a subprocess the 1-D F(4,3) / F(2,3) kernels
it superseded (csrc/convwino4.hpp, convwino.hpp: reached with FLOWSCI_FWD_NO_WINO2D / _NO_WINO4); weight gradient: F(4,3)
(csrc/convwrwwino4.hpp).  Every fused epilogue (bias, PReLU output + residual addend, plain addend, the PReLU-backward
form), both weight modes (forward taps / flipped + transposed for the input gradient), volume edges that are not
multiples of the brick, channel counts below the 64-row tile.  The coefficients reach 8 and 1/24: the error against fp64
stays within 2x the direct kernel's bound (3e-5 of the output's magnitude)."""
import os
import subprocess
import sys
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 6e-5


@pytest.fixture(scope="module")
def ops():
    from opticalflowscivis_amd import ops as o
    return o


def _slab_kind(ops, x, w, cin, cout, size, wmode):
    """The library's own dispatch: which filter slab does this call take (FsWprepJob kind: 0 direct taps, 4 the F(2,3)
    transform, 5 the F(4,3) transform, 6 the 2-D F(2,3) x F(4,3) transform)?"""
    from opticalflowscivis_amd import _lib
    L = _lib.lib()
    buf = (_lib.FsWprepJob * 4)()
    ws = torch.empty(int(L.fs_conv3d_fwd_ws_floats(cin, cout, 3)), device=DEV)
    n = L.fs_conv3d_fwd_wprep_jobs(buf, 4, x.data_ptr(), w.data_ptr(), ws.data_ptr(), x.shape[0], cin, cout, *size, *size,
                                   3, 1, 1, wmode)
    assert n == 1
    return buf[0].kind


def _expected_kind(B, size):
    """The 2-D form wherever its bricks (2 x 2 x 64, or 2 x 4 x 32 on rows of 32) give every CU one (the 1-D forms it
    superseded need 512 bricks of the same or twice the size: they are reached only with the 2-D form switched off, see
    the subprocess test below)."""
    D, H, W = size
    if W % 32:
        return 0
    yt = 1 if W % 64 == 0 else 2
    xb = W // 64 if W % 64 == 0 else W // 32
    return 6 if B * ((D + 1) // 2) * ((H + 2 * yt - 1) // (2 * yt)) * xb >= 256 else 0


def _is_wino(ops, x, w, cin, cout, size, wmode):
    return _slab_kind(ops, x, w, cin, cout, size, wmode) in (4, 5, 6)


@pytest.mark.parametrize("B,cin,cout,size", [(2, 64, 64, (32, 32, 64)), (2, 64, 64, (33, 31, 64)), (2, 8, 20, (32, 32, 64)),
                                             (2, 64, 64, (16, 32, 128)), (2, 12, 64, (64, 64, 64)),
                                             # F(4,3) bricks (4 x 2 x 64): whole, ragged in z and y, few channels, two x bricks
                                             (2, 64, 64, (64, 32, 64)), (2, 64, 64, (62, 33, 64)), (2, 8, 20, (64, 32, 64)),
                                             (2, 64, 64, (32, 32, 128)),
                                             # the 2-D form's 2 x 2 x 64 bricks: ragged in z and y, few channels, two x bricks
                                             (2, 64, 64, (63, 33, 64)), (2, 12, 24, (64, 32, 64)), (2, 64, 64, (66, 34, 64)),
                                             # rows of 32 (2 x 4 x 32 bricks): whole, ragged, three x bricks
                                             (2, 64, 64, (32, 32, 32)), (2, 64, 48, (31, 30, 32)), (2, 16, 64, (32, 18, 96)),
                                             # persistent workgroups (round 4): runs of 4 and 12 bricks in the patch order (one / three
                                             # z-steps per XCD), a run of 5 in the linear order, more bricks than an even split
                                             (1, 16, 64, (64, 64, 64)), (3, 8, 64, (64, 64, 64)), (2, 8, 64, (40, 64, 64)),
                                             (1, 8, 64, (38, 66, 64))])
def test_wino_forward_epilogues_vs_fp64(ops, B, cin, cout, size):
    g = torch.Generator().manual_seed(cin * 100 + size[0])
    x = torch.randn((B, cin) + size, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (cin * 27) ** 0.5
    b = torch.randn(cout, generator=g)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    assert _slab_kind(ops, xd, wd, cin, cout, size, 0) == _expected_kind(B, size)
    ref = F.conv3d(x.double(), w.double(), b.double(), 1, 1)
    scale = float(ref.abs().max())
    got = ops.conv3d_fwd(xd, wd, bd, 3, 1, 1, 0)
    assert float((got.cpu().double() - ref).abs().max()) < TOL * scale
    # no bias
    got = ops.conv3d_fwd(xd, wd, None, 3, 1, 1, 0)
    assert float((got.cpu().double() - (ref - b.double().view(1, -1, 1, 1, 1))).abs().max()) < TOL * scale
    # PReLU output + residual addend (conv2 of a residual unit), per-channel and shared slope
    add = torch.randn(ref.shape, generator=g)
    for slope in (torch.rand(cout, generator=g) - 0.3, torch.tensor([0.2])):
        y, z = ops.conv3d_fwd(xd, wd, bd, 3, 1, 1, 0, prelu_weight=slope.to(DEV), addend=add.to(DEV))
        zr = F.prelu(ref, slope.double()) + add.double()
        assert float((y.cpu().double() - ref).abs().max()) < TOL * scale
        assert float((z.cpu().double() - zr).abs().max()) < TOL * max(scale, float(zr.abs().max()))
    # plain addend (the skip gradient of conv1's input gradient)
    got = ops.conv3d_fwd(xd, wd, bd, 3, 1, 1, 0, addend=add.to(DEV))
    assert float((got.cpu().double() - (ref + add.double())).abs().max()) < TOL * max(scale, float((ref + add.double()).abs().max()))


@pytest.mark.parametrize("B,cg,cx,size", [(2, 64, 64, (32, 32, 64)), (2, 64, 64, (33, 34, 64)), (2, 64, 32, (32, 32, 64)),
                                          (2, 64, 64, (64, 32, 64)), (2, 64, 64, (61, 35, 64)), (2, 64, 32, (64, 32, 64)),
                                          (2, 64, 64, (48, 32, 64)), (2, 64, 64, (32, 34, 32))])
def test_wino_input_gradient_and_prelu_backward_vs_fp64(ops, B, cg, cx, size):
    """wmode 1: the layer weight [Cout_layer = cg][Cin_layer = cx] read flipped + transposed; and the same convolution
    with the PReLU backward as its epilogue (fs_conv3d_fwd_dprelu, kernel 3)."""
    g = torch.Generator().manual_seed(cg + cx + size[1])
    gy = torch.randn((B, cg) + size, generator=g)
    w = torch.randn(cg, cx, 3, 3, 3, generator=g) / (cg * 27) ** 0.5
    gyd, wd = gy.to(DEV), w.to(DEV)
    assert _slab_kind(ops, gyd, wd, cg, cx, size, 1) == _expected_kind(B, size)
    ref = F.conv3d(gy.double(), w.transpose(0, 1).flip(2, 3, 4).double(), None, 1, 1)
    scale = float(ref.abs().max())
    got = ops.conv3d_fwd(gyd, wd, None, 3, 1, 1, 1)
    assert float((got.cpu().double() - ref).abs().max()) < TOL * scale
    act = torch.randn(ref.shape, generator=g)
    for slope in (torch.rand(cx, generator=g) - 0.3, torch.tensor([0.25])):
        fused = ops.conv3d_k3_grad_input_dprelu(gyd, wd, act.to(DEV), slope.to(DEV))
        assert fused is not None
        sl = slope.double().view(1, -1, 1, 1, 1) if slope.numel() > 1 else slope.double()
        neg = act.double() <= 0
        gx = torch.where(neg, sl * ref, ref)
        ga = torch.where(neg, act.double() * ref, torch.zeros_like(ref))
        ga = ga.sum((0, 2, 3, 4)) if slope.numel() > 1 else ga.sum().view(1)
        gb = gx.sum((0, 2, 3, 4))
        assert torch.equal(fused[0], torch.where(act.to(DEV) > 0, got, slope.to(DEV).view(1, -1, 1, 1, 1) * got)
                           if slope.numel() > 1 else torch.where(act.to(DEV) > 0, got, slope.to(DEV) * got))
        assert float((fused[0].cpu().double() - gx).abs().max()) < TOL * scale
        n = float(ref.numel() / cx) ** 0.5
        assert float((fused[1].cpu().double() - ga).abs().max()) < 2e-5 * max(1.0, float(ga.abs().max())) + 1e-5 * n * scale
        assert float((fused[2].cpu().double() - gb).abs().max()) < 2e-5 * max(1.0, float(gb.abs().max())) + 1e-5 * n * scale


@pytest.mark.parametrize("env,kind", [({"FLOWSCI_FWD_NO_WINO2D": "1"}, 5), ({"FLOWSCI_FWD_NO_WINO4": "1"}, 4)])
def test_superseded_1d_kernels_in_the_ablation_build(env, kind, ablation_lib):
    """The product library holds ONE kernel per job and reads no environment variable; the 1-D forms it superseded --
    F(4,3) (kind 5) and F(2,3) (kind 4) along x only -- live in the ablation build (`make ablation`), whose dispatch
    switches are read once per process: same checks there, in a fresh process."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ab = ablation_lib
    # the product library itself must not react to the switch: same slab kind (6) with it set
    r0 = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "wino_check.py"), "6"],
                        env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r0.returncode == 0 and "kind=6" in r0.stdout, r0.stdout[-2000:] + r0.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "wino_check.py"), str(kind)],
                       env=dict(os.environ, FLOWSCI_HIP_LIBRARY=ab, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ("kind=%d" % kind) in r.stdout and "OK" in r.stdout


def test_persistent_trunk_kernel_is_bit_identical_to_the_round3_kernel(ablation_lib):
    """Round 4 rebuilt the 2-D Winograd trunk kernel (persistent workgroups, hand-counted loader waits, packed transforms,
    register-level Ay^T) with the SAME operations in the same order: every fused form it is launched in -- plain, PReLU,
    PReLU + residual, input gradient, input gradient + addend, PReLU-backward epilogue; 64^3 and 32^3 bricks -- must give
    the CRC-32 of the round-3 kernel, which the ablation build keeps (FLOWSCI_WINO2D_R3=1).  Two fresh processes
    (scripts/wino2d_ab.py); the PReLU-backward form's two gradient VECTORS are sums in another order: 1e-5 relative."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ab = ablation_lib
    script = os.path.join(root, "scripts", "wino2d_ab.py")
    env0 = {k: v for k, v in os.environ.items() if not k.startswith("FLOWSCI_")}
    r_new = subprocess.run([sys.executable, script], env=env0, capture_output=True, text=True, timeout=600)
    r_old = subprocess.run([sys.executable, script], env=dict(env0, FLOWSCI_HIP_LIBRARY=ab, FLOWSCI_WINO2D_R3="1"),
                           capture_output=True, text=True, timeout=600)
    assert r_new.returncode == 0 and r_old.returncode == 0, r_new.stderr[-2000:] + r_old.stderr[-2000:]

    def rows(out):
        d = {}
        for line in out.splitlines():
            f = line.split()
            if len(f) >= 5 and f[0].endswith("^3") and f[3] == "ms":
                crcs = [x for x in f[4:] if len(x) == 8 and all(c in "0123456789abcdef" for c in x)]
                vals = [float(f[i + 1]) for i, x in enumerate(f) if x in ("ga", "gb")]
                d[(f[0], f[1])] = (crcs, vals)
        return d

    a, b = rows(r_new.stdout), rows(r_old.stdout)
    assert len(a) == 16 and a.keys() == b.keys(), (sorted(a), sorted(b))
    for k in a:
        assert a[k][0] == b[k][0] and len(a[k][0]) >= 1, (k, a[k], b[k])
        for x, y in zip(a[k][1], b[k][1]):
            assert abs(x - y) <= 1e-5 * max(1.0, abs(y)), (k, x, y)


def test_wino_is_not_taken_where_it_does_not_apply(ops):
    x = torch.randn(2, 64, 64, 64, 16, device=DEV)   # rows of 16
    w = torch.randn(64, 64, 3, 3, 3, device=DEV)
    assert not _is_wino(ops, x, w, 64, 64, (64, 64, 16), 0)
    x = torch.randn(1, 64, 8, 8, 64, device=DEV)     # too few bricks to fill the chip
    assert not _is_wino(ops, x, w, 64, 64, (8, 8, 64), 0)
    w = torch.randn(128, 64, 3, 3, 3, device=DEV)    # more than one 64-channel group
    x = torch.randn(2, 64, 32, 32, 64, device=DEV)
    assert not _is_wino(ops, x, w, 64, 128, (32, 32, 64), 0)


@pytest.mark.parametrize("B,size", [(2, (32, 32, 64)), (2, (33, 32, 64)), (1, (64, 64, 64)), (2, (16, 32, 128))])
def test_wino_weight_gradient_vs_fp64(ops, B, size):
    """fs_conv3d_wrw on the 64 -> 64 k3 layers in the Winograd domain (csrc/convwrwwino.hpp: both operands transformed
    as they are read from LDS, G^T applied in the atomic epilogue) against the fp64 weight gradient, and the whole
    layer through the autograd node (forward, input gradient and weight gradient all on the F(2,3) kernels)."""
    from opticalflowscivis_amd import convgrad
    assert ops.conv3d_wrw_takes_winograd(B, 64, 64, size, size, 3, 1, 1, 0, 0)
    g = torch.Generator().manual_seed(size[0] + B)
    x = torch.randn((B, 64) + size, generator=g)
    w = torch.randn(64, 64, 3, 3, 3, generator=g) / (64 * 27) ** 0.5
    G = torch.randn((B, 64) + size, generator=g)
    # fp64 reference of dW[co, ci, k] = sum G[b, co, o] x_pad[b, ci, o + k]: evaluated tap by tap (the fp64 convolution
    # backward of a 64^3 volume is slow on the host)
    xp = F.pad(x.double(), (1, 1, 1, 1, 1, 1))
    Gd = G.double()
    D, H, W = size
    ref = torch.empty(64, 64, 27, dtype=torch.float64)
    for k in range(27):
        kz, ky, kx = k // 9, (k // 3) % 3, k % 3
        ref[:, :, k] = torch.einsum("bgzyx,bczyx->gc", Gd, xp[:, :, kz:kz + D, ky:ky + H, kx:kx + W])
    ref = ref.view(64, 64, 3, 3, 3)
    got = ops.conv3d_wrw(G.to(DEV), x.to(DEV), 3, 1, 1)
    scale = float(ref.abs().max())
    assert float((got.cpu().double() - ref).abs().max()) < 3e-5 * scale
    # through the layer's autograd node
    xd, wd = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_()
    y = convgrad._ConvFn.apply(xd, wd, None, (1, 1, 1), (1, 1, 1), False)
    gx, gw = torch.autograd.grad((y * G.to(DEV)).sum(), [xd, wd])
    assert float((gw.cpu().double() - ref).abs().max()) < 3e-5 * scale
    gx_ref = F.conv3d(G.double(), w.transpose(0, 1).flip(2, 3, 4).double(), None, 1, 1)
    assert float((gx.cpu().double() - gx_ref).abs().max()) < TOL * float(gx_ref.abs().max())


def test_wino_weight_gradient_is_not_taken_where_it_does_not_apply(ops):
    assert not ops.conv3d_wrw_takes_winograd(2, 64, 64, (32, 32, 32), (32, 32, 32), 3, 1, 1, 0, 0)   # rows of 32
    assert not ops.conv3d_wrw_takes_winograd(2, 64, 32, (32, 32, 64), (32, 32, 64), 3, 1, 1, 0, 0)   # 32 source channels
    assert not ops.conv3d_wrw_takes_winograd(1, 64, 64, (8, 8, 64), (8, 8, 64), 3, 1, 1, 0, 0)       # too few bricks
    assert not ops.conv3d_wrw_takes_winograd(2, 64, 64, (32, 32, 64), (32, 32, 64), 3, 1, 1, 4, 0)   # misaligned
