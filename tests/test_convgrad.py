"""CPU: the convgrad modules keep torch.nn's parameters / keys, and on non-GPU tensors every layer runs the
torch.nn base class (ATen) -- forward, input, weight and bias gradients equal stock autograd."""
import pytest
import torch
import torch.nn.functional as F

from opticalflowscivis_amd import convgrad, ops


@pytest.mark.parametrize("cfg", [
    dict(cin=5, cout=7, k=3, s=1, p=1, size=(6, 7, 8), tr=False),
    dict(cin=11, cout=8, k=4, s=2, p=1, size=(8, 10, 12), tr=False),
    dict(cin=3, cout=4, k=4, s=2, p=1, size=(9, 7, 11), tr=False),   # odd sizes: trailing rows unused
    dict(cin=6, cout=5, k=4, s=2, p=1, size=(4, 5, 6), tr=True),
    dict(cin=8, cout=1, k=4, s=2, p=1, size=(3, 4, 5), tr=True),
])
def test_convfn_off_gpu_is_aten(cfg):
    g = torch.Generator().manual_seed(0)
    B = 2
    x = torch.randn((B, cfg["cin"]) + cfg["size"], generator=g, requires_grad=True)
    if cfg["tr"]:
        w = torch.randn(cfg["cin"], cfg["cout"], *(cfg["k"],) * 3, generator=g, requires_grad=True)
        fn = F.conv_transpose3d
    else:
        w = torch.randn(cfg["cout"], cfg["cin"], *(cfg["k"],) * 3, generator=g, requires_grad=True)
        fn = F.conv3d
    b = torch.randn(cfg["cout"], generator=g, requires_grad=True)
    s3, p3 = (cfg["s"],) * 3, (cfg["p"],) * 3
    y_ref = fn(x, w, b, s3, p3)
    G = torch.randn(y_ref.shape, generator=g)
    ref = torch.autograd.grad((y_ref * G).sum(), [x, w, b])
    y = convgrad._ConvFn.apply(x, w, b, s3, p3, cfg["tr"])
    got = torch.autograd.grad((y * G).sum(), [x, w, b])
    assert torch.equal(y, y_ref)
    for a, r in zip(got, ref):
        assert a.shape == r.shape
        assert float((a - r).abs().max()) < 2e-4 * max(1.0, float(r.abs().max()))


def test_modules_keep_state_dict_keys():
    m = convgrad.Conv3d(3, 4, 3, 1, 1)
    t = convgrad.ConvTranspose3d(4, 2, 4, 2, 1)
    assert sorted(m.state_dict()) == ["bias", "weight"] and sorted(t.state_dict()) == ["bias", "weight"]
    x = torch.randn(1, 3, 4, 4, 4)
    assert torch.equal(m(x), F.conv3d(x, m.weight, m.bias, 1, 1))  # CPU -> torch.nn base class


def test_size_predicates_mirror_the_kernel_limits():
    """a layer beyond the kernels' 32-bit chunk offsets must be routed to torch.nn, not raise.
    At 512^3 the first conv0 layer (in 512^3) and the last head layer (out 512^3) are over the limits."""
    assert ops.conv3d_fwd_fits((256,) * 3, (128,) * 3, 4) and ops.conv3d_fwd_fits((64,) * 3, (64,) * 3, 3)
    assert not ops.conv3d_fwd_fits((1024,) * 3, (512,) * 3, 4)
    assert ops.conv3d_wrw_fits((256,) * 3, (128,) * 3) and not ops.conv3d_wrw_fits((512,) * 3, (256,) * 3)
    assert not ops.conv3d_wrw_fits((256,) * 3, (512,) * 3)
    assert ops.conv3d_tr_fits((128,) * 3) and not ops.conv3d_tr_fits((1024,) * 3)
