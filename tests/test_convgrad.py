"""CPU: the im2col + split-K GEMM weight gradient equals autograd's convolution backward."""
import pytest
import torch
import torch.nn.functional as F

from opticalflowscivis_amd import convgrad


@pytest.mark.parametrize("cfg", [
    dict(cin=5, cout=7, k=3, s=1, p=1, size=(6, 7, 8), tr=False),
    dict(cin=11, cout=8, k=4, s=2, p=1, size=(8, 10, 12), tr=False),
    dict(cin=3, cout=4, k=4, s=2, p=1, size=(9, 7, 11), tr=False),   # odd sizes: trailing rows unused
    dict(cin=6, cout=5, k=4, s=2, p=1, size=(4, 5, 6), tr=True),
    dict(cin=8, cout=1, k=4, s=2, p=1, size=(3, 4, 5), tr=True),
])
def test_wrw_gemm_matches_autograd(cfg):
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


def test_splitk_matmul_splits():
    g = torch.Generator().manual_seed(1)
    G, C = torch.randn(6, 2048 * 8, generator=g), torch.randn(2048 * 8, 10, generator=g)
    out = convgrad._splitk_matmul(G, C)
    assert float((out - G @ C).abs().max()) < 1e-3


def test_modules_keep_state_dict_keys():
    m = convgrad.Conv3d(3, 4, 3, 1, 1)
    t = convgrad.ConvTranspose3d(4, 2, 4, 2, 1)
    assert sorted(m.state_dict()) == ["bias", "weight"] and sorted(t.state_dict()) == ["bias", "weight"]
    x = torch.randn(1, 3, 4, 4, 4)
    assert torch.equal(m(x), F.conv3d(x, m.weight, m.bias, 1, 1))  # CPU -> stock path
