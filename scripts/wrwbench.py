"""Micro-benchmark of fs_conv3d_wrw on the IFNet-3D layer shapes at 256^3 (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opticalflowscivis_amd import ops


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(name, cg, cs, k, s, out, B=2):
    inn = (out - 1) * s + k - 2
    g = torch.randn(B, cg, out, out, out, device="cuda")
    src = torch.randn(B, cs, inn, inn, inn, device="cuda")
    ms = t(lambda: ops.conv3d_wrw(g, src, k, s, 1))
    fl = 2.0 * cg * cs * k ** 3 * B * out ** 3
    print("%-28s Cg=%3d Cs=%3d k%d s%d out=%3d^3: %.3f ms  %.1f TFLOP/s" % (name, cg, cs, k, s, out, ms, fl / ms / 1e9), flush=True)


case("conv0a (11->32)", 32, 11, 4, 2, 128)
case("conv0b (32->64)", 64, 32, 4, 2, 64)
case("convblock (64->64)", 64, 64, 3, 1, 64)
case("deconv1 (64->32)", 64, 32, 4, 2, 64)
case("deconv2 flow (32->6)", 32, 6, 4, 2, 128)
case("deconv2 mask (32->1)", 32, 1, 4, 2, 128)
case("block0 conv (128->128) @16^3", 128, 128, 3, 1, 16)
