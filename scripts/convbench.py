"""Which IFNet-3D layer shapes hit MIOpen's slow GEMM (im2col/col2im) path?  (GPU box only)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import opticalflowscivis_amd  # sets the MIOpen env


def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def conv_case(cin, cout, k, s, size, tr=False, B=2):
    dev = "cuda"
    if tr:
        x = torch.randn(B, cin, size, size, size, device=dev, requires_grad=True)
        w = torch.randn(cin, cout, k, k, k, device=dev)
        f = lambda: F.conv_transpose3d(x, w, None, s, 1)
    else:
        x = torch.randn(B, cin, size, size, size, device=dev, requires_grad=True)
        w = torch.randn(cout, cin, k, k, k, device=dev)
        f = lambda: F.conv3d(x, w, None, s, 1)
    y = f()
    g = torch.randn_like(y)
    tf = t(f)
    tb = t(lambda: torch.autograd.grad(y, [x], g, retain_graph=True))
    print("%s cin=%3d cout=%3d k%d s%d in=%d^3: fwd %.2f ms, bwd-data %.2f ms" % (
        "deconv" if tr else "conv  ", cin, cout, k, s, size, tf, tb), flush=True)


S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for cin in (11, 12, 16):
    conv_case(cin, 32, 4, 2, S)
conv_case(32, 64, 4, 2, S // 2)
conv_case(64, 64, 3, 1, S // 4)
conv_case(64, 32, 4, 2, S // 4, tr=True)
for cout in (6, 8, 1):
    conv_case(32, cout, 4, 2, S // 2, tr=True)
