"""Per-brick fixed cost of the loader-wave convolution kernels: the same layer with Cin and 2 Cin input channels -- a brick's
periods double, its fixed part (workgroup launch, first chunk with nothing to overlap, epilogue, store drain) does not:
T(C) = n (C p + f), so n f = 2 T(C) - T(2C).  k4 s2 forward (conv0a / conv0b shapes), transposed k4 s2, k3 direct."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opticalflowscivis_amd import ops


def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def fwd(cin, cout, k, s, size, B=2):
    x = torch.randn(B, cin, size, size, size, device="cuda")
    w = torch.randn(cout, cin, k, k, k, device="cuda") * 0.02
    b = torch.randn(cout, device="cuda")
    return t(lambda: ops.conv3d_fwd(x, w, b, k, s, 1, 0))


def tr(cin, cout, size, B=2):
    x = torch.randn(B, cin, size, size, size, device="cuda")
    w = torch.randn(cin, cout, 4, 4, 4, device="cuda") * 0.02
    return t(lambda: ops.conv3d_tr(x, w, None))


for name, f, c in (("fwd k4 s2 C->32 256^3", lambda c: fwd(c, 32, 4, 2, 256), 12),
                   ("fwd k4 s2 C->64 128^3", lambda c: fwd(c, 64, 4, 2, 128), 32),
                   ("fwd k3 s1 C->64 64^3 (direct when C != 64)", lambda c: fwd(c, 64, 3, 1, 64), 32),
                   ("tr  k4 s2 C->32 64^3 -> 128^3", lambda c: tr(c, 32, 64), 64),
                   ("tr  k4 s2 C->6 128^3 -> 256^3", lambda c: tr(c, 6, 128), 32)):
    t1, t2 = f(c), f(2 * c)
    print("%-46s T(%d) %.3f ms  T(%d) %.3f ms  fixed part %.3f ms = %.0f %% of T(%d)" % (name, c, t1, 2 * c, t2, 2 * t1 - t2, 100 * (2 * t1 - t2) / t1, c), flush=True)
