"""fs_conv3d_fwd vs MIOpen per IFNet-3D layer shape: max error and time (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import opticalflowscivis_amd  # noqa: F401  (sets the MIOpen env)
from opticalflowscivis_amd import ops


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(cin, cout, k, s, size, B=2, wmode=0):
    torch.manual_seed(0)
    x = torch.randn(B, cin, size, size, size, device="cuda")
    w = torch.randn(cout, cin, k, k, k, device="cuda") / (cin * k ** 3) ** 0.5
    bias = torch.randn(cout, device="cuda")
    if wmode:
        wm = w.transpose(0, 1).flip(2, 3, 4).contiguous()  # stored [Cin][Cout], flipped
    else:
        wm = w
    ref = F.conv3d(x, w, bias, s, 1)
    got = ops.conv3d_fwd(x, wm, bias, k, s, 1, wmode)
    err = float((got - ref).abs().max())
    tm = t(lambda: F.conv3d(x, w, bias, s, 1))
    th = t(lambda: ops.conv3d_fwd(x, wm, bias, k, s, 1, wmode))
    fl = 2.0 * ref.numel() * cin * k ** 3
    print("cin=%3d cout=%3d k%d s%d in=%3d^3 wmode %d: err %.2e | miopen %.3f ms (%.1f TF/s) | hip %.3f ms (%.1f TF/s)" % (
        cin, cout, k, s, size, wmode, err, tm, fl / tm / 1e9, th, fl / th / 1e9), flush=True)


S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
case(5, 7, 3, 1, 20, B=1)
case(5, 7, 4, 2, 22, B=1)
case(64, 64, 3, 1, S // 4)
case(64, 64, 3, 1, S // 4, wmode=1)
case(64, 64, 3, 1, S // 8)
case(128, 128, 3, 1, S // 16)
case(128, 128, 3, 1, S // 16, wmode=1)
case(64, 64, 3, 1, S // 8, wmode=1)
case(11, 32, 4, 2, S)
case(12, 32, 4, 2, S)
case(32, 64, 4, 2, S // 2)
case(11, 32, 4, 2, S // 2)
case(32, 64, 4, 2, S // 4)
case(2, 64, 4, 2, S // 4)
case(64, 128, 4, 2, S // 8)
