"""fs_conv3d_tr vs MIOpen for the IFNet-3D head / conv0 input-gradient shapes (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import opticalflowscivis_amd  # noqa: F401
from opticalflowscivis_amd import ops


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(cin, cout, size, B=2):
    torch.manual_seed(0)
    x = torch.randn(B, cin, size, size, size, device="cuda")
    w = torch.randn(cin, cout, 4, 4, 4, device="cuda") / (cin * 8) ** 0.5
    bias = torch.randn(cout, device="cuda")
    ref = F.conv_transpose3d(x, w, bias, 2, 1)
    got = ops.conv3d_tr(x, w, bias)
    err = float((got - ref).abs().max())
    tm = t(lambda: F.conv_transpose3d(x, w, bias, 2, 1))
    th = t(lambda: ops.conv3d_tr(x, w, bias))
    fl = 2.0 * x.numel() * cout * 64
    print("cin=%3d cout=%3d in=%3d^3: err %.2e | miopen %.3f ms (%.1f TF/s) | hip %.3f ms (%.1f TF/s)" % (
        cin, cout, size, err, tm, fl / tm / 1e9, th, fl / th / 1e9), flush=True)


S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
case(5, 7, 9, B=1)
case(6, 20, 11, B=1)
case(64, 32, S // 4)
case(32, 6, S // 2)
case(32, 1, S // 2)
case(32, 11, S // 2)
case(32, 12, S // 2)
case(64, 32, S // 8)
case(32, 6, S // 4)
case(32, 11, S // 4)
