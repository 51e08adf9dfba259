"""The k4 s2 p1 forward-convolution shapes of the 256^3 Flow-3D step on fs_conv3d_fwd: ms per launch, useful TFLOP/s and the
error against an fp64 evaluation of a crop (GPU box only).  Run on the product library (round 5: fp32-accurate kernel on the
bf16 matrix rate, csrc/convfwd_s3.hpp) and on the ablation build with FLOWSCI_FWD_NO_S3=1 (the fp32-MFMA kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import opticalflowscivis_amd  # noqa: F401
from opticalflowscivis_amd import ops


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(cin, cout, size, what, B=2):
    torch.manual_seed(0)
    x = torch.randn(B, cin, size, size, size, device="cuda")
    w = torch.randn(cout, cin, 4, 4, 4, device="cuda") / (cin * 64) ** 0.5
    bias = torch.randn(cout, device="cuda")
    got = ops.conv3d_fwd(x, w, bias, 4, 2, 1, 0)
    # fp64 on a crop: output rows [0, 8) x [0, 16) x [0, 32) of sample 0 need input [0, 17) x [0, 33) x [0, 65) (pad 1 on the low side)
    xc = F.pad(x[:1, :, :17, :33, :65].double(), (1, 0, 1, 0, 1, 0))
    ref = F.conv3d(xc, w.double(), bias.double(), 2, 0)[:, :, :8, :16, :32]
    err = float((got[:1, :, :8, :16, :32].double() - ref).abs().max()) / float(ref.abs().max())
    ms = t(lambda: ops.conv3d_fwd(x, w, bias, 4, 2, 1, 0))
    fl = 2.0 * got.numel() * cin * 64
    print("%-34s %2d -> %2d at %3d^3: %.4f ms  %6.1f TFLOP/s  rel err vs fp64 %.2e" % (what, cin, cout, size, ms, fl / ms / 1e9, err), flush=True)


print("library:", os.environ.get("FLOWSCI_HIP_LIBRARY", "product"), "| FLOWSCI_FWD_NO_S3 =", os.environ.get("FLOWSCI_FWD_NO_S3"))
case(11, 32, 256, "conv0a (blocks at scale 1)")
case(12, 32, 256, "conv0a (teacher)")
case(32, 64, 128, "conv0b")
case(6, 32, 256, "flow head deconv2 input gradient")
case(1, 32, 256, "mask head deconv2 input gradient")
case(32, 64, 128, "head deconv1 input gradient")
case(11, 32, 128, "conv0a (block at scale 2)")
case(32, 64, 66, "conv0b (block at scale 2)")
