"""fs_conv3d_fwd on the 64 -> 64 k3 layer of the 64^3 trunk (B = 2): the Winograd F(4,3) kernel vs F(2,3)
(FLOWSCI_FWD_NO_WINO4=1) vs the direct loader-wave kernel (FLOWSCI_FWD_NO_WINO=1), each in its own process: time and
error against fp64 on a sub-volume."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from opticalflowscivis_amd import ops

torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H = int(sys.argv[3]) if len(sys.argv) > 3 else 64
CIN = int(os.environ.get("WINO_BENCH_CIN", "64"))  # fewer input channels = fewer periods per brick: separates the per-brick fixed cost
x = torch.randn(B, CIN, D, H, 64, device="cuda")
w = torch.randn(64, CIN, 3, 3, 3, device="cuda") / (CIN * 27) ** 0.5
b = torch.randn(64, device="cuda")
for wmode in ((0, 1) if CIN == 64 else (0,)):
    for _ in range(3):
        y = ops.conv3d_fwd(x, w, b if wmode == 0 else None, 3, 1, 1, wmode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = ops.conv3d_fwd(x, w, b if wmode == 0 else None, 3, 1, 1, wmode)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    ww = w if wmode == 0 else w.transpose(0, 1).flip(2, 3, 4)
    ref = F.conv3d(x[:1, :, :10].double(), ww.double(), (b.double() if wmode == 0 else None), 1, 1)[:, :, 1:9]
    err = float((y[:1, :, 1:9].double() - ref).abs().max()) / float(ref.abs().max())
    fl = 2 * y.numel() * CIN * 27
    print("B=%d %dx%dx64 wmode %d: %.3f ms/launch (incl. the weight re-layout launch) = %.1f TFLOP/s direct-equivalent; max err vs fp64 %.2e "
          "[FLOWSCI_FWD_NO_WINO=%s FLOWSCI_FWD_NO_WINO4=%s FLOWSCI_FWD_NO_WINO2D=%s]" % (B, D, H, wmode, ms, fl / ms / 1e9, err, os.environ.get("FLOWSCI_FWD_NO_WINO", ""),
                                                                 os.environ.get("FLOWSCI_FWD_NO_WINO4", ""), os.environ.get("FLOWSCI_FWD_NO_WINO2D", "")), flush=True)
