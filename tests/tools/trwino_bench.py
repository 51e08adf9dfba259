"""fs_conv3d_tr on the 64 -> 32 layer between the 64^3 trunk and the 128^3 grid (B = 2): the Winograd F(4,2) kernel vs the
class kernel (FLOWSCI_TR_NO_WINO=1 in a second process): time, direct-equivalent TFLOP/s, error against fp64 on a sub-volume."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from opticalflowscivis_amd import ops

torch.manual_seed(0)
x = torch.randn(2, 64, 64, 64, 64, device="cuda")
w = torch.randn(64, 32, 4, 4, 4, device="cuda") / (64 * 8) ** 0.5
b = torch.randn(32, device="cuda")
a = torch.rand(32, device="cuda") - 0.3
for mode in ("plain", "prelu"):
    fn = (lambda: ops.conv3d_tr(x, w, b)) if mode == "plain" else (lambda: ops.conv3d_tr(x, w, b, None, a))
    for _ in range(3):
        y = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    y = y[0] if isinstance(y, tuple) else y
    ref = F.conv_transpose3d(x[:1, :, :6].double(), w.double(), b.double(), 2, 1)[:, :, 1:9]
    err = float((y[:1, :, 1:9].double() - ref).abs().max()) / float(ref.abs().max())
    fl = 2 * x.numel() * 32 * 64
    print("%s: %.3f ms/launch (incl. the weight re-layout launch) = %.1f TFLOP/s direct-equivalent; max err vs fp64 %.2e "
          "[FLOWSCI_TR_NO_WINO=%s]" % (mode, ms, fl / ms / 1e9, err, os.environ.get("FLOWSCI_TR_NO_WINO", "")), flush=True)
