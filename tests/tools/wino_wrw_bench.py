"""fs_conv3d_wrw on the 64 -> 64 k3 layer of the 64^3 trunk (B = 2): Winograd-domain kernel vs the direct DMA kernel
(FLOWSCI_WRW_NO_WINO=1 in a second process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from opticalflowscivis_amd import ops

torch.manual_seed(0)
g = torch.randn(2, 64, 64, 64, 64, device="cuda")
x = torch.randn(2, 64, 64, 64, 64, device="cuda")
for _ in range(3):
    dw = ops.conv3d_wrw(g, x, 3, 1, 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    dw = ops.conv3d_wrw(g, x, 3, 1, 1)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
fl = 2 * g.numel() * 64 * 27
print("wrw: %.3f ms/launch (incl. the zero fill) = %.1f TFLOP/s direct-equivalent; dW abs sum %.6e "
      "[FLOWSCI_WRW_NO_WINO=%s FLOWSCI_WRW_NO_WINO4=%s]" % (
          ms, fl / ms / 1e9, float(dw.double().abs().sum()), os.environ.get("FLOWSCI_WRW_NO_WINO", ""),
          os.environ.get("FLOWSCI_WRW_NO_WINO4", "")), flush=True)
