"""Epilogue share of the loader-wave transposed-convolution kernels: the layer timed on the ablation build with and without
FLOWSCI_TR_AB=4 (epilogue skipped: wrong results by design).  64 -> 32 at 64^3 -> 128^3 (convtr_mfma_ws), 32 -> 11 at
128^3 -> 256^3 (convtr_mfma16_ws), both with bias + PReLU output as in the step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opticalflowscivis_amd import ops
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for cin, cout, size in ((64, 32, 64), (32, 11, 128), (128, 64, 16)):
    x = torch.randn(2, cin, size, size, size, device="cuda"); w = torch.randn(cin, cout, 4, 4, 4, device="cuda") * 0.02
    b = torch.randn(cout, device="cuda"); a = torch.rand(cout, device="cuda")
    print("tr %3d -> %2d at %3d^3: plain %.3f ms   bias + PReLU (two outputs) %.3f ms   [FLOWSCI_TR_AB=%s]" % (
        cin, cout, size, t(lambda: ops.conv3d_tr(x, w, None)), t(lambda: ops.conv3d_tr(x, w, b, None, a)), os.environ.get("FLOWSCI_TR_AB", "")), flush=True)
