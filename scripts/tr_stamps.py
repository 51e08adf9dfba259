"""Where one brick of the parities-in-rows heads kernel (convtr_p8_kernel) spends its cycles: s_memtime stamps of
the matrix waves (library built with -DFS_TR_STAMPS; csrc/convtr.hip).  GPU box only."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opticalflowscivis_amd import ops, _lib
x = torch.randn(2, 32, 128, 128, 128, device="cuda")
w = torch.randn(32, 6, 4, 4, 4, device="cuda") * 0.05
for _ in range(3):
    ops.conv3d_tr(x, w, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.conv3d_tr(x, w, None)
e1.record(); torch.cuda.synchronize()
print("ms per launch %.4f" % (e0.elapsed_time(e1) / 5))
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 32)()
if hasattr(L, "fs_debug_tr_stamps"):
    L.fs_debug_tr_stamps(buf)
    for w_ in range(4):
        n = buf[w_ * 8 + 7] or 1
        print("wave %d: mfma %.0f  barrier %.0f  epilogue %.0f  (cycles per brick, %d bricks)" % (
            w_, buf[w_ * 8] / n, buf[w_ * 8 + 1] / n, buf[w_ * 8 + 2] / n, n))
