"""Where one brick of the DMA-staged weight-gradient kernel spends its cycles: s_memtime stamps of the matrix
waves (library built with -DFS_WRW_STAMPS; csrc/convwrw.hip).  GPU box only."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opticalflowscivis_amd import ops, _lib
g = torch.randn(2, 64, 64, 64, 64, device="cuda")
src = torch.randn(2, 64, 64, 64, 64, device="cuda")
for _ in range(3):
    ops.conv3d_wrw(g, src, 3, 1, 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.conv3d_wrw(g, src, 3, 1, 1)
e1.record(); torch.cuda.synchronize()
print("ms per launch %.4f" % (e0.elapsed_time(e1) / 5))
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 32)()
if hasattr(L, "fs_debug_wrw_stamps"):
    L.fs_debug_wrw_stamps(buf)
    names = ["begin", "issue", "mfma", "vmwait", "barrier", "-", "-", "bricks"]
    for w in range(4):
        n = buf[w * 8 + 7] or 1
        print("wave %d: " % w + "  ".join("%s %.0f" % (names[i], buf[w * 8 + i] / n) for i in range(5)) + "  (cycles per brick, %d bricks)" % n)
