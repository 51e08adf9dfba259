"""Where the persistent 2-D Winograd trunk kernel (conv3d_wino2d_ps_kernel) spends its cycles: s_memtime sums per wave of one
workgroup (library built by `make -C opticalflowscivis_amd/csrc w2stamps`, loaded through FLOWSCI_HIP_LIBRARY).  GPU box only."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opticalflowscivis_amd import ops, _lib
x = torch.randn(2, 64, 64, 64, 64, device="cuda"); w = torch.randn(64, 64, 3, 3, 3, device="cuda") * .02
b = torch.randn(64, device="cuda")
for _ in range(3):
    ops.conv3d_fwd(x, w, b, 3, 1, 1, 0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.conv3d_fwd(x, w, b, 3, 1, 1, 0)
e1.record(); torch.cuda.synchronize()
print("ms per launch %.4f (incl. the weight re-layout launch)" % (e0.elapsed_time(e1) / 10))
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 64)()
if not hasattr(L, "fs_debug_w2_stamps"):
    sys.exit("library without stamps: %s" % _lib.LIB_PATH)
L.fs_debug_w2_stamps(buf)
periods = 8 * 32
for wv in range(4):
    v = [buf[wv * 8 + k] for k in range(8)]
    print("matrix wave %d: MFMA loop %6.0f  period barrier %5.0f  (s_memtime ticks per period) | epilogue: Ax^T %5.0f  requests + exchange write %5.0f  barriers %5.0f  read + finish + stores %5.0f (per brick)" % (
        wv, v[0] / periods, v[1] / periods, v[5] / 8, v[2] / 8, v[3] / 8, v[4] / 8))
for wv in range(4, 8):
    v = [buf[wv * 8 + k] for k in range(8)]
    print("loader wave %d: waits %5.0f  work %5.0f  LDS drain %4.0f  period barrier %5.0f  (per period) | exchange barriers %6.0f (per brick)" % (
        wv - 4, v[0] / periods, v[1] / periods, v[2] / periods, v[3] / periods, v[4] / 8))
