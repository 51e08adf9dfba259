"""Two launches each of the Winograd-domain trunk kernels (forward and weight gradient, 64 -> 64 at 2 x 64^3) and of the
direct kernels they replace (FLOWSCI_{FWD,WRW}_NO_WINO=1 in a second run), for `rocprofv3 --pmc <group>`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opticalflowscivis_amd import ops
x = torch.randn(2, 64, 64, 64, 64, device="cuda")
w = torch.randn(64, 64, 3, 3, 3, device="cuda") * 0.02
g = torch.randn(2, 64, 64, 64, 64, device="cuda")
for _ in range(2):
    ops.conv3d_fwd(x, w, None, 3, 1, 1, 0)
    ops.conv3d_wrw(g, x, 3, 1, 1)
torch.cuda.synchronize()
