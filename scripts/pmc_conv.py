"""Two launches each of the 64-channel k3 forward and weight-gradient kernels at 64^3 (for rocprofv3 --pmc)."""
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
