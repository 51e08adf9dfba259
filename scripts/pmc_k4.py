"""Two launches each of the k4 s2 convolution family at the conv0a / conv0b / head shapes of the 2 x 256^3 step (forward,
transposed, weight gradient), for `rocprofv3 --pmc <group>`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opticalflowscivis_amd import ops
x = torch.randn(2, 11, 256, 256, 256, device="cuda")
w = torch.randn(32, 11, 4, 4, 4, device="cuda") * 0.02
x2 = torch.randn(2, 32, 128, 128, 128, device="cuda")
w2 = torch.randn(64, 32, 4, 4, 4, device="cuda") * 0.02
for _ in range(2):
    y = ops.conv3d_fwd(x, w, None, 4, 2, 1, 0)          # conv0a forward  11 -> 32
    ops.conv3d_tr(y, w, None, x.shape[2:])              # conv0a input gradient 32 -> 11
    ops.conv3d_wrw(y, x, 4, 2, 1)                       # conv0a weight gradient
    y2 = ops.conv3d_fwd(x2, w2, None, 4, 2, 1, 0)       # conv0b forward  32 -> 64
    ops.conv3d_tr(y2, w2, None, x2.shape[2:])           # conv0b input gradient 64 -> 32
    ops.conv3d_wrw(y2, x2, 4, 2, 1)                     # conv0b weight gradient
torch.cuda.synchronize()
