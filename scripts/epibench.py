import os, sys
sys.path.insert(0, "/root/repo")
import torch
from opticalflowscivis_amd import ops
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
x = torch.randn(2, 64, 64, 64, 64, device="cuda"); w = torch.randn(64, 64, 3, 3, 3, device="cuda") * .02
b = torch.randn(64, device="cuda"); a = torch.rand(64, device="cuda"); r = torch.randn_like(x)
print("plain      %.3f" % t(lambda: ops.conv3d_fwd(x, w, b, 3, 1, 1, 0)))
print("prelu      %.3f" % t(lambda: ops.conv3d_fwd(x, w, b, 3, 1, 1, 0, a)))
print("prelu+res  %.3f" % t(lambda: ops.conv3d_fwd(x, w, b, 3, 1, 1, 0, a, r)))
print("wmode1     %.3f" % t(lambda: ops.conv3d_fwd(x, w, None, 3, 1, 1, 1)))
print("wmode1+add %.3f" % t(lambda: ops.conv3d_fwd(x, w, None, 3, 1, 1, 1, None, r)))
print("plain nobias %.3f" % t(lambda: ops.conv3d_fwd(x, w, None, 3, 1, 1, 0)))
print("wmode1 bias  %.3f" % t(lambda: ops.conv3d_fwd(x, w, b, 3, 1, 1, 1)))
print("plain again  %.3f" % t(lambda: ops.conv3d_fwd(x, w, b, 3, 1, 1, 0)))
