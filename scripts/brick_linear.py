"""T(C) of the k4 s2 C -> 32 forward layer at 256^3 for C = 12, 24, 36, 48: is the part that does not scale with C per
brick or per launch?"""
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
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for c in (12, 24, 36, 48):
    x = torch.randn(2, c, S, S, S, device="cuda"); w = torch.randn(32, c, 4, 4, 4, device="cuda") * 0.02; b = torch.randn(32, device="cuda")
    print("S %d C %2d: %.3f ms" % (S, c, t(lambda: ops.conv3d_fwd(x, w, b, 4, 2, 1, 0))), flush=True)
    del x
