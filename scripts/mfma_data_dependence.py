import sys; sys.path.insert(0, "/root/repo")
import torch
from opticalflowscivis_amd import ops
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
w = torch.randn(64, 64, 3, 3, 3, device="cuda") * .02
fl = 2.0 * 2 * 64**3 * 64 * 64 * 27
for name, x in (("zeros", torch.zeros(2, 64, 64, 64, 64, device="cuda")), ("randn", torch.randn(2, 64, 64, 64, 64, device="cuda")),
                ("randn*1e3", 1e3 * torch.randn(2, 64, 64, 64, 64, device="cuda"))):
    for wn, ww in (("w randn", w), ("w zeros", torch.zeros_like(w))):
        ms = t(lambda: ops.conv3d_fwd(x, ww, None, 3, 1, 1, 0))
        print("fwd  x %-10s %-8s %.3f ms  %.1f TF/s" % (name, wn, ms, fl / ms / 1e9), flush=True)
g = torch.randn(2, 64, 64, 64, 64, device="cuda")
for name, x in (("zeros", torch.zeros(2, 64, 64, 64, 64, device="cuda")), ("randn", torch.randn(2, 64, 64, 64, 64, device="cuda"))):
    for gn, gg in (("g randn", g), ("g zeros", torch.zeros_like(g))):
        ms = t(lambda: ops.conv3d_wrw(gg, x, 3, 1, 1))
        print("wrw  x %-10s %-8s %.3f ms  %.1f TF/s" % (name, gn, ms, fl / ms / 1e9), flush=True)
