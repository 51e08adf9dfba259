"""The 2-D Winograd trunk kernel in every fused form it is launched in (64 -> 64, k3, 2 x 64^3; XT = 8 variant: 2 x 32^3):
time per launch and a CRC-32 of every output.  Run once per library / switch (the ablation build with
FLOWSCI_WINO2D_R3=1 is the round-3 kernel): equal CRCs = bit-identical outputs.  The PReLU-backward form's two
gradient vectors are sums in a different order (printed as values, not as CRCs)."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opticalflowscivis_amd import ops


def t(fn, n=20):
    out = fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out


def crc(*ts):
    return " ".join("%08x" % zlib.crc32(a.detach().contiguous().cpu().numpy().tobytes()) for a in ts)


torch.manual_seed(0)
for S in (64, 32):
    x = torch.randn(2, 64, S, S, S, device="cuda"); w = torch.randn(64, 64, 3, 3, 3, device="cuda") * .02
    b = torch.randn(64, device="cuda"); a = torch.rand(64, device="cuda"); a1 = torch.rand(1, device="cuda"); r = torch.randn_like(x)
    forms = [
        ("plain", lambda: (ops.conv3d_fwd(x, w, b, 3, 1, 1, 0),)),
        ("prelu", lambda: ops.conv3d_fwd(x, w, b, 3, 1, 1, 0, a)),
        ("prelu+res", lambda: ops.conv3d_fwd(x, w, b, 3, 1, 1, 0, a, r)),
        ("prelu1+res", lambda: ops.conv3d_fwd(x, w, b, 3, 1, 1, 0, a1, r)),
        ("wmode1", lambda: (ops.conv3d_fwd(x, w, None, 3, 1, 1, 1),)),
        ("wmode1+add", lambda: (ops.conv3d_fwd(x, w, None, 3, 1, 1, 1, None, r),)),
    ]
    for name, fn in forms:
        ms, out = t(fn)
        print("%d^3 %-11s %.4f ms  %s" % (S, name, ms, crc(*out)), flush=True)
    ms, out = t(lambda: ops.conv3d_k3_grad_input_dprelu(x, w, r, a))
    print("%d^3 %-11s %.4f ms  %s  ga %.6e gb %.6e" % (S, "dprelu", ms, crc(out[0]), float(out[1].double().sum()), float(out[2].double().sum())), flush=True)
    ms, out = t(lambda: ops.conv3d_k3_grad_input_dprelu(x, w, r, a1))
    print("%d^3 %-11s %.4f ms  %s  ga %.6e gb %.6e" % (S, "dprelu1", ms, crc(out[0]), float(out[1].double().sum()), float(out[2].double().sum())), flush=True)
