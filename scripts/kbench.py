"""Micro-benchmark of the warp3d pair kernels at the BASELINE size (GPU box only).
usage: kbench.py [S=256] [smooth|noise|zero]   (FS_W3_VARIANT selects the tile geometry)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflowscivis_amd import ops


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def make_flow(kind, B, S, dev, ch=6):
    if kind == "noise":
        return (torch.rand(B, ch, S, S, S, device=dev) * 2 - 1) * 4
    if kind == "zero":
        return torch.zeros(B, ch, S, S, S, device=dev)
    ax = torch.linspace(0, 6.28318, S, device=dev)
    a = 4 * torch.sin(ax).view(1, S, 1, 1) * torch.cos(ax).view(1, 1, S, 1).expand(B, S, S, S)
    b = 3 * torch.cos(ax * 2).view(1, 1, 1, S).expand(B, S, S, S) + 0 * ax.view(1, S, 1, 1)
    c = 2 * torch.sin(ax * 3).view(1, 1, S, 1) * torch.sin(ax).view(1, 1, 1, S).expand(B, S, S, S)
    return torch.stack([a, b, c, -a, c, b][:ch], 1).contiguous()


def main():
    dev = "cuda:0"
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    kind = sys.argv[2] if len(sys.argv) > 2 else "smooth"
    B = 2
    i0 = torch.rand(B, 1, S, S, S, device=dev)
    i1 = torch.rand(B, 1, S, S, S, device=dev)
    f = make_flow(kind, B, S, dev).requires_grad_()
    G0, G1 = torch.randn_like(i0), torch.randn_like(i1)
    nvox = B * S ** 3
    tf = timeit(lambda: ops.warp_pair(i0, i1, f.detach()))
    o0, o1 = ops.warp_pair(i0, i1, f)
    tb = timeit(lambda: torch.autograd.grad([o0, o1], [f], [G0, G1], retain_graph=True))
    print("variant %s %s %dx%d^3: pair fwd %.3f ms %.0f GB/s | pair bwd(flow) %.3f ms %.0f GB/s" % (
        os.environ.get("FS_W3_VARIANT", "0"), kind, B, S, tf, nvox * 40 / tf / 1e6, tb, nvox * 64 / tb / 1e6))


if __name__ == "__main__":
    main()
