"""Micro-benchmark of individual HIP kernels at BASELINE sizes (GPU box only)."""
import sys
import os
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


def main():
    dev = "cuda:0"
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    B = 2
    x = torch.rand(B, 1, S, S, S, device=dev)
    kind = sys.argv[2] if len(sys.argv) > 2 else "smooth"
    if kind == "noise":
        f = ((torch.rand(B, 3, S, S, S, device=dev) * 2 - 1) * 4)
    else:  # low-frequency field, |F| <= 4 voxels, like an upsampled coarse flow
        ax = torch.linspace(0, 6.28318, S, device=dev)
        f = torch.stack([4 * torch.sin(ax).view(1, S, 1, 1) * torch.cos(ax).view(1, 1, S, 1).expand(B, S, S, S),
                         3 * torch.cos(ax * 2).view(1, 1, 1, S).expand(B, S, S, S) + 0 * ax.view(1, S, 1, 1),
                         2 * torch.sin(ax * 3).view(1, 1, S, 1) * torch.sin(ax).view(1, 1, 1, S).expand(B, S, S, S)], 1).contiguous()
    print("flow kind:", kind, tuple(f.shape))
    f.requires_grad_()
    G = torch.randn(B, 1, S, S, S, device=dev)
    nvox = B * S ** 3
    t = timeit(lambda: ops.warp3d(x, f.detach()))
    print("warp3d fwd  %dx%d^3: %.3f ms  %.1f GB/s (20 B/voxel)" % (B, S, t, nvox * 20 / t / 1e6))
    out = ops.warp3d(x, f)
    t = timeit(lambda: torch.autograd.grad(out, [f], G, retain_graph=True))
    print("warp3d bwd(flow) : %.3f ms  %.1f GB/s (32 B/voxel)" % (t, nvox * 32 / t / 1e6))
    xg = x.clone().requires_grad_()
    out = ops.warp3d(xg, f)
    t = timeit(lambda: torch.autograd.grad(out, [xg, f], G, retain_graph=True))
    print("warp3d bwd(in+flow): %.3f ms  %.1f GB/s (36 B/voxel + memset)" % (t, nvox * 36 / t / 1e6))
    # ATen for scale (same box): grid_sample on a precomputed grid
    grid = torch.rand(B, S, S, S, 3, device=dev) * 2 - 1
    t = timeit(lambda: torch.nn.functional.grid_sample(x, grid, mode='bilinear', padding_mode='border', align_corners=True))
    print("ATen grid_sample 3d fwd (grid precomputed): %.3f ms" % t)
    a = torch.empty(B * 5, S, S, S, device=dev); b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    print("copy %.0f MB: %.3f ms  %.1f GB/s (r+w)" % (a.numel() * 4 / 1e6, t, a.numel() * 8 / t / 1e6))


if __name__ == "__main__":
    main()
