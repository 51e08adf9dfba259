"""Micro-benchmark + checksums of the trilinear-warp family at the bench shapes (GPU box only).

    python scripts/w3bench.py [S=256] [smooth|noise|zero]

Prints per entry point: ms per launch (HIP events, 20 launches), algorithmic GB/s (the byte counts of DESIGN.md §4)
and an fp64 checksum + CRC of every output, so that two builds can be compared for bit-identity
(scripts/gpu/* run it on the product and on a previous library through FLOWSCI_HIP_LIBRARY)."""
import os
import sys
import zlib

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflowscivis_amd import ops  # noqa: E402


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


def make_flow(kind, B, S, dev, ch=6, amp=1.0):
    if kind == "noise":
        return (torch.rand(B, ch, S, S, S, device=dev) * 2 - 1) * 4 * amp
    if kind == "zero":
        return torch.zeros(B, ch, S, S, S, device=dev)
    if kind == "small":  # the flows of a freshly initialised IFNet: a fraction of a voxel, smooth
        amp = amp * 0.1
    ax = torch.linspace(0, 6.28318, S, device=dev)
    a = 4 * torch.sin(ax).view(1, S, 1, 1) * torch.cos(ax).view(1, 1, S, 1).expand(B, S, S, S)
    b = 3 * torch.cos(ax * 2).view(1, 1, 1, S).expand(B, S, S, S) + 0 * ax.view(1, S, 1, 1)
    c = 2 * torch.sin(ax * 3).view(1, 1, S, 1) * torch.sin(ax).view(1, 1, 1, S).expand(B, S, S, S)
    return (torch.stack([a, b, c, -a, c, b][:ch], 1) * amp).contiguous()


def sig(*ts):
    out = []
    for t in ts:
        t = t.detach()
        out.append("%.9e/%08x" % (float(t.double().sum()), zlib.crc32(t.contiguous().cpu().numpy().tobytes())))
    return " ".join(out)


def main():
    dev = "cuda:0"
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    kind = sys.argv[2] if len(sys.argv) > 2 else "smooth"
    only = sys.argv[3] if len(sys.argv) > 3 else ""
    B = 2
    torch.manual_seed(0)
    i0 = torch.rand(B, 1, S, S, S, device=dev)
    i1 = torch.rand(B, 1, S, S, S, device=dev)
    nvox = B * S ** 3
    gb = lambda bytes_per_vox, ms: nvox * bytes_per_vox / ms / 1e6
    lib = os.environ.get("FLOWSCI_HIP_LIBRARY", "product")
    print("library:", lib, "| flow:", kind, "| %d x %d^3" % (B, S))

    # plain pair: forward 2 x 20 B/voxel, backward (flow gradient only) 2 x 32 B/voxel
    # (only = fwd | bwd | acc3: that launch alone is timed -- one kernel per PMC run; plain = fwd + bwd; acc = + acc3)
    f = make_flow(kind, B, S, dev).requires_grad_()
    G0, G1 = torch.randn_like(i0), torch.randn_like(i1)
    o0, o1 = ops.warp_pair(i0, i1, f)
    if only in ("", "plain", "acc", "fwd"):
        t = timeit(lambda: ops.warp_pair(i0, i1, f.detach()))
        print("fs_warp3d_pair_fwd            %.4f ms %6.0f GB/s  %s" % (t, gb(40, t), sig(o0, o1)))
    if only in ("", "plain", "acc", "bwd"):
        t = timeit(lambda: torch.autograd.grad([o0, o1], [f], [G0, G1], retain_graph=True))
        (gf,) = torch.autograd.grad([o0, o1], [f], [G0, G1], retain_graph=True)
        print("fs_warp3d_pair_bwd            %.4f ms %6.0f GB/s  %s" % (t, gb(64, t), sig(gf)))
        del gf
    if only in ("plain", "fwd", "bwd"):
        return
    # three addends (block 2 of the step): + 3 x 24 B/voxel read
    w0, w1, (fa, fb, fc) = ops.warp_pair_acc(i0, i1, f)
    A = [torch.randn_like(f) for _ in range(3)]
    run = lambda: torch.autograd.grad([w0, w1, fa, fb, fc], [f], [G0, G1] + A, retain_graph=True)
    t = timeit(run)
    (gf,) = run()
    print("fs_warp3d_pair_bwd_acc3       %.4f ms %6.0f GB/s  %s" % (t, gb(64 + 72, t), sig(gf)))
    del w0, w1, fa, fb, fc, gf, o0, o1
    if only in ("acc", "acc3"):
        return
    # fused up-sample + accumulate + warp: factor 2 with a running flow, factor 4 without
    for factor, has_prev in ((2, True), (4, False)):
        s = S // factor
        delta = (make_flow(kind, B, s, dev, amp=1.0 / factor)).requires_grad_()
        prev = make_flow(kind, B, S, dev, amp=0.5).requires_grad_() if has_prev else None
        run = lambda: ops.upsample_warp_pair(i0, i1, delta.detach(), None if prev is None else prev.detach(), factor)
        t = timeit(run)
        (fa, fb, fc), w0, w1 = ops.upsample_warp_pair(i0, i1, delta, prev, factor)
        bpv = 2 * 4 + 2 * 4 + 24 + (24 if has_prev else 0) + 24.0 / factor ** 3
        print("fs_upsample_warp3d_pair_fwd x%d %.4f ms %6.0f GB/s  %s" % (factor, t, gb(bpv, t), sig(fa, w0, w1)))
        ins = [delta] + ([prev] if has_prev else [])
        gr = [G0, G1] + [torch.randn_like(fa) for _ in range(3)]
        run = lambda: torch.autograd.grad([w0, w1, fa, fb, fc], ins, gr, retain_graph=True)
        t = timeit(run)
        g = run()
        print("fs_upsample_warp3d_pair_bwd3 x%d %.4f ms %6.0f GB/s  %s" % (factor, t, gb(64 + 72 + 24.0 / factor ** 3, t), sig(*g)))
        del fa, fb, fc, w0, w1, g
    # the resizes either side
    for factor in (2, 4):
        s = S // factor
        small = make_flow(kind, B, s, dev)
        prev = make_flow(kind, B, S, dev, amp=0.5)
        t = timeit(lambda: ops.upsample3d_scale_add(small, prev, factor, float(factor)))
        y = ops.upsample3d_scale_add(small, prev, factor, float(factor))
        print("fs_upsample3d_scale_add x%d     %.4f ms %6.0f GB/s  %s" % (factor, t, gb(48 + 24.0 / factor ** 3, t), sig(y)))
        x = make_flow(kind, B, S, dev)
        t = timeit(lambda: ops.interpolate3d(x, 1.0 / factor, 1.0 / factor))
        y = ops.interpolate3d(x, 1.0 / factor, 1.0 / factor)
        print("fs_downsample3d_fwd /%d         %.4f ms %6.0f GB/s  %s" % (factor, t, gb(24 + 24.0 / factor ** 3, t), sig(y)))


if __name__ == "__main__":
    main()
