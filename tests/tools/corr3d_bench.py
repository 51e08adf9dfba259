"""fs_corr3d_{fwd,bwd} at the BASELINE config-4 feature shape [2, 32, 32, 32, 32], md = 2 and md = 4 (GPU box
only): time per launch (HIP events around back-to-back C-ABI launches, no Python autograd in the loop),
algorithmic GB/s against the 8 TB/s HBM roofline (4 (2C + (2md+1)^3) B/voxel forward,
4 (4C + (2md+1)^3) backward), TFLOP/s, and the max error against the CPU oracle (tests/ only import it;
here it is the checker).  `--profile`: a few launches only, for rocprofv3 --kernel-trace / --pmc runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from opticalflowscivis_amd import _lib

L = _lib.lib()
profile = "--profile" in sys.argv
N = 3 if profile else 30


def timed(fn, n=N):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(B, C, S, md):
    g = torch.Generator().manual_seed(md)
    f1 = torch.randn(B, C, S, S, S, generator=g).cuda()
    f2 = torch.randn(B, C, S, S, S, generator=g).cuda()
    nd = 2 * md + 1
    out = torch.empty(B, nd ** 3, S, S, S, device="cuda")
    gout = torch.randn(B, nd ** 3, S, S, S, generator=g).cuda()
    g1, g2 = torch.empty_like(f1), torch.empty_like(f2)
    st = torch.cuda.current_stream().cuda_stream
    fwd = lambda: _lib.check(L.fs_corr3d_fwd(f1.data_ptr(), f2.data_ptr(), out.data_ptr(), B, C, S, S, S, md, st), "fwd")
    bwd = lambda: _lib.check(L.fs_corr3d_bwd(f1.data_ptr(), f2.data_ptr(), gout.data_ptr(), g1.data_ptr(),
                                             g2.data_ptr(), B, C, S, S, S, md, st), "bwd")
    vox = B * S ** 3
    tf, tb = timed(fwd), timed(bwd)
    bf, bb = 4 * (2 * C + nd ** 3) * vox, 4 * (4 * C + nd ** 3) * vox
    ff, fb = 2 * nd ** 3 * C * vox, 4 * nd ** 3 * C * vox
    err = ""
    if not profile and S <= 32:
        from oracle.corr import corr3d_closed
        a, b = f1[:1, :, :12].cpu().requires_grad_(), f2[:1, :, :12].cpu().requires_grad_()
        o2 = torch.empty(1, nd ** 3, 12, S, S, device="cuda")
        fa, fb2 = f1[:1, :, :12].contiguous(), f2[:1, :, :12].contiguous()
        _lib.check(L.fs_corr3d_fwd(fa.data_ptr(), fb2.data_ptr(), o2.data_ptr(), 1, C, 12, S, S, md, st), "fwd")
        err = "  max err vs oracle (D=12 slab) %.1e" % float((o2.cpu() - corr3d_closed(a, b, md).detach()).abs().max())
    print("corr3d [%d,%d,%d^3] md=%d  fwd %.4f ms  %7.1f GB/s (%.3f of 8 TB/s)  %5.1f TFLOP/s | bwd %.4f ms  %7.1f GB/s "
          "(%.3f)  %5.1f TFLOP/s%s" % (B, C, S, md, tf, bf / tf / 1e6, bf / tf / 1e6 / 8000, ff / tf / 1e9, tb,
                                       bb / tb / 1e6, bb / tb / 1e6 / 8000, fb / tb / 1e9, err), flush=True)


case(2, 32, 32, 2)
case(2, 32, 32, 4)
if not profile:
    case(2, 64, 32, 4)
    case(2, 32, 64, 2)
