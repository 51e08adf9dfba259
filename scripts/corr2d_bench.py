"""fs_corr2d at the five UPFlow pyramid levels of BASELINE config C3 (B = 32, 150 x 450 input): GPU time per
launch measured with HIP events around back-to-back C-ABI launches (no Python autograd / allocator in the
loop -- scripts/bench_kernels_2d.py times the Python op, whose ~50 us per call of host work hides every kernel
shorter than that), single-direction and both-directions-per-launch forms, plain and normalize_features-folded.
Algorithmic bytes 4 (2C + 81) h w per sample and direction forward, 4 (4C + 81) backward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opticalflowscivis_amd import _lib

L = _lib.lib()
profile = "--profile" in sys.argv
N = 3 if profile else 200


def timed(fn, n=N):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


B, md = 32, 4
st = torch.cuda.current_stream().cuda_stream
print("level (C,h,w)      | single fwd      bwd       | pair fwd (2 dirs)  bwd       | pair+norm fwd   bwd   [ms per launch; GB/s algorithmic]")
for C, h, w in [(196, 3, 8), (128, 5, 15), (96, 10, 29), (64, 19, 57), (32, 38, 113)]:
    t = [torch.randn(B, C, h, w, device="cuda") for _ in range(4)]
    oa, ob = torch.empty(B, 81, h, w, device="cuda"), torch.empty(B, 81, h, w, device="cuda")
    ga, gb = torch.randn_like(oa), torch.randn_like(ob)
    g = [torch.empty_like(t[0]) for _ in range(4)]
    stats = torch.empty(4, B * C, 2, device="cuda")
    _lib.check(L.fs_plane_moments4(*[x.data_ptr() for x in t], stats.data_ptr(), B * C, h * w, st), "moments")
    p = lambda x: x.data_ptr()
    f1 = timed(lambda: L.fs_corr2d_fwd(p(t[0]), p(t[1]), p(oa), B, C, h, w, md, st))
    b1 = timed(lambda: L.fs_corr2d_bwd(p(t[0]), p(t[1]), p(ga), p(g[0]), p(g[1]), B, C, h, w, md, st))
    f2 = timed(lambda: L.fs_corr2d_pair_fwd(p(t[0]), p(t[1]), p(t[2]), p(t[3]), None, p(oa), p(ob), B, C, h, w, md, st))
    b2 = timed(lambda: L.fs_corr2d_pair_bwd(p(t[0]), p(t[1]), p(t[2]), p(t[3]), None, p(ga), p(gb), p(g[0]), p(g[1]),
                                            p(g[2]), p(g[3]), B, C, h, w, md, st))
    f3 = timed(lambda: L.fs_corr2d_pair_fwd(p(t[0]), p(t[1]), p(t[2]), p(t[3]), p(stats), p(oa), p(ob), B, C, h, w,
                                            md, st))
    b3 = timed(lambda: L.fs_corr2d_pair_bwd(p(t[0]), p(t[1]), p(t[2]), p(t[3]), p(stats), p(ga), p(gb), p(g[0]),
                                            p(g[1]), p(g[2]), p(g[3]), B, C, h, w, md, st))
    bf, bb = 4 * (2 * C + 81) * h * w * B, 4 * (4 * C + 81) * h * w * B
    gbs = lambda nb, ms: nb / ms / 1e6
    print("(%3d,%2d,%3d)  | %.4f %7.1f  %.4f %7.1f | %.4f %7.1f  %.4f %7.1f | %.4f %7.1f  %.4f %7.1f" % (
        C, h, w, f1, gbs(bf, f1), b1, gbs(bb, b1), f2, gbs(2 * bf, f2), b2, gbs(2 * bb, b2), f3, gbs(2 * bf, f3), b3,
        gbs(2 * bb, b3)), flush=True)
