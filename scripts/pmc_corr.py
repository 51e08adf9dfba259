"""A few launches of the tiled correlation kernels at the finest C3 level (B = 32, C = 32, 38 x 113, md = 4) and of
fs_corr3d at [2,32,32^3] md = 4, for `rocprofv3 --pmc <group> -- python scripts/pmc_corr.py` (one counter group per
run; scripts/pmc_summary.py prints the per-kernel means -> profiles/rNN_corr_pmc_counters.txt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opticalflowscivis_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
B, C, h, w, md = 32, 32, 38, 113, 4
t = [torch.randn(B, C, h, w, device="cuda") for _ in range(2)]
out = torch.empty(B, 81, h, w, device="cuda")
g = [torch.empty_like(t[0]) for _ in range(2)]
G = torch.randn_like(out)
p = lambda x: x.data_ptr()
for _ in range(3):
    _lib.check(L.fs_corr2d_fwd(p(t[0]), p(t[1]), p(out), B, C, h, w, md, st), "fwd")
    _lib.check(L.fs_corr2d_bwd(p(t[0]), p(t[1]), p(G), p(g[0]), p(g[1]), B, C, h, w, md, st), "bwd")
f = [torch.randn(2, 32, 32, 32, 32, device="cuda") for _ in range(2)]
o3 = torch.empty(2, 729, 32, 32, 32, device="cuda")
g3 = [torch.empty_like(f[0]) for _ in range(2)]
G3 = torch.randn_like(o3)
for _ in range(3):
    _lib.check(L.fs_corr3d_fwd(p(f[0]), p(f[1]), p(o3), 2, 32, 32, 32, 32, 4, st), "fwd3")
    _lib.check(L.fs_corr3d_bwd(p(f[0]), p(f[1]), p(G3), p(g3[0]), p(g3[1]), 2, 32, 32, 32, 32, 4, st), "bwd3")
torch.cuda.synchronize()
