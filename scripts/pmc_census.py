"""A few launches of fs_census_dist_{fwd,bwd} at the C3 image pair (32 x 3 x 150 x 450) for
`rocprofv3 --pmc <group> -- python scripts/pmc_census.py` (scripts/pmc_summary.py <csv> census prints the means)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opticalflowscivis_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
B, H, W = 32, 150, 450
im1, im2 = torch.rand(B, 3, H, W, device="cuda"), torch.rand(B, 3, H, W, device="cuda")
dist = torch.empty(B, 1, H, W, device="cuda")
G = torch.randn_like(dist)
g1, g2 = torch.empty_like(im1), torch.empty_like(im2)
p = lambda x: x.data_ptr()
for _ in range(3):
    _lib.check(L.fs_census_dist_fwd(p(im1), p(im2), p(dist), B, H, W, 3, st), "fwd")
    _lib.check(L.fs_census_dist_bwd(p(im1), p(im2), p(G), p(g1), p(g2), B, H, W, 3, st), "bwd")
torch.cuda.synchronize()
