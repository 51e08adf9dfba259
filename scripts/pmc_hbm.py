"""Two launches each of the HBM-bound hot-path kernels at the shapes of the 2 x 256^3 Flow-3D step, forward and backward,
for `rocprofv3 --pmc <group>` (scripts/pmc_hbm.sh): the trilinear warp pair, the fused up-sample + warp pair of the
scale-2 block, the running-flow accumulation, the x2 down-sampling, merge and the distillation terms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opticalflowscivis_amd import ops

torch.manual_seed(0)
B, S = 2, 256
dev = "cuda"
img0 = torch.rand(B, 1, S, S, S, device=dev).requires_grad_()
img1 = torch.rand(B, 1, S, S, S, device=dev).requires_grad_()
# smooth displacement fields of a few voxels, as the network produces them (a white-noise flow scatters every gather and
# is 3-8x slower: not the step's case)
smooth = lambda n, amp: torch.nn.functional.interpolate(torch.randn(B, 6, 8, 8, 8, device=dev) * amp, size=(n, n, n),
                                                        mode="trilinear", align_corners=False)
flow = smooth(S, 3.0).requires_grad_()
delta = smooth(S // 2, 1.0).requires_grad_()
mask = torch.randn(B, 1, S, S, S, device=dev).requires_grad_()
for _ in range(2):
    w0, w1 = ops.warp_pair(img0, img1, flow)                                   # fs_warp3d_pair_fwd
    # flow gradient only, as in the step (the image-gradient variant scatters with float atomics: 5.9 ms, API completeness)
    torch.autograd.grad((w0 * w0).sum() + (w1 * w1).sum(), [flow])             # fs_warp3d_pair_bwd
    (fa, fb, fc), u0, u1 = ops.upsample_warp_pair(img0.detach(), img1.detach(), delta, flow.detach(), 2)  # fs_upsample_warp3d_pair_fwd
    torch.autograd.grad((u0 * u0).sum() + (u1 * u1).sum() + fa.sum(), [delta])  # its backward (warp backward + x2 adjoint)
    a = ops.upsample3d_scale_add(delta, flow.detach(), 2, 2.0)                 # fs_upsample3d_scale_add
    d = ops.interpolate3d(flow, 0.5, 0.5)                                      # fs_downsample3d_fwd
    torch.autograd.grad(d.sum(), [flow])                                       # fs_interp3d_bwd
    m, _sig = ops.merge(w0.detach().requires_grad_(), w1.detach(), mask)       # fs_merge_fwd
    torch.autograd.grad((m * m).sum(), [mask])                                 # fs_merge_bwd
torch.cuda.synchronize()
