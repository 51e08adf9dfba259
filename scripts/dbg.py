import torch, sys
sys.path.insert(0,'.')
from opticalflowscivis_amd import ops
g = torch.Generator().manual_seed(8)
for shape in [(2,1,12,20,16),(2,1,12,320),(2,3,17,23)]:
    a=torch.rand(shape,generator=g); b=torch.rand(shape,generator=g)
    for mode in (0,1,2,3):
        v = ops.robust_loss(a.cuda(), b.cuda(), None, mode, 1.0, 1e-6, 0, "mean")
        v2 = ops.robust_loss(a.cuda(), b.cuda(), None, mode, 1.0, 1e-6, 0, "mean")
        print(shape, mode, float(v), float(v2), float((a-b).abs().mean()))
