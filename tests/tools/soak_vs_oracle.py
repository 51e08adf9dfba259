"""Loss trajectory of the product (GPU) and of the CPU oracle from the same weights / data, constant lr.
Test infrastructure (imports oracle/); GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from opticalflowscivis_amd.flow3d.model.RIFE import Model
from opticalflowscivis_amd.data import synthetic
from oracle.ifnet_ref import ModelRef

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 80
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 3e-5
torch.manual_seed(0)
m = Model(local_rank=-1, device="cuda:0")
o = ModelRef(3)
o.flownet.load_state_dict({k: v.cpu() for k, v in m.flownet.state_dict().items()})
data = synthetic.droplet3d_batch(4, S, seed=5)
imgs, gt = data[:, :2].contiguous(), data[:, 2:3].contiguous()
gi, gg = imgs.cuda(), gt.cuda()
torch.set_num_threads(16)
for i in range(steps + 1):
    _, pi = m.update(gi, gg, learning_rate=lr, training=True)
    _, oi = o.update(imgs, gt, learning_rate=lr)
    if i % 10 == 0:
        print(i, "gpu loss_G %.5f distill %.5f | oracle loss_G %.5f distill %.5f" % (
            float(pi["loss_G"].detach()), float(pi["loss_distill"].detach()), float(oi["loss_G"].detach()),
            float(oi["loss_distill"].detach())), flush=True)
