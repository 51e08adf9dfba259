"""Overfit a handful of synthetic volumes for a few hundred steps: loss must stay finite and fall (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opticalflowscivis_amd.flow3d.model.RIFE import Model
from opticalflowscivis_amd.data import synthetic

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 3e-4
every = int(sys.argv[4]) if len(sys.argv) > 4 else 50
torch.manual_seed(0)
m = Model(local_rank=-1, device="cuda:0")
data = synthetic.droplet3d_batch(4, S, seed=5, device="cuda:0")
imgs, gt = data[:, :2].contiguous(), data[:, 2:3].contiguous()
for i in range(steps + 1):
    pred, info = m.update(imgs, gt, learning_rate=lr, training=True)
    if i % every == 0 or not torch.isfinite(info["loss_G"]):
        print(i, "loss_G %.5f l1 %.5f tea %.5f distill %.5f psnr %.2f" % (
            float(info["loss_G"].detach()), float(info["loss_l1"].detach()), float(info["loss_tea"].detach()),
            float(info["loss_distill"].detach()), synthetic.psnr(pred.detach().cpu(), gt.cpu())), flush=True)
        if not torch.isfinite(info["loss_G"]):
            bad = [n for n, p in m.flownet.named_parameters() if not torch.isfinite(p).all()]
            print("non-finite parameters:", bad[:8], len(bad))
            fl = info["flow"]; print("flow finite:", bool(torch.isfinite(fl).all()), "max", float(fl.nan_to_num().abs().max()))
            break
