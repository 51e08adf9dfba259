"""Can the UPFlow C3 train step (150x450, batch 32, census on) be replayed from one HIP graph?  Eager vs graph time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from opticalflowscivis_amd.data import synthetic
from opticalflowscivis_amd.upflow.scripts.simple_train import Loss_manager, Trainer

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
conf = Trainer.Config(exp_dir="/tmp/upflow_bench")
conf.net_params = dict(conf.net_params, photo_loss_census_weight=1)
torch.manual_seed(0)
tr = Trainer(conf, device=dev)
opt = torch.optim.Adam(tr.net.parameters(), lr=1e-4, weight_decay=1e-4, amsgrad=True, capturable=True)
pairs = synthetic.vortex2d_pairs(B, 150, 450, seed=0, device=dev)
im1, im2 = pairs[:, 0].contiguous(), pairs[:, 1].contiguous()
lm = Loss_manager()
out = {}


def step():
    o = tr.net({'im1': im1, 'im2': im2, 'if_loss': True})
    loss = lm.compute_loss(o['loss_dict'], B)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    out["loss"] = loss.detach()


def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
print("eager %.2f ms/step, loss %.6f" % (timed(step, 10), float(out["loss"])), flush=True)
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    step()
print("captured", flush=True)
print("graph replay %.2f ms/step, loss %.6f" % (timed(g.replay, 10), float(out["loss"])), flush=True)
