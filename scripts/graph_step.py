"""Experiment: the whole Flow-3D train step captured into one HIP graph (1 GPU).  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.optim import AdamW
from opticalflowscivis_amd.flow3d.model.RIFE import Model
from opticalflowscivis_amd.data import synthetic

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(1234)
m = Model(local_rank=-1, device="cuda:0")
m.optimG = AdamW(m.flownet.parameters(), lr=torch.tensor(1e-6, device="cuda"), weight_decay=1e-3, capturable=True)
m._set_lr = lambda lr: None  # lr lives in the optimiser's device tensor
data = synthetic.droplet3d_batch(2, S, seed=1234, device="cuda:0")
imgs, gt = data[:, :2].contiguous(), data[:, 2:3].contiguous()


def timed(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        m.update(imgs, gt, learning_rate=1e-6, training=True)
torch.cuda.current_stream().wait_stream(s)
print("eager: %.2f ms/step" % timed(lambda: m.update(imgs, gt, learning_rate=1e-6, training=True)), flush=True)
g = torch.cuda.CUDAGraph()
m.optimG.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    pred, info = m.update(imgs, gt, learning_rate=1e-6, training=True)
print("captured", flush=True)
print("graph: %.2f ms/step" % timed(g.replay), flush=True)
print("loss_G", float(info["loss_G"]))
