"""Does replaying ONE HIP graph instance back to back block the host?  (profiles/r05_train_256.txt: the host-data path is slower
under graph replay than with eager launches.)  Times the host side of graph.replay() with and without 50 ms of host work between
replays, at the bench step (GPU box only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opticalflowscivis_amd.data import synthetic
from opticalflowscivis_amd.flow3d.model.RIFE import Model

dev = torch.device("cuda:0")
torch.manual_seed(1234)
m = Model(local_rank=-1, device=dev)
data = synthetic.droplet3d_batch(2, int(sys.argv[1]) if len(sys.argv) > 1 else 256, seed=1234, device=dev)
imgs, gt = data[:, :2].contiguous(), data[:, 2:3].contiguous()
step = m.graphed_update(imgs, gt)
for _ in range(3):
    step(imgs, gt, 1e-6)
torch.cuda.synchronize()
for host_ms in (0, 50):
    host, t0 = [], time.perf_counter()
    for i in range(8):
        a = time.perf_counter()
        step(imgs, gt, 1e-6)
        host.append((time.perf_counter() - a) * 1e3)
        if host_ms:
            b = time.perf_counter()
            while (time.perf_counter() - b) * 1e3 < host_ms:
                pass
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) * 1e3 / 8
    print("host work between replays %2d ms: host time inside step() per replay: %s ms; wall %.1f ms per step" % (
        host_ms, " ".join("%.1f" % h for h in host), tot))
