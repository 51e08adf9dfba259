"""A/B on one box: Flow-2D C2 step (B = 16, 160x224) with IFNet's (conv, PReLU) pairs as one autograd node (bias gradient
from fs_prelu_bwd) vs two stock nodes (ATen reduction per layer).  `python tests/tools/ab_c2.py` on the GPU box."""
import torch

import bench
from opticalflowscivis_amd import convgrad

dev = torch.device("cuda:0")
for rep in range(2):
    for fuse in (True, False):
        convgrad._FUSE_2D = fuse
        r = bench.config_c2(dev, 30, 5)
        print("fused pairs" if fuse else "stock pairs", "eager", round(r["ms_per_step"], 3), "ms  graph replay",
              round(r["graph_replay"]["ms_per_step"], 3), "ms", flush=True)
