"""Every convolution launch of one 2 x 256^3 Flow-3D train step, grouped by (entry point, algorithmic flops): count, mean
ms, executed and direct-equivalent TFLOP/s.  `python tests/tools/conv_calls.py [size] [batch]` on the GPU box."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from opticalflowscivis_amd import ops
from opticalflowscivis_amd.data import synthetic
from opticalflowscivis_amd.flow3d.model.RIFE import Model

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = Model(local_rank=-1, device=dev)
data = synthetic.droplet3d_batch(B, S, seed=1234, device=dev)
imgs, gt = data[:, :2].contiguous(), data[:, 2:3].contiguous()
for _ in range(2):
    model.update(imgs, gt, learning_rate=1e-6, training=True)
torch.cuda.synchronize()
ops.enable_kernel_timing(True)
model.update(imgs, gt, learning_rate=1e-6, training=True)
torch.cuda.synchronize()
t = ops.kernel_timings()
ops.enable_kernel_timing(False)
tot = 0.0
for name in ("fs_conv3d_fwd", "fs_conv3d_tr", "fs_conv3d_wrw"):
    g = collections.defaultdict(list)
    for ms, nb, fe, fq in t.get(name, []):
        g[(fq, fe, nb)].append(ms)
    print(name)
    for (fq, fe, nb), v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
        m = sum(v) / len(v)
        tot += sum(v)
        print("  n=%2d  %7.3f ms  sum %6.2f ms  %7.2f GFLOP direct  %6.1f TFLOP/s executed  %6.1f direct-equivalent  %7.1f MB algorithmic" % (
            len(v), m, sum(v), fq / 1e9, fe / m / 1e9, fq / m / 1e9, nb / 1e6))
print("convolutions: %.1f ms" % tot)
