"""Which eager ATen ops remain in one Flow-3D train step, and who calls them (torch profiler, GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from opticalflowscivis_amd.flow3d.model.RIFE import Model
from opticalflowscivis_amd.data import synthetic

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(1234)
m = Model(local_rank=-1, device="cuda:0")
data = synthetic.droplet3d_batch(2, S, seed=1234).cuda()
imgs, gt = data[:, :2], data[:, 2:3]
for _ in range(2):
    m.update(imgs, gt, learning_rate=1e-4, training=True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    m.update(imgs, gt, learning_rate=1e-4, training=True)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=40,
                                                          max_shapes_column_width=60))
print("==== by stack: copy_/contiguous/add_/mul/cat ====")
for e in sorted(prof.key_averages(group_by_stack_n=12), key=lambda e: -e.device_time_total):
    if e.key in ("aten::copy_", "aten::add_", "aten::mul", "aten::cat", "aten::add", "aten::mul_", "aten::fill_", "aten::zero_", "aten::sum") and e.device_time_total > 100:
        st = [f for f in e.stack if "opticalflowscivis_amd" in f or "bench" in f][:4]
        print("%-12s n=%3d  %8.3f ms  %s" % (e.key, e.count, e.device_time_total / 1e3, " <- ".join(x.split("/")[-1] for x in st)))
print("==== all aten ops by device time ====")
for e in sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)[:40]:
    if e.self_device_time_total > 50:
        print("%-60s n=%4d  self %8.3f ms" % (e.key[:60], e.count, e.self_device_time_total / 1e3))
print("==== aten ops by input shape ====")
for e in sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.self_device_time_total):
    if e.key.startswith("aten::") and e.self_device_time_total > 20:
        print("%-28s n=%4d  self %8.3f ms  %s" % (e.key, e.count, e.self_device_time_total / 1e3, str(e.input_shapes)[:150]))
print("==== parents of the large copy_/add_/fill_ calls ====")
import collections
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::add_", "aten::fill_", "aten::cat", "aten::sum") and e.device_time_total > 30:
        chain, p = [], e.cpu_parent
        while p is not None and len(chain) < 4:
            chain.append(p.name[:40]); p = p.cpu_parent
        cnt[(e.name, str(e.input_shapes)[:60], " <- ".join(chain))] += 1
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(v, k)
