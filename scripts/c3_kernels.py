"""C2 / C3 steps with the table of their hot-path kernels (HIP events per launch): bench.py's own legs, stand-alone."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
for name, fn in (("C2", bench.config_c2), ("C3", bench.config_c3)):
    if len(sys.argv) > 1 and name.lower() not in sys.argv[1:]:
        continue
    r = fn(dev, 10, 3)
    print(name, "%.2f ms/step" % r["ms_per_step"], "launches", r["flowsci_launches_per_step"])
    for k, v in sorted(r["hot_path_kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
        print("   %-24s %6.1f launches  %8.4f ms/step  %8s GB/s  %s of HBM" % (k, v["launches_per_step"], v["ms_per_step"], v["algo_GBps"], v["frac_of_hbm_peak"]))
