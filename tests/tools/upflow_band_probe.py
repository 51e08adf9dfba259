"""GPU box: how far from the reference's CPU golden (tests/golden/upflow_e2e.npz) do (a) the HIP path and (b) the
reference's stock torch ops on this GPU land, run to run, with and without torch.backends.cudnn.deterministic?
Data for the band of tests/test_gpu_e2e.py::test_upflow_matches_reference_golden.  usage: upflow_band_probe.py [runs]"""
import contextlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_DEBUG_CONV_WINOGRAD", "0")
from opticalflowscivis_amd.upflow.model.upflow import UPFlow_net  # noqa: E402
from oracle.upflow_port import stock_ops  # noqa: E402

g = dict(np.load(os.path.join(ROOT, "tests", "golden", "upflow_e2e.npz")))
DEV = "cuda:0"


def run(stock):
    conf = UPFlow_net.config()
    conf.update({'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
                 'norm_moments_across_images': False, 'photo_loss_census_weight': 1,
                 'multi_scale_distillation_weight': 1})
    torch.manual_seed(0)
    net = conf().to(DEV)
    with (stock_ops() if stock else contextlib.nullcontext()):
        out = net({'im1': torch.from_numpy(g["im1"]), 'im2': torch.from_numpy(g["im2"]), 'if_loss': True})
        keys = [str(k) for k in g["loss_keys"]]
        got = np.array([float(out['loss_dict'][k].detach()) for k in keys])
        sum(out['loss_dict'][k] for k in keys).backward()
    ref_f = torch.from_numpy(g["flow_f_out"])
    scale = float(ref_f.abs().max())
    err = (out['flow_f_out'].detach().cpu() - ref_f).abs()
    gsum = np.array([float(p.grad.detach().double().abs().sum()) if p.grad is not None else 0.0 for p in net.parameters()])
    rel = np.abs(gsum - g["grad_abs_sums"]) / (np.abs(g["grad_abs_sums"]) + 1e-3)
    occ = float((out['occ_fw'].cpu() != torch.from_numpy(g["occ_fw"])).float().mean())
    return dict(med=float(err.median()) / scale, p99=float(err.flatten().quantile(0.99)) / scale,
                mx=float(err.max()) / scale, loss=float(np.max(np.abs(got - g["losses"]) / g["losses"])), occ=occ,
                gmed=float(np.median(rel)), gmax=float(rel.max()))


n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for det in (False, True):
    torch.backends.cudnn.deterministic = det
    for stock in (True, False):
        for i in range(n):
            r = run(stock)
            print("deterministic=%d %s run %d: " % (det, "stock" if stock else "hip  ", i) +
                  " ".join("%s=%.2e" % kv for kv in r.items()), flush=True)
