"""Throughput of the other BASELINE configs (parity-test cases, not bench lines): C2 Flow-2D Droplet
160x224 B=16 and C3 UPFlow 150x450 B=32 train steps on one MI355X, with the oracle / stock CPU step
beside them.  GPU box only.  usage: bench_configs.py [c2|c3|all]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from opticalflowscivis_amd.data import synthetic


def timed(fn, steps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def c2():
    from opticalflowscivis_amd.flow2d.model.RIFE import Model
    from oracle.ifnet_ref import ModelRef
    B = 16
    data = synthetic.droplet2d_batch(B, 160, 224, seed=1234)
    torch.manual_seed(1234)
    m = Model(local_rank=-1, device="cuda:0")
    d = data.cuda()
    dt = timed(lambda: m.update(d[:, :2], d[:, 2:3], "droplet2d", learning_rate=1e-5), 20, 5)
    print("C2 Flow-2D 160x224 B=16: GPU %.2f ms/step = %.0f frame-pairs/s" % (dt * 1e3, B / dt), flush=True)
    torch.manual_seed(1234)
    o = ModelRef(2)
    torch.set_num_threads(16)
    o.update(data[:, :2], data[:, 2:3], learning_rate=1e-5)
    t0 = time.perf_counter()
    for _ in range(3):
        o.update(data[:, :2], data[:, 2:3], learning_rate=1e-5)
    dc = (time.perf_counter() - t0) / 3
    print("C2 CPU oracle (16 threads): %.0f ms/step = %.1f frame-pairs/s" % (dc * 1e3, B / dc), flush=True)


def c3():
    from opticalflowscivis_amd.upflow.scripts.simple_train import Trainer, Loss_manager
    B = 32
    conf = Trainer.Config(exp_dir="/tmp/upflow_bench")
    conf.net_params = dict(conf.net_params, photo_loss_census_weight=1)
    torch.manual_seed(0)
    tr = Trainer(conf, device="cuda:0")
    opt = torch.optim.Adam(tr.net.parameters(), lr=1e-4, weight_decay=1e-4, amsgrad=True)
    pairs = synthetic.vortex2d_pairs(B, 150, 450, seed=0, device="cuda:0")
    lm = Loss_manager()

    def step():
        out = tr.net({'im1': pairs[:, 0], 'im2': pairs[:, 1], 'if_loss': True})
        loss = lm.compute_loss(out['loss_dict'], B)
        opt.zero_grad()
        loss.backward()
        opt.step()
    dt = timed(step, 10, 3)
    print("C3 UPFlow 150x450 B=32 (census on): GPU %.1f ms/step = %.0f frame-pairs/s" % (dt * 1e3, B / dt), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("c2", "all"):
        c2()
    if which in ("c3", "all"):
        c3()
