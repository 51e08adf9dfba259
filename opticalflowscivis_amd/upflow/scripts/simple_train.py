"""Drop-in for UPFlow/scripts/simple_train.py: `Loss_manager`, `Trainer` and a CLI.

    python -m opticalflowscivis_amd.upflow.scripts.simple_train --epochs 2 --batchsize 32

Data are seeded synthetic vortex-field pairs at the reference's pipedcylinder2d size (150 x 450,
grey replicated to 3 channels: BASELINE config C3); KITTI loaders / evaluation benches of the
reference are out of scope.  Optimiser and schedule follow simple_train.py:236-260
(Adam lr 1e-4, weight decay 1e-4, exponential LR decay gamma).  Unlike the reference's training body
(:206,309-310) exceptions are NOT swallowed.
"""
import argparse
import os
import time

import torch

from ...data import synthetic
from ..model.upflow import UPFlow_net


class Loss_manager:
    """simple_train.py:65-95: sums whichever terms the network produced.  The reference reads every term back with
    `.item()` as it is formed (three host synchronisations per step); here the running sums stay on the device -- one
    stack + one add per step -- and are read when `log_info()` is asked for them: same log line, a step without host
    synchronisation (and therefore capturable into a HIP graph)."""
    NAMES = ('photo_loss', 'smooth_loss', 'census_loss', 'msd_loss', 'eq_loss', 'oi_loss')

    def __init__(self):
        self.prepare_epoch()

    def prepare_epoch(self):
        self.names, self.acc, self.count = None, None, 0

    def compute_loss(self, loss_dict, batch_N):
        loss = 0
        names, vals = [], []
        for name in self.NAMES:
            v = loss_dict.get(name)
            if v is None:
                continue
            v = v.mean()
            names.append(name)
            vals.append(v.detach())
            loss = loss + v
        if vals:
            if self.names != names:  # first step of an epoch (or the set of terms changed: start over)
                self.names, self.count = names, 0
                self.acc = torch.zeros(len(names), dtype=torch.float64, device=vals[0].device)
            self.acc.add_(torch.stack(vals), alpha=batch_N)
        self.count += batch_N
        return loss

    @property
    def sums(self):
        """{term: sum over the epoch's samples} as Python floats (synchronises)."""
        if self.acc is None:
            return {}
        return dict(zip(self.names, self.acc.tolist()))

    def log_info(self):
        return " ".join("%s:%.4f" % (k, v / max(self.count, 1)) for k, v in self.sums.items())


class Trainer:
    class Config:
        def __init__(self, **kw):
            self.exp_dir = './demo_exp'
            self.batchsize = 8
            self.n_epoch = 1000
            self.batch_per_epoch = 5
            self.batch_per_print = 20
            self.lr = 1e-4
            self.weight_decay = 1e-4
            self.scheduler_gamma = 1
            self.size = (150, 450)
            self.model_name = 'upflow.pth'
            self.net_params = {'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
                               'norm_moments_across_images': False, 'if_froze_pwc': False,
                               'if_sgu_upsample': False}  # simple_train.py:321-329
            for k, v in kw.items():
                setattr(self, k, v)

    def __init__(self, conf, device="cuda"):
        self.conf, self.device = conf, torch.device(device)
        os.makedirs(conf.exp_dir, exist_ok=True)
        nc = UPFlow_net.config()
        nc.update(conf.net_params)
        self.net = nc()
        path = os.path.join(conf.exp_dir, conf.model_name)
        if os.path.exists(path):
            self.net.load_model(path, if_relax=True, if_print=False)
        self.net = self.net.to(self.device)

    def training(self):
        conf = self.conf
        opt = torch.optim.Adam(self.net.parameters(), lr=conf.lr, weight_decay=conf.weight_decay, amsgrad=True)
        sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=conf.scheduler_gamma)
        lm = Loss_manager()
        step = 0
        for epoch in range(conf.n_epoch):
            lm.prepare_epoch()
            self.net.train()
            t0 = time.time()
            for i in range(conf.batch_per_epoch):
                pairs = synthetic.vortex2d_pairs(conf.batchsize, conf.size[0], conf.size[1], seed=step,
                                                 device=self.device)
                out = self.net({'im1': pairs[:, 0], 'im2': pairs[:, 1], 'if_loss': True})
                loss = lm.compute_loss(out['loss_dict'], conf.batchsize)
                opt.zero_grad()
                loss.backward()
                opt.step()
                step += 1
                if step % conf.batch_per_print == 0:
                    print("epoch %d step %d %s (%.2f s)" % (epoch, step, lm.log_info(), time.time() - t0))
            sched.step()
            self.net.save_model(os.path.join(conf.exp_dir, conf.model_name))
            print("epoch %d done: %s" % (epoch, lm.log_info()))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument('--epochs', type=int, default=2)
    ap.add_argument('--batchsize', type=int, default=32)
    ap.add_argument('--batch_per_epoch', type=int, default=5)
    ap.add_argument('--census', type=float, default=1.0, help='photo_loss_census_weight')
    ap.add_argument('--exp_dir', default='./demo_exp')
    a = ap.parse_args()
    conf = Trainer.Config(n_epoch=a.epochs, batchsize=a.batchsize, batch_per_epoch=a.batch_per_epoch,
                          batch_per_print=1, exp_dir=a.exp_dir)
    conf.net_params = dict(conf.net_params, photo_loss_census_weight=a.census)
    Trainer(conf).training()
