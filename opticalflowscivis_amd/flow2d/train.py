"""Drop-in for Flow-2D/train.py.  `python -m opticalflowscivis_amd.flow2d.train --dataset droplet2d
--mode train`; multi-GPU: `python -m torch.distributed.run --nproc-per-node N ...`."""
import argparse

from ..trainer import add_common_args, run
from .model.RIFE import Model

if __name__ == "__main__":
    p = add_common_args(argparse.ArgumentParser(), 2)
    p.add_argument('--exp', default=1, type=int)
    args = p.parse_args()
    assert args.dataset is not None
    run(args, Model, 2)
