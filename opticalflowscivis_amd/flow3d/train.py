"""Drop-in for Flow-3D/train.py.  `python -m opticalflowscivis_amd.flow3d.train --dataset droplet3d
--mode train [--size 64]`; multi-GPU: `python -m torch.distributed.run --nproc-per-node N ...`."""
import argparse

from ..trainer import add_common_args, run
from .model.RIFE import Model

if __name__ == "__main__":
    args = add_common_args(argparse.ArgumentParser(), 3).parse_args()
    assert args.dataset is not None
    run(args, Model, 3)
