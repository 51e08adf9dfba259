"""One rank of the two-rank HIP-path DDP test (tests/test_gpu_ddp.py); started as a FRESH child process, never by
re-executing a process that has touched the GPU.  All ranks share cuda:0 (the GPU box has one card; RCCL refuses two
ranks on one device, so the process group runs on gloo, which stages CUDA tensors through the host): what is
exercised is the product's N > 1 path on the real kernels -- DDP hooks, gradients as bucket views, the three-alias
flow gradients of `ops._WarpPairAcc`, the fused AdamW on bucket views, `DistributedSampler` sharding.
Reference: Flow-3D/train.py:82-84,139,490; Flow-3D/model/RIFE.py:33-34."""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--samples", type=int, default=4)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(a.port), RANK=str(a.rank), WORLD_SIZE=str(a.world))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if a.world > 1:
        dist.init_process_group("gloo", rank=a.rank, world_size=a.world)
    from opticalflowscivis_amd.flow3d.model.RIFE import Model
    from opticalflowscivis_amd.trainer import SyntheticTriplets
    from torch.utils.data.distributed import DistributedSampler
    torch.manual_seed(1234)
    m = Model(local_rank=0 if a.world > 1 else -1, device=dev)
    ds = SyntheticTriplets("droplet3d", a.samples, (a.size,), seed=1234)
    # the same sampler on both sides: the single process (world 1) replays the two shards as ONE global batch
    shards = []
    for r in range(max(a.world, 2)):
        s = DistributedSampler(ds, num_replicas=max(a.world, 2), rank=r, shuffle=True)
        s.set_epoch(0)
        shards.append(list(iter(s)))
    rec = {"rank": a.rank, "world": a.world, "shards": shards, "losses": []}
    grads = None
    for step in range(a.steps):
        if a.world > 1:
            idx = [shards[a.rank][step]]
        else:
            idx = [sh[step] for sh in shards]
        data = torch.stack([ds[i] for i in idx]).to(dev)
        pred, info = m.update(data[:, :2], data[:, 2:3], learning_rate=1e-4, training=True)
        rec["losses"].append([float(info[k].detach()) for k in ("loss_l1", "loss_tea", "loss_distill", "loss_G")])
        if step == 0:
            grads = [p.grad.detach().cpu().clone() for p in m.flownet.parameters()]
    torch.cuda.synchronize()
    rec["param_sums"] = [float(p.detach().double().sum()) for p in m.flownet.parameters()]
    rec["param_abs_sums"] = [float(p.detach().double().abs().sum()) for p in m.flownet.parameters()]
    torch.save(grads, a.out + ".grads.pt")
    with open(a.out + ".json", "w") as f:
        json.dump(rec, f)
    if a.world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
