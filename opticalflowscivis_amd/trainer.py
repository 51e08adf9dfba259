"""Training / evaluation loop shared by the Flow-2D and Flow-3D `train.py` entry points.

Mirrors Flow-2D/train.py:70-553 and Flow-3D/train.py:72-478 (same flags, LR schedule, per-epoch
evaluation with PSNR, rank-0 checkpointing, one barrier per epoch) with two deliberate changes:
  * data: seeded synthetic stand-ins (data/synthetic.py) -- the reference's pickles are not in
    its tree and there is no network;
  * sharding: `DistributedSampler(shuffle=True)` + `set_epoch` is ON.  The reference left it
    commented out (Flow-3D/train.py:82-83,139; Flow-2D/train.py:88-89,137), so under
    torch.distributed every rank trained on the same samples.
"""
import math
import os
import time

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler

from .data import synthetic


class SyntheticTriplets(Dataset):
    """Triplets (img0, img1, gt) generated on demand; item i depends only on (seed, i)."""

    def __init__(self, kind, n, size, seed=1234):
        self.kind, self.n, self.size, self.seed = kind, n, size, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return self.item(i, "cpu")

    def item(self, i, device):
        """Triplet i generated on `device`: the shape parameters come from the seeded CPU generator either way, the
        volume itself is evaluated where it is asked for (same elementwise fp32 operations: same values)."""
        s = self.seed + 7919 * i
        if self.kind == "droplet3d":
            return synthetic.droplet3d_batch(1, self.size[0], seed=s, device=device)[0]
        if self.kind == "5jets3d":
            return synthetic.jets3d_batch(1, self.size[0], seed=s, device=device)[0]
        if self.kind == "droplet2d":
            h, w = self.size
            r = (max(4, h // 8), max(8, h // 4))
            return synthetic.droplet2d_batch(1, h, w, seed=s, device=device, radius=r)[0]
        raise ValueError("no synthetic generator for dataset %r" % self.kind)


class DeviceTripletLoader:
    """DataLoader stand-in for synthetic data: the same batches (same sampler, same `drop_last` rule, same item
    values) as `DataLoader(SyntheticTriplets(...))`, but every triplet is generated ON the GPU.  A 256^3 triplet is
    201 MB: at the bench's 18.6 pairs/s one rank would need 3.7 GB/s of single-threaded host generation plus the
    H2D copy (the reference feeds its 64^3 pickles with 8 worker processes, Flow-3D/train.py:84); generated on the
    device it costs ~1 ms of HBM-bound elementwise kernels per sample and no PCIe traffic."""

    def __init__(self, dataset, batch_size, device, sampler=None, shuffle=False, drop_last=False, seed=0):
        self.dataset, self.batch_size, self.device = dataset, batch_size, device
        self.sampler, self.shuffle, self.drop_last = sampler, shuffle, drop_last
        self._gen = torch.Generator().manual_seed(seed)

    def __len__(self):
        n = len(self.sampler) if self.sampler is not None else len(self.dataset)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        if self.sampler is not None:
            order = list(iter(self.sampler))
        elif self.shuffle:
            order = torch.randperm(len(self.dataset), generator=self._gen).tolist()
        else:
            order = list(range(len(self.dataset)))
        for k in range(len(self)):
            idx = order[k * self.batch_size:(k + 1) * self.batch_size]
            yield torch.stack([self.dataset.item(i, self.device) for i in idx])


class HostCachedLoader:
    """The reference's data arrangement at volume sizes: the whole (small) training set lives in host memory -- its
    `load_data` un-pickles every triplet into one numpy array (Flow-3D/load_datasets.py:47-48) -- and a batch is a gather of
    B samples.  At 256^3 a triplet is 201 MB, so the gather goes straight into one of two PINNED staging buffers (no
    per-sample collation, no worker processes copying through shared memory): `DevicePrefetcher` then moves the buffer
    to the GPU on its side stream while the previous step computes.  Same batches (sampler, `drop_last`, item values) as
    `DataLoader(SyntheticTriplets(...))`; the samples are generated once, on `device`, and parked on the host."""

    def __init__(self, dataset, batch_size, device, sampler=None, shuffle=False, drop_last=False, seed=0):
        self.batch_size, self.sampler, self.shuffle, self.drop_last = batch_size, sampler, shuffle, drop_last
        self._gen = torch.Generator().manual_seed(seed)
        first = dataset.item(0, device)
        self.data = torch.empty((len(dataset),) + tuple(first.shape), dtype=first.dtype).pin_memory()
        self.data[0].copy_(first)
        for i in range(1, len(dataset)):
            self.data[i].copy_(dataset.item(i, device))
        self.stage = [torch.empty((batch_size,) + tuple(first.shape), dtype=first.dtype).pin_memory() for _ in range(3)]
        self._copied = [None, None, None]  # per staging buffer: the event behind its last host-to-device copy
        self._last = 0

    def copy_issued(self, event):
        """DevicePrefetcher: the copy of the buffer yielded last has been enqueued; `event` completes with it."""
        self._copied[self._last] = event

    def __len__(self):
        n = len(self.sampler) if self.sampler is not None else self.data.shape[0]
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        if self.sampler is not None:
            order = list(iter(self.sampler))
        elif self.shuffle:
            order = torch.randperm(self.data.shape[0], generator=self._gen).tolist()
        else:
            order = list(range(self.data.shape[0]))
        for k in range(len(self)):
            idx = order[k * self.batch_size:(k + 1) * self.batch_size]
            # three staging buffers: the one being filled, the one in flight to the GPU, the one the step may still read from
            if self._copied[k % 3] is not None:
                self._copied[k % 3].synchronize()  # (its previous contents have left for the GPU: three batches ago)
            buf = self.stage[k % 3][:len(idx)]
            for j, i in enumerate(idx):
                buf[j].copy_(self.data[i])
            self._last = k % 3
            yield buf


class DevicePrefetcher:
    """Host-resident data (any DataLoader with pin_memory=True): the H2D copy of batch k+1 runs on a side stream
    while step k computes -- what UPFlow's `tools.data_prefetcher` does in the reference (UPFlow/utils/tools.py:
    177-260); the Flow-2D / Flow-3D loops of the reference copy synchronously in front of every step
    (Flow-3D/train.py:144)."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, device
        self.stream = torch.cuda.Stream(device=device)

    def __len__(self):
        return len(self.loader)

    def _load(self, it):
        try:
            host = next(it)
        except StopIteration:
            return None
        with torch.cuda.stream(self.stream):
            dev = host.to(self.device, non_blocking=True)
            if hasattr(self.loader, "copy_issued"):  # a loader that reuses pinned staging buffers
                ev = torch.cuda.Event()
                ev.record(self.stream)
                self.loader.copy_issued(ev)
            return dev

    def __iter__(self):
        it = iter(self.loader)
        nxt = self._load(it)
        while nxt is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self.stream)
            batch = nxt
            batch.record_stream(cur)  # its memory may not be reused by the side stream while the step reads it
            nxt = self._load(it)
            yield batch


def get_learning_rate(step, total_steps):
    """Flow-3D/train.py:50-56: linear warm-up to 3e-4 over 2000 steps, then cosine to 3e-5."""
    if step < 2000:
        return 3e-4 * (step / 2000.)
    mul = np.cos((step - 2000) / (total_steps - 2000.) * math.pi) * 0.5 + 0.5
    return (3e-4 - 3e-5) * mul + 3e-5


def psnr01(pred, gt):
    """-10 log10(mean((gt - pred)^2)) on [0,1] data (Flow-3D/train.py:385)."""
    return -10 * math.log10(max(float(torch.mean((gt - pred) * (gt - pred))), 1e-20))


def evaluate(model, val_data, nd, dataset, device):
    """Per-epoch validation: loss means and PSNR of student / teacher (train.py `evaluate`)."""
    losses, psnr, psnr_tea = [], [], []
    for data in val_data:
        data = data.to(device, non_blocking=True)
        imgs, gt = data[:, :2], data[:, 2:3]
        with torch.no_grad():
            if nd == 3:
                pred, info = model.update(imgs, gt, training=False)
            else:
                pred, info = model.update(imgs, gt, dataset, training=False)
        losses.append(float(info['loss_G']))
        sp = tuple(min(a, b) for a, b in zip(gt.shape[2:], pred.shape[2:]))
        cut = (slice(None), slice(None)) + tuple(slice(0, s) for s in sp)
        for j in range(gt.shape[0]):
            psnr.append(psnr01(pred[cut][j], gt[cut][j]))
            psnr_tea.append(psnr01(info['merged_tea'][cut][j], gt[cut][j]))
    return float(np.mean(losses)), float(np.mean(psnr)), float(np.mean(psnr_tea))


def run(args, Model, nd):
    world = int(os.environ.get("WORLD_SIZE", args.world_size))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", args.local_rank))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs a ROCm GPU: the HIP hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", world_size=world, rank=rank)  # RCCL over xGMI
    seed = 1234
    np.random.seed(seed)
    torch.manual_seed(seed)
    model = Model(local_rank if distributed else -1, device=device)
    size = tuple(args.size) if nd == 2 else (args.size[0],) * 3
    train_set = SyntheticTriplets(args.dataset, args.samples, size, seed)
    val_set = SyntheticTriplets(args.dataset, max(args.batch_size, args.samples // 8), size, seed + 10 ** 6)
    sampler = DistributedSampler(train_set, num_replicas=world, rank=rank, shuffle=True) if distributed else None
    if args.host_data and getattr(args, "host_cache", False):
        # the reference's arrangement at volume sizes: the training set resident in host memory, batches gathered into
        # pinned staging buffers, the H2D copy of the next batch on a side stream under the current step
        train_data = DevicePrefetcher(HostCachedLoader(train_set, args.batch_size, device, sampler=sampler, shuffle=True,
                                                       drop_last=True, seed=seed), device)
        val_data = DeviceTripletLoader(val_set, args.batch_size, device)
    elif args.host_data:
        # the reference's arrangement (Flow-3D/train.py:84: DataLoader workers + pinned memory), with the H2D copy of
        # the next batch overlapped with the current step
        train_data = DevicePrefetcher(DataLoader(train_set, batch_size=args.batch_size, num_workers=args.workers,
                                                 pin_memory=True, drop_last=True, sampler=sampler,
                                                 shuffle=(sampler is None)), device)
        val_data = DevicePrefetcher(DataLoader(val_set, batch_size=args.batch_size, num_workers=args.workers,
                                               pin_memory=True), device)
    else:
        train_data = DeviceTripletLoader(train_set, args.batch_size, device, sampler=sampler, shuffle=True,
                                         drop_last=True, seed=seed)
        val_data = DeviceTripletLoader(val_set, args.batch_size, device)
    steps_per_epoch = len(train_data)
    log_path = args.log_path
    os.makedirs(log_path, exist_ok=True)
    model_name = args.model_name
    try:
        model.load_model(model_name, log_path)
        if rank == 0:
            print("loaded", model_name)
    except (FileNotFoundError, RuntimeError, KeyError):
        if rank == 0:
            print("no weights found, training from scratch")

    if args.mode != "train":
        loss, p, pt = evaluate(model, val_data, nd, args.dataset, device)
        if rank == 0:
            print("test: loss_G %.4e  PSNR %.2f dB  (teacher %.2f dB)" % (loss, p, pt))
        if distributed:
            dist.destroy_process_group()
        return

    step, best = 0, None
    total = args.epoch * steps_per_epoch
    # step driver (round 5): at N = 1 the whole step -- forward, losses, backward, AdamW -- is captured once into a HIP
    # graph and replayed per batch (Model.graphed_update: 0.4-0.5 ms of an 83 ms Flow-3D step at 256^3, a quarter of
    # the launch-bound Flow-2D step); --eager keeps per-launch dispatch; under DDP the step is always eager
    use_graph = (not distributed) and (not getattr(args, "eager", False))
    graph_step = None
    for epoch in range(args.epoch):
        if sampler is not None:
            sampler.set_epoch(epoch)
        t0 = te = time.time()
        for i, data in enumerate(train_data):
            data = data.to(device, non_blocking=True)  # (already there on both data paths)
            imgs, gt = data[:, :2], data[:, 2:3]
            lr = get_learning_rate(step, max(total, 2001)) * world / 4  # train.py:167
            if use_graph:
                if graph_step is None:
                    graph_step = model.graphed_update(imgs, gt, **({} if nd == 3 else {"dataset": args.dataset}))
                pred, info = graph_step(imgs, gt, lr)
            elif nd == 3:
                pred, info = model.update(imgs, gt, lr, training=True)
            else:
                pred, info = model.update(imgs, gt, args.dataset, lr, training=True)
            if rank == 0 and (i % args.log_every == 0):
                print('epoch:{}/{} {}/{} time:{:.2f} loss_G:{:.4e}'.format(
                    epoch, args.epoch, i, steps_per_epoch, time.time() - t0, float(info['loss_G'].detach())))
                t0 = time.time()
            step += 1
        torch.cuda.synchronize(device)
        if rank == 0 and steps_per_epoch:
            dt = time.time() - te
            print("epoch %d train loop: %d steps in %.2f s = %.1f ms/step = %.2f pairs/s per rank (%s data, %s)" % (
                epoch, steps_per_epoch, dt, dt / steps_per_epoch * 1e3, steps_per_epoch * args.batch_size / dt,
                "host" if args.host_data else "device-generated", "hip-graph replay" if use_graph else "eager launches"))
        loss, p, pt = evaluate(model, val_data, nd, args.dataset, device)
        if rank == 0:
            print("eval epoch %d: loss_G %.4e  PSNR %.2f dB  (teacher %.2f dB)" % (epoch, loss, p, pt))
            if nd == 2 or best is None or loss <= best:  # 2-D saves every epoch, 3-D on improvement
                best = loss if best is None else min(best, loss)
                model.save_model(model_name, log_path, 0)
        if distributed:
            dist.barrier()
    if distributed:
        dist.destroy_process_group()


def add_common_args(parser, nd):
    parser.add_argument('--epoch', default=100, type=int)
    parser.add_argument('--batch_size', default=1 if nd == 3 else 16, type=int, help='minibatch size')
    parser.add_argument('--local_rank', default=0, type=int, help='local rank')
    parser.add_argument('--world_size', default=1, type=int, help='world size')
    parser.add_argument('--dataset', dest='dataset', type=str, default=None)
    parser.add_argument('--mode', dest='mode', type=str, default='test')
    # additions (synthetic data, no hard-coded checkpoint names)
    parser.add_argument('--size', type=int, nargs='+', default=[64] if nd == 3 else [160, 224])
    parser.add_argument('--samples', type=int, default=64, help='synthetic training triplets')
    parser.add_argument('--workers', type=int, default=0, help='DataLoader worker processes (with --host_data)')
    parser.add_argument('--host_data', action='store_true',
                        help='generate the synthetic triplets on the host and feed them through a DataLoader + '
                             'pinned-memory prefetcher (the reference\'s arrangement) instead of on the GPU')
    parser.add_argument('--host_cache', action='store_true',
                        help='with --host_data: keep the whole training set in (pinned) host memory, as the reference\'s '
                             'load_data does, and gather batches from it (no per-sample generation in the loader)')
    parser.add_argument('--eager', action='store_true',
                        help='N = 1: eager launches per step instead of replaying the step from one HIP graph')
    parser.add_argument('--log_every', type=int, default=10)
    parser.add_argument('--log_path', default='train_log')
    parser.add_argument('--model_name', default='flownet.pkl')
    return parser
