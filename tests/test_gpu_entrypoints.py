"""-m gpu: the drop-in entry points run as fresh child processes -- `flow{2,3}d/train.py` (train, then test
from the checkpoint it wrote), `flow{2,3}d/inference_img.py`, and `bench.py` on the N > 1 code path (RCCL
process group + DDP wrapper) rehearsed with one rank.  Reference: Flow-3D/train.py:72-232,344-412,479-587,
Flow-3D/inference_img.py, Flow-3D/model/RIFE.py:33-34."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable] + args, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout)
    assert r.returncode == 0, (r.stdout.decode()[-1500:], r.stderr.decode()[-3000:])
    return r.stdout.decode(), r.stderr.decode()


@pytest.mark.parametrize("nd", [3, 2])
def test_train_then_test_from_checkpoint(tmp_path, nd):
    mod = "opticalflowscivis_amd.flow%dd.train" % nd
    size = ["32"] if nd == 3 else ["64", "96"]
    ds = "droplet3d" if nd == 3 else "droplet2d"
    common = ["-m", mod, "--dataset", ds, "--size"] + size + ["--samples", "4", "--batch_size", "2",
                                                              "--log_path", str(tmp_path), "--log_every", "1"]
    out, _ = _run(common + ["--mode", "train", "--epoch", "1"])
    assert "no weights found" in out and "epoch:0/1" in out
    m = re.search(r"eval epoch 0: loss_G ([-+0-9.e]+)\s+PSNR ([-+0-9.]+) dB", out)
    assert m and np.isfinite(float(m.group(1))) and np.isfinite(float(m.group(2)))
    ck = os.path.join(str(tmp_path), "flownet.pkl")
    assert os.path.exists(ck)
    sd = torch.load(ck, map_location="cpu")
    assert len(sd) == 160 and all(k.startswith("module.") for k in sd)  # the reference's on-disk format
    assert all(torch.isfinite(v).all() for v in sd.values())
    out2, _ = _run(common + ["--mode", "test"])
    assert "loaded flownet.pkl" in out2
    m2 = re.search(r"test: loss_G ([-+0-9.e]+)\s+PSNR ([-+0-9.]+) dB", out2)
    assert m2 and np.isfinite(float(m2.group(2)))
    # the test run evaluates the saved weights on the same validation set as the training run's last eval
    assert abs(float(m2.group(2)) - float(m.group(2))) < 0.02
    assert abs(float(m2.group(1)) - float(m.group(1))) < 1e-3 * abs(float(m.group(1))) + 1e-6


@pytest.mark.parametrize("nd", [3, 2])
def test_inference_img(tmp_path, nd):
    from opticalflowscivis_amd.data import synthetic
    if nd == 3:
        d = synthetic.droplet3d_batch(1, 40, seed=3)  # 40 is padded to 64 like the reference pads to /32
    else:
        d = synthetic.droplet2d_batch(1, 72, 100, seed=3, radius=(8, 16))
    a, b = str(tmp_path / "a.npy"), str(tmp_path / "b.npy")
    np.save(a, d[0, 0].numpy())
    np.save(b, d[0, 1].numpy())
    out_dir = str(tmp_path / "out")
    _run(["-m", "opticalflowscivis_amd.flow%dd.inference_img" % nd, "--img", a, b, "--exp", "2", "--model",
          str(tmp_path), "--out", out_dir])
    frames = [np.load(os.path.join(out_dir, "img%d.npy" % i)) for i in range(5)]  # 2**exp + 1 frames
    for f in frames:
        assert f.shape == tuple(d.shape[2:]) and np.isfinite(f).all()
    np.testing.assert_array_equal(frames[0], d[0, 0].numpy())
    np.testing.assert_array_equal(frames[4], d[0, 1].numpy())
    # with weights from a training run of the same package the checkpoint is picked up
    from opticalflowscivis_amd.flow3d.model.RIFE import Model as M3
    from opticalflowscivis_amd.flow2d.model.RIFE import Model as M2
    torch.manual_seed(3)
    m = (M3 if nd == 3 else M2)(local_rank=-1, device="cuda:0")
    m.save_model("flownet.pkl", str(tmp_path))
    out_dir2 = str(tmp_path / "out2")
    so, _ = _run(["-m", "opticalflowscivis_amd.flow%dd.inference_img" % nd, "--img", a, b, "--exp", "1", "--model",
                  str(tmp_path), "--out", out_dir2])
    assert "random-init" not in so
    mid = np.load(os.path.join(out_dir2, "img1.npy"))
    m.eval()
    with torch.no_grad():
        pad = [0, 24] * 3 if nd == 3 else [0, 28, 0, 24]
        x = torch.nn.functional.pad(d[:, 0:1].to("cuda:0"), pad)
        y = torch.nn.functional.pad(d[:, 1:2].to("cuda:0"), pad)
        want = m.inference(x, y)[0]
        want = want[2] if isinstance(want, list) else want
    cut = (0, 0) + tuple(slice(0, n) for n in d.shape[2:])
    assert float(np.abs(mid - want[cut].cpu().numpy()).max()) < 1e-5


def _bench(extra_env, port):
    env = dict(extra_env, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out, err = _run([os.path.join(ROOT, "bench.py"), "--size", "32", "--batch", "1", "--steps", "2", "--warmup", "1",
                     "--no-cpu-baseline"], env=env)
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_ddp_code_path_with_one_rank():
    """The N > 1 path of bench.py (init_process_group('nccl') = RCCL, DDP-wrapped model, barrier + max over
    ranks) with world_size 1; the child process sets the environment before any GPU call.  Same JSON
    contract, same loss as the plain run."""
    plain = _bench({}, 29641)
    ddp = _bench({"FLOWSCI_BENCH_FORCE_DDP": "1"}, 29642)
    for d in (plain, ddp):
        assert d["n_gpus"] == 1 and d["config"]["parallelism"] == "dp1" and d["scaling"] == "weak"
        assert d["value"] > 0 and d["steps"] == 2
    assert abs(ddp["loss_G"] - plain["loss_G"]) <= 2e-4 * abs(plain["loss_G"])


def test_device_generated_batches_equal_the_dataloaders():
    """the trainer's default data path generates every synthetic triplet on the GPU; the batches
    are the ones `DataLoader(SyntheticTriplets)` delivers (same sampler, same values bit for bit), and the
    pinned-memory prefetcher of the host path hands out the loader's batches unchanged."""
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    from opticalflowscivis_amd.trainer import DevicePrefetcher, DeviceTripletLoader, SyntheticTriplets
    dev = torch.device("cuda:0")
    for kind, size in (("droplet3d", (32,)), ("5jets3d", (24,)), ("droplet2d", (64, 96))):
        ds = SyntheticTriplets(kind, 7, size, seed=1234)
        sampler = DistributedSampler(ds, num_replicas=2, rank=1, shuffle=True)
        sampler.set_epoch(3)
        host = list(DataLoader(ds, batch_size=2, drop_last=True, sampler=sampler))
        devl = DeviceTripletLoader(ds, 2, dev, sampler=sampler, drop_last=True)
        got = list(devl)
        assert len(got) == len(host) == len(devl) == 2
        for a, b in zip(got, host):
            assert a.is_cuda and a.shape == b.shape
            if kind == "5jets3d":  # exp() differs by an ulp between the host's and the GPU's libm
                assert float((a.cpu() - b).abs().max()) < 2e-6
            else:
                assert torch.equal(a.cpu(), b)
        pre = list(DevicePrefetcher(DataLoader(ds, batch_size=2, drop_last=True, sampler=sampler, pin_memory=True), dev))
        torch.cuda.synchronize()
        assert len(pre) == len(host) and all(torch.equal(a.cpu(), b) for a, b in zip(pre, host))
        # round 5: the training set resident in pinned host memory, batches gathered into staging buffers (--host_cache)
        from opticalflowscivis_amd.trainer import HostCachedLoader
        sampler.set_epoch(3)
        cached = list(DevicePrefetcher(HostCachedLoader(ds, 2, dev, sampler=sampler, drop_last=True), dev))
        torch.cuda.synchronize()
        assert len(cached) == len(host)
        for a, b in zip(cached, host):
            assert a.is_cuda and (float((a.cpu() - b).abs().max()) < 2e-6 if kind == "5jets3d" else torch.equal(a.cpu(), b))
    # without a sampler: a seeded permutation, all samples once, last short batch kept unless drop_last
    ds = SyntheticTriplets("droplet3d", 5, (16,), seed=1)
    assert [b.shape[0] for b in DeviceTripletLoader(ds, 2, dev, shuffle=True)] == [2, 2, 1]
    assert [b.shape[0] for b in DeviceTripletLoader(ds, 2, dev, shuffle=True, drop_last=True)] == [2, 2]


def test_train_host_data_path(tmp_path):
    """`--host_data --workers 2`: the reference's arrangement (DataLoader workers, pinned memory) behind the
    side-stream prefetcher trains to the same first-epoch evaluation as the device-generated default."""
    common = ["-m", "opticalflowscivis_amd.flow3d.train", "--dataset", "droplet3d", "--size", "32", "--samples", "4",
              "--batch_size", "2", "--log_every", "1", "--mode", "train", "--epoch", "1"]
    a, _ = _run(common + ["--log_path", str(tmp_path / "a")])
    b, _ = _run(common + ["--log_path", str(tmp_path / "b"), "--host_data", "--workers", "2"])
    pa = re.search(r"eval epoch 0: loss_G ([-+0-9.e]+)\s+PSNR ([-+0-9.]+) dB", a)
    pb = re.search(r"eval epoch 0: loss_G ([-+0-9.e]+)\s+PSNR ([-+0-9.]+) dB", b)
    assert pa and pb and "device-generated data" in a and "host data" in b
    # same weights (seed), same validation set; the training ORDER differs (torch.randperm vs DataLoader's sampler
    # stream), so the two runs agree as two 2-step trainings do, not bit for bit
    assert abs(float(pa.group(2)) - float(pb.group(2))) < 0.5
