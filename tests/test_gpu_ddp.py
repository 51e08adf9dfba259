"""-m gpu: more than one rank on the real HIP path.

* two fresh rank processes on the box's one GPU (gloo) run the product's DDP-wrapped Flow-3D model on the HIP ops:
  disjoint `DistributedSampler` shards, replicas identical after every step, and gradients / weights equal to ONE
  process stepping on the two-sample global batch (reference: Flow-3D/train.py:82-84,139; model/RIFE.py:33-34);
* `python bench.py --gpus 2` starts its own two ranks (shared-GPU rehearsal on one card; over RCCL when the box
  has two) and its JSON line carries the process-group witness."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "tools", "ddp_hip_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(world, size, out_prefix, steps=2):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, WORKER, "--rank", str(r), "--world", str(world), "--port", str(port),
                               "--size", str(size), "--steps", str(steps), "--out", "%s%d" % (out_prefix, r)],
                              cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    try:
        outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    finally:
        for p in procs:  # the exact children started above
            if p.poll() is None:
                p.kill()
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    recs = [json.load(open("%s%d.json" % (out_prefix, r))) for r in range(world)]
    grads = [torch.load("%s%d.grads.pt" % (out_prefix, r)) for r in range(world)]
    return recs, grads


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("size", [64])
def test_two_ranks_on_the_hip_path_equal_one_process_on_the_global_batch(tmp_path, size):
    two, g2 = _spawn(2, size, str(tmp_path / "ddp"))
    one, g1 = _spawn(1, size, str(tmp_path / "one"))
    # disjoint shards that cover the data set
    sh = two[0]["shards"]
    assert sh == two[1]["shards"] and sorted(sh[0] + sh[1]) == [0, 1, 2, 3] and not set(sh[0]) & set(sh[1])
    # replicas: the all-reduced gradients and the weights after two AdamW steps are the same numbers on both ranks
    for a, b in zip(g2[0], g2[1]):
        assert torch.equal(a, b)
    assert two[0]["param_sums"] == two[1]["param_sums"] and two[0]["param_abs_sums"] == two[1]["param_abs_sums"]
    # each rank's loss is its own sample's; their mean is the global batch's (every term is a batch mean)
    l2 = np.mean([r["losses"] for r in two], axis=0)
    np.testing.assert_allclose(l2[0], np.array(one[0]["losses"])[0], rtol=2e-5, atol=1e-8)
    # DDP's gradient average == the gradient of the global-batch loss, tensor by tensor (fp32 atomics in the
    # weight-gradient epilogue and a different batch decomposition: 1e-4 of each tensor's norm)
    worst = 0.0
    for a, b in zip(g2[0], g1[0]):
        den = float(b.double().norm())
        err = float((a.double() - b.double()).norm())
        worst = max(worst, err / max(den, 1e-12))
        assert err <= 1e-4 * den + 1e-9, (err, den)
    # ... and so are the weights after the second step (AdamW's first steps are ~ lr * sign(g): entries whose
    # gradient is within noise of 0 may move the other way, hence sums at 1e-4 of the absolute sum's step share)
    ps2, ps1 = np.array(two[0]["param_sums"]), np.array(one[0]["param_sums"])
    pabs = np.array(one[0]["param_abs_sums"])
    assert np.all(np.abs(ps2 - ps1) <= 1e-4 * pabs + 1e-6), float(np.max(np.abs(ps2 - ps1) / (pabs + 1e-6)))
    np.testing.assert_allclose(l2[1], np.array(one[0]["losses"])[1], rtol=2e-4, atol=1e-7)
    print("worst relative gradient difference DDP(2 ranks) vs global batch: %.2e" % worst)


def _bench(args, env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines  # rank 0's line only: the other ranks' stdout goes to stderr
    return json.loads(lines[0])


def _check_two_rank_line(d, backend):
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 2
    rk = d["ranks"]
    assert rk["world_size"] == 2 and rk["backend"] == backend and rk["launcher"] == "self"
    assert rk["allreduce_sum_of_rank_plus_1"] == rk["expected"] == 3.0
    assert sorted(p["rank"] for p in rk["per_rank"]) == [0, 1]
    assert len({p["pid"] for p in rk["per_rank"]}) == 2
    assert abs(d["value"] - 2 * 1 * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert rk["ms_per_step_min"] <= rk["ms_per_step_max"] <= d["ms_per_step"] * 1.5 + 5
    # the replicas stay in step: same weights, different samples -> different but finite losses
    assert all(np.isfinite(p["loss_G"]) for p in rk["per_rank"])
    assert "cpu_baseline" not in d and d["roofline"]["frac"] > 0


def test_bench_starts_its_own_two_ranks_on_one_card():
    d = _bench(["--gpus", "2", "--size", "32", "--batch", "1", "--steps", "2", "--warmup", "1"],
               {"FLOWSCI_BENCH_SHARE_GPU": "1"})
    _check_two_rank_line(d, "gloo")
    assert rk_devices(d) == 1


def rk_devices(d):
    return d["ranks"]["devices"]


def test_bench_starts_its_own_two_ranks_over_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the round-end scaling run covers N > 1 on an 8-GPU node)")
    d = _bench(["--gpus", "2", "--size", "32", "--batch", "1", "--steps", "2", "--warmup", "1"], {})
    _check_two_rank_line(d, "nccl")
    assert rk_devices(d) == 2


def test_bench_refuses_more_ranks_than_cards():
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and b"visible" in r.stderr and not r.stdout.strip()
