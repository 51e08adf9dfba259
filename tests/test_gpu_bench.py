"""-m gpu: the bench.py contract on a tiny workload -- ONE JSON line on stdout with the required keys."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_at_cpu_size"}


def test_bench_json_contract():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--size", "32", "--batch", "1", "--steps", "2",
           "--warmup", "1", "--cpu-size", "32", "48", "--no-configs"]
    r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600,
                       env=dict(os.environ, FLOWSCI_BENCH_NO_CPU_256="1"))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines  # exactly one line on stdout (RCCL / MIOpen chatter must not leak)
    d = json.loads(lines[0])
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] - 1 * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"] + 1e-9
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source"} <= set(rf) and rf["bound"] in ("hbm", "mfma")
    assert rf["traffic"] is None and "256^3" in rf["traffic_source"]  # PMC traffic is recorded for the BASELINE size only
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    rh = d["roofline_hbm"]
    # (round 5: the three-addend backward; at this size the gather kernels take it, so the record carries the entry point's name)
    assert rh["bound"] == "hbm" and rh["entry_point"] == "fs_warp3d_pair_bwd_acc3" and 0 < rh["frac"] < 1
    assert rh["kernel"] in ("fs_warp3d_pair_bwd_acc3", "warp3d_rc_kernel<true, 4, 5, 0>")
    cb = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] in ("port", "reference")
    assert cb["value"] > 0 and cb["cores"] >= 1
    # one bounded sample per --cpu-size edge, then the record of the metric's own size (one measured 256^3 step where memory
    # and the time budget allow; switched off here: it takes minutes)
    assert [x["size"] for x in cb["samples"]] == [32, 48, 256]
    assert cb["samples"][-1]["measured"] is False and "switched off" in cb["samples"][-1]["reason"]
    # the bench line's own parity witness: the GPU model against the oracle on the CPU samples' batches
    pw = d["parity_at_cpu_size"]
    assert {"loss_gpu", "loss_oracle", "rel", "tolerance_rel", "sizes"} <= set(pw)
    assert pw["rel"] <= pw["tolerance_rel"] == 5e-4
    assert abs(pw["loss_gpu"] - pw["loss_oracle"]) <= 5e-4 * abs(pw["loss_oracle"])
    # round 5: the N = 1 step is replayed from one HIP graph; the per-launch events come from the eager pass behind it
    assert rf["measured"].startswith("HIP events on the launch stream inside the eager pass")
    # round 4: `frac` is a fraction (executed flops or algorithmic bytes over the peak), the line says what could have
    # changed dispatch, and which driver stepped the model; the bench-size witness belongs to the 256^3 workload only
    assert 0 < rf["frac"] <= 1 and "entry_point" in rf
    assert d["switches"] == {"library": "product", "env": dict({k: v for k, v in os.environ.items() if k.startswith("FLOWSCI_")},
                                                               FLOWSCI_BENCH_NO_CPU_256="1")}
    assert d["step_driver"].startswith("hip-graph replay")
    sd = d["step_drivers"]
    assert sd["hip_graph_replay_ms_per_step"] > 0 and sd["eager_ms_per_step"] > 0
    assert abs(sd["hip_graph_replay_ms_per_step"] - d["ms_per_step"]) < 1e-9
    assert "parity_at_bench_size" not in d


def test_bench_line_carries_the_other_single_gpu_configs():
    """C2 (Flow-2D 160x224 B=16) and C3 (UPFlow 150x450 B=32, census on) are timed in the same
    run, as extra keys; the top-level metric stays config 4's."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--size", "32", "--batch", "1", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--config-steps", "3"]
    r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["metric"].startswith("volume-pairs/sec") and d["unit"] == "volume-pairs/s"
    c2, c3 = d["configs"]["C2"], d["configs"]["C3"]
    for c, B in ((c2, 16), (c3, 32)):
        assert c["unit"] == "frame-pairs/s" and c["ms_per_step"] > 0
        assert abs(c["pairs_per_s"] - B * 1000.0 / c["ms_per_step"]) < 1e-6 * c["pairs_per_s"]
        assert c["dominant_hot_path_kernel"] in c["hot_path_kernels"]
        k = c["hot_path_kernels"][c["dominant_hot_path_kernel"]]
        assert k["algo_GBps"] > 0 and 0 < k["frac_of_hbm_peak"] < 1
    assert "fs_warp2d_pair_fwd" in c2["hot_path_kernels"] and "fs_robust_sum" in c2["hot_path_kernels"]
    assert {"fs_corr2d_pair_fwd", "fs_corr2d_pair_bwd", "fs_census_dist_fwd", "fs_census_dist_bwd",
            "fs_warp2d_fwd", "fs_warp2d_bwd"} <= set(c3["hot_path_kernels"])
    # the launch-bound C2 step replayed from one HIP graph is not slower than eager
    assert c2["graph_replay"]["ms_per_step"] < c2["eager"]["ms_per_step"] * 1.05
    assert c2["ms_per_step"] == c2["graph_replay"]["ms_per_step"] and c2["step_driver"].startswith("hip-graph replay")
