#!/usr/bin/env python3
"""bench.py -- volume-pairs/s of one full unsupervised Flow-3D train step on MI355X.

Workload (BASELINE.json metric / config "Flow-3D Droplet 256^3 volumes ... batch 2, 1xMI355X"):
synthetic Droplet-3D triplets [B=2, 3, 256, 256, 256] per GPU, random-init RIFE IFNet-3D, one step =
forward (3 student blocks + teacher, 4 warp-pair launches) + L1/distillation losses + backward +
AdamW.  Warps, loss / merge epilogues, resizes and the 3-D convolutions (implicit GEMM on the fp32 matrix
cores) are this repo's HIP kernels behind the C-ABI; what is left to ATen is listed in DESIGN.md §5.

In the K timed steps only the launches of the dominant entry point (found in the warm-up steps: ~90 of ~540
launches per step) carry HIP events on the launch stream, so `roofline` is measured inside the timed region at no
visible cost; a short separate pass afterwards (same model, same batch) records every C-ABI launch for the
`kernels` table.  After
that, rank 0 of an N = 1 run times the CPU oracle on bounded samples (64^3 and 128^3) and runs the GPU
model on the same batches from the same seed: `parity_at_cpu_size` is the bench line's own parity witness
(the run exits non-zero when the losses disagree by more than 5e-4 relative).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus 8 --steps 5 --warmup 2          # starts its own 8 rank processes (see launch())
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, DDP over RCCL; each rank draws its own volume pairs (seed 1234+rank):
weak scaling, the only collective is DDP's gradient all-reduce.  Rank 0 prints ONE JSON line; its `ranks`
record is the witness that the process group really had N members (an all-reduce of rank+1, every rank's device
and its own step time).  Either launch form works: under torch.distributed.run the rank variables are already in
the environment; called bare with --gpus N > 1 the process starts N fresh children of itself BEFORE anything
touches the GPU (the reference does the same through torch.distributed.launch, Flow-3D/train.py:2,490).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 achievable)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32-input MFMA peak (same guide: v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (same guide: ~2.5 PF, v_mfma_f32_32x32x16_bf16 at 32 cycles per SIMD)

# Algorithmic (compulsory) HBM bytes per launch are supplied by ops.py next to every C-ABI call
# (DESIGN.md §4): e.g. one warp3d pair launch = 2 warps x 20 B/voxel forward (12 flow + 4 gather +
# 4 store) or 2 x 32 B/voxel backward (12 flow + 4 gather + 4 grad_out + 12 grad_flow; the images
# carry no gradient in training).


def roofline(dom, k, S, B):
    """Roofline record of a hand-written kernel (a kernel SYMBOL where ops.py can name it, else a C-ABI entry point):
    the fp32 implicit-GEMM convolutions are priced against the dense fp32 MFMA peak, every other kernel against HBM.
    For an MFMA-bound kernel `achieved` = the multiply-adds its matrix cores EXECUTE per second (x2 flops), so
    `frac` <= 1 by construction.  The Winograd-domain trunk kernels execute a third (forward / input gradient:
    F(2,3) along y x F(4,3) along x, csrc/convwino2d.hpp) or half (weight gradient: F(4,3) along x,
    csrc/convwrwwino4.hpp) of the direct convolution's 2 * out * Cin * 27 flops: that ratio is `algorithmic_speedup`,
    and `direct_equivalent_TFLOPps` (= achieved x it) is a speed-up, not a roofline figure."""
    traffic, src = pmc_traffic(dom, S, B)
    if "TFLOPps" in k and "s3_kernel" in dom:
        # fp32-accurate convolution on the bf16 matrix rate (csrc/convfwd_s3.hpp): six v_mfma_f32_32x32x16_bf16 products per
        # fp32 multiply-add are EXECUTED -- that rate against the dense bf16 MFMA peak; the useful fp32 rate beside it
        ex = k["TFLOPps"]
        de = k.get("TFLOPps_direct_equivalent", ex)
        return {"bound": "mfma", "kernel": dom, "achieved": ex, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ex / MFMA_BF16_PEAK_TFLOPS, 4), "useful_fp32_TFLOPps": de,
                "arithmetic": "fp32 operands as 3 bf16 pieces, 6 products, fp32 accumulate",
                "traffic": traffic, "traffic_source": src}
    if "TFLOPps" in k:
        ex = k["TFLOPps"]
        de = k.get("TFLOPps_direct_equivalent", ex)
        return {"bound": "mfma", "kernel": dom, "achieved": ex, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ex / MFMA_F32_PEAK_TFLOPS, 4), "algorithmic_speedup": round(de / ex, 3) if ex else None,
                "direct_equivalent_TFLOPps": de, "traffic": traffic, "traffic_source": src}
    return {"bound": "hbm", "kernel": dom, "achieved": k["algo_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(k["algo_GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": src}


def roofline_hbm(kern, ktimes, S, B):
    """The hot-path row the metric is named after against the HBM roof: the trilinear backward warp pair in the form that
    dominates the hot path -- the three-addend launch (fs_warp3d_pair_bwd_acc3: flow gradient + the three other
    gradients of the flow in one pass, 136 B per voxel pair) -- named by its kernel SYMBOL where ops.py can ask the library
    for it (fs_warp3d_kernel_id), with the PMC traffic recorded for that entry point."""
    for ep in ("fs_warp3d_pair_bwd_acc3", "fs_warp3d_pair_bwd"):
        if ep in kern:
            r = roofline(ep, kern[ep], S, B)
            syms = {x[4] for x in ktimes.get(ep, []) if len(x) > 4 and x[4]}
            r["entry_point"] = ep
            r["kernel"] = sorted(syms)[0] if len(syms) == 1 else ep
            r["launches_per_step"] = round(kern[ep]["ms_per_step"] / kern[ep]["avg_ms"]) if kern[ep]["avg_ms"] else None
            r["avg_ms"] = kern[ep]["avg_ms"]
            return r
    return None


def _aggregate(recs, steps):
    """[(ms, bytes, flops executed, flops direct, symbol)] -> one table entry."""
    tot_ms, tot_b, tot_f, tot_q = (sum(r[i] for r in recs) for i in range(4))
    out = {"launches": len(recs), "avg_ms": round(tot_ms / len(recs), 4), "ms_per_step": round(tot_ms / steps, 3),
           "algo_GBps": round(tot_b / (tot_ms * 1e-3) / 1e9, 1)}
    if tot_f:
        # TFLOPps: matrix-core flops EXECUTED per second (what the MFMA roofline is about); the Winograd convolutions
        # execute 1/3 .. 1/2 of the direct formulation's flops -- the direct-equivalent rate is useful work per second
        out["TFLOPps"] = round(tot_f / (tot_ms * 1e-3) / 1e12, 2)
        if tot_q != tot_f:
            out["TFLOPps_direct_equivalent"] = round(tot_q / (tot_ms * 1e-3) / 1e12, 2)
    return out


def by_symbol(times):
    """{entry point: records} -> {kernel symbol: (entry point, records)} for the launches ops.py could name."""
    out = {}
    for name, recs in times.items():
        for r in recs:
            if len(r) > 4 and r[4]:
                out.setdefault(r[4], (name, []))[1].append(r)
    return out


def kernel_sources_sha256():
    """Hash of every kernel source + the C-ABI header (the same function as scripts/pmc_traffic.py's)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "opticalflowscivis_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".hpp")))
    for path in files + [os.path.join(ROOT, "include", "flowsci_hip.h")]:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


PMC_TRAFFIC_FILE = "r05_pmc_traffic.json"


def pmc_traffic(kernel, S, B):
    """HBM bytes per launch (averaged over the entry point's launches of one step) from the committed rocprofv3 --pmc
    passes of this same command (separate FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE x2 on gfx950, as
    MI355X_MICROARCH.md prescribes; scripts/final_profile.sh).  PMC counters cannot be read from inside the timed
    run, so this is a recorded figure -- valid only for the workload AND the kernels it was taken on: the file carries
    the hash of the kernel sources, and a file taken on other sources yields null (with the reason in
    `roofline.traffic_source`), never a stale number."""
    if not (S == 256 and B == 2):
        return None, "recorded for 2 x 256^3 only"
    try:
        with open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, "profiles/%s not found" % PMC_TRAFFIC_FILE
    if d.get("kernel_sources_sha256") != kernel_sources_sha256():
        return None, "profiles/%s was taken on other kernel sources (stale): rerun scripts/final_profile.sh" % PMC_TRAFFIC_FILE
    rec = d.get("symbols", {}).get(kernel) or d.get("kernels", {}).get(kernel)
    if rec is None:
        return None, "no record for %s" % kernel
    return rec["hbm_bytes_corrected"], "profiles/%s (rocprofv3 --pmc, same kernel sources)" % PMC_TRAFFIC_FILE


def _lib_switches():
    from opticalflowscivis_amd import _lib
    return _lib.switches()


def parity_at_bench_size(dev):
    """The bench line's own parity witness ON THE KERNELS IT TIMES: a fresh model takes the first two train steps at
    B = 1, 256^3 on `droplet3d_batch(1, 256, seed=1234)` and is compared with what the REFERENCE's `Model.update`
    (Flow-3D/model/RIFE.py:81) produced for the same seed and input on the CPU -- tests/golden/flow3d_256.npz, written by
    tests/golden/make_golden.py; data only, no oracle import.  At this size the 64-channel trunk layers run the
    Winograd-domain kernels the timed region runs (2048 bricks at B = 2, 1024 at B = 1; threshold 256) -- the record says
    which kernels the two steps dispatched.  Bands: losses 5e-4 relative (both steps: the second sees the first
    one's backward and AdamW), flows 1e-4 px and merged frames 2e-5 on every 8th voxel per axis, PSNR 0.01 dB."""
    import numpy as np
    from opticalflowscivis_amd import ops
    from opticalflowscivis_amd.data import synthetic
    from opticalflowscivis_amd.flow3d.model.RIFE import Model
    g = np.load(os.path.join(ROOT, "tests", "golden", "flow3d_256.npz"))
    S = int(g["size"])
    data = synthetic.droplet3d_batch(1, S, seed=1234)
    ok = bool(np.array_equal(np.array([float(data[0, c].double().sum()) for c in range(3)]), g["data_sums"]))
    rec = {"reference": "tests/golden/flow3d_256.npz (the reference's Model.update on the CPU, B=1 at %d^3, seed 1234)" % S,
           "input_matches_fixture": ok, "tolerance": {"loss_rel": 5e-4, "flow_px": 1e-4, "merged": 2e-5, "psnr_dB": 0.01}}
    torch.manual_seed(1234)
    m = Model(local_rank=-1, device=dev)
    imgs, gt = data[:, :2].to(dev), data[:, 2:3].to(dev)
    sl = (slice(None), slice(None), slice(0, None, 8), slice(0, None, 8), slice(0, None, 8))
    names = ("loss_l1", "loss_tea", "loss_distill", "loss_G")
    ops.enable_kernel_timing(True)
    worst = 0.0
    for step in range(2):
        pred, info = m.update(imgs, gt, learning_rate=1e-4, training=True)
        torch.cuda.synchronize()
        for k, b in zip(names, g["update_losses"][step]):
            a = float(info[k].detach())
            rel = abs(a - float(b)) / abs(float(b))
            rec["step%d_%s" % (step + 1, k)] = {"gpu": a, "reference": float(b), "rel": rel}
            worst = max(worst, rel)
        if step == 0:
            for name, t, tol in (("flow", info["flow"], 1e-4), ("flow_tea", info["flow_tea"], 1e-4),
                                 ("merged", pred, 2e-5), ("merged_tea", info["merged_tea"], 2e-5)):
                d = float((t.detach()[sl].cpu() - torch.from_numpy(g[name + "_s8"])).abs().max())
                rec[name + "_max_abs_diff"] = d
                ok = ok and d < tol
            dp = abs(synthetic.psnr(pred.detach(), gt) - float(g["psnr"]))
            rec["psnr_gpu_dB"], rec["psnr_reference_dB"] = synthetic.psnr(pred.detach(), gt), float(g["psnr"])
            ok = ok and dp < 0.01
        del pred, info
    sym = by_symbol(ops.kernel_timings())
    ops.enable_kernel_timing(False)
    rec["kernels_exercised"] = {k: len(v[1]) for k, v in sorted(sym.items())}
    rec["loss_rel_worst"] = worst
    rec["ok"] = bool(ok and worst <= 5e-4)
    del m
    torch.cuda.empty_cache()
    return rec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=256, help="volume edge (BASELINE: 256)")
    ap.add_argument("--batch", type=int, default=2, help="volume pairs per GPU (BASELINE: 2)")
    ap.add_argument("--dataset", default="droplet3d", choices=["droplet3d", "jets3d"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="(default at N = 1 since round 5) replay the step from one HIP graph (Model.graphed_update)")
    ap.add_argument("--eager", action="store_true",
                    help="N = 1: time eager launches (Model.update) instead of the HIP-graph replay; N > 1 (DDP) is always "
                         "eager.  The graph-replayed run reports the eager figure next to it either way (`step_drivers`)")
    ap.add_argument("--cpu-size", type=int, nargs="+", default=[64, 128],
                    help="edges of the bounded CPU samples (SURVEY 8d: 64^3 and 128^3); the last one is `value`")
    ap.add_argument("--deterministic", action="store_true",
                    help="torch.use_deterministic_algorithms(True): the weight gradients take the workspace form without "
                         "float atomics (fs_conv3d_wrw_det) -- the whole step is then bitwise reproducible")
    ap.add_argument("--no-bench-parity", action="store_true",
                    help="skip parity_at_bench_size (two steps at B=1 x 256^3 against tests/golden/flow3d_256.npz)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the C2 (Flow-2D) and C3 (UPFlow) legs that follow the headline leg at N = 1")
    ap.add_argument("--config-steps", type=int, default=20, help="timed steps of the C2 / C3 legs")
    ap.add_argument("--record-steps", type=int, default=2,
                    help="steps of the separate per-launch recording pass (roofline / kernels entries)")
    return ap.parse_args()


def usable_cores():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return min(n, 64)


def log(msg):
    print("[bench %6.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def cpu_baseline(sizes, dataset, dev, steps=2):
    """The oracle's Flow-3D train step (the reference's CPU PyTorch path, restated) on this box's host cores,
    on bounded samples: B = 2 at 64^3 (the reference's own training resolution) and B = 1 at 128^3, each
    reported as 256^3-equivalent pairs/s (scaled by voxel count); `value` is the largest sample's.  The
    first oracle step of every sample starts from the seed-1234 weights; the GPU model takes the same step
    from the same seed on the same batch -> `parity` (losses, final flow, interpolation PSNR)."""
    from oracle.ifnet_ref import ModelRef
    from opticalflowscivis_amd.data import synthetic
    from opticalflowscivis_amd.flow3d.model.RIFE import Model
    cores = usable_cores()
    torch.set_num_threads(cores)
    gen = synthetic.droplet3d_batch if dataset == "droplet3d" else synthetic.jets3d_batch
    samples, parity = [], []
    for size in sizes:
        B = 2 if size <= 64 else 1
        log("cpu_baseline: oracle step at B=%d x %d^3 on %d cores" % (B, size, cores))
        torch.manual_seed(1234)
        m = ModelRef(3)
        data = gen(B, size, seed=1234)
        imgs, gt = data[:, :2], data[:, 2:3]
        # first step: also the warm-up (allocator, thread pool) and the parity reference
        po, oi = m.update(imgs, gt, learning_rate=1e-4, training=True)
        t0 = time.perf_counter()
        o2 = None
        for _ in range(steps):
            _, oi_n = m.update(imgs, gt, learning_rate=1e-4, training=True)
            o2 = float(oi_n["loss_G"].detach()) if o2 is None else o2  # step 2: sees step 1's backward + AdamW
            del oi_n
        dt = (time.perf_counter() - t0) / steps
        vox_ratio = (size / 256.0) ** 3
        samples.append({"size": size, "batch": B, "s_per_step": round(dt, 4), "timed_steps": steps,
                        "pairs_per_s": B / dt, "pairs_per_s_256eq": B / dt * vox_ratio})
        # the GPU side of the parity witness
        torch.manual_seed(1234)
        g = Model(local_rank=-1, device=dev)
        pg, gi = g.update(imgs.to(dev), gt.to(dev), learning_rate=1e-4, training=True)
        g2 = float(g.update(imgs.to(dev), gt.to(dev), learning_rate=1e-4, training=True)[1]["loss_G"].detach())
        torch.cuda.synchronize()
        rec = {"size": size, "batch": B,
               "loss_G_step2": {"gpu": g2, "oracle": o2, "rel": abs(g2 - o2) / max(abs(o2), 1e-12)}}
        for k in ("loss_l1", "loss_tea", "loss_distill", "loss_G"):
            a, b = float(gi[k].detach()), float(oi[k].detach())
            rec[k] = {"gpu": a, "oracle": b, "rel": abs(a - b) / max(abs(b), 1e-12)}
        rec["flow_max_abs_diff_px"] = float((gi["flow"].detach().cpu() - oi["flow"].detach()).abs().max())
        gtc = gt[(slice(None), slice(None)) + tuple(slice(0, n) for n in po.shape[2:])]
        rec["psnr_gpu_dB"] = synthetic.psnr(pg.detach().cpu(), gtc)
        rec["psnr_oracle_dB"] = synthetic.psnr(po.detach(), gtc)
        parity.append(rec)
        del g, m
        torch.cuda.empty_cache()
    # the metric's own size, measured instead of scaled: ONE timed oracle step at B = 1, 256^3 after
    # one warm-up step -- only when the host can hold it (~35 GB) and the bounded samples say it fits the time budget
    full = {"size": 256, "batch": 1, "measured": False}
    try:
        with open("/proc/meminfo") as f:
            avail_gb = next(int(l.split()[1]) for l in f if l.startswith("MemAvailable:")) / 2 ** 20
    except (OSError, StopIteration, ValueError):
        avail_gb = 0.0
    est = samples[-1]["s_per_step"] * (256.0 / samples[-1]["size"]) ** 3 / samples[-1]["batch"] * 1.15
    if 256 in sizes:
        full = None  # already one of the samples
    elif os.environ.get("FLOWSCI_BENCH_NO_CPU_256") == "1":
        full["reason"] = "switched off (FLOWSCI_BENCH_NO_CPU_256=1)"
    elif avail_gb < 56:
        full["reason"] = "MemAvailable %.0f GB < 56 GB (the oracle step peaks at ~35 GB)" % avail_gb
    elif 2 * est > 200:
        full["reason"] = "estimated %.0f s per step from the %d^3 sample: two steps exceed the 200 s budget" % (est, samples[-1]["size"])
    else:
        log("cpu_baseline: one timed oracle step at B=1 x 256^3 (estimate %.0f s per step, %.0f GB available)" % (est, avail_gb))
        torch.manual_seed(1234)
        m = ModelRef(3)
        data = gen(1, 256, seed=1234)
        imgs, gt = data[:, :2], data[:, 2:3]
        m.update(imgs, gt, learning_rate=1e-4, training=True)      # warm-up (allocator, thread pool)
        t0 = time.perf_counter()
        m.update(imgs, gt, learning_rate=1e-4, training=True)
        dt = time.perf_counter() - t0
        full.update(measured=True, s_per_step=round(dt, 3), timed_steps=1, pairs_per_s=1.0 / dt, pairs_per_s_256eq=1.0 / dt)
        del m, data, imgs, gt
    if full is not None:
        samples.append(full)
    last = next(x for x in reversed(samples) if x.get("measured", True))
    base = {"value": last["pairs_per_s_256eq"], "unit": "volume-pairs/s (256^3-equivalent)",
            "cores": cores, "kind": "port",
            "sample": "oracle Flow-3D train step on %s: %s; value = the %d^3 sample scaled by voxel count "
                      "(x%.4f) to 256^3" % (dataset, "; ".join(
                          "B=%d at %d^3: %d timed steps, %.2f s/step" % (x["batch"], x["size"], x["timed_steps"],
                                                                         x["s_per_step"]) for x in samples if x.get("measured", True)),
                          last["size"], (last["size"] / 256.0) ** 3),
            "samples": samples}
    top = parity[-1]
    witness = {"loss_gpu": top["loss_G"]["gpu"], "loss_oracle": top["loss_G"]["oracle"],
               "rel": max(p[k]["rel"] for p in parity for k in ("loss_l1", "loss_tea", "loss_distill", "loss_G")),
               "tolerance_rel": 5e-4,
               # the flow AFTER the step's forward is the backward fusions' witness only at step 2; the first
               # step's flow pins the fused forward (distillation aliases, up-sample + warp launches)
               "flow_max_abs_diff_px": max(p["flow_max_abs_diff_px"] for p in parity), "tolerance_flow_px": 1e-4,
               "loss_G_step2_rel": max(p["loss_G_step2"]["rel"] for p in parity),
               "sizes": parity}
    return base, witness


def kfd_gpus():
    """The GPUs of this host as the kernel driver lists them -- /sys/class/kfd/kfd/topology/nodes/*/properties, nodes
    with simd_count > 0, in node order (the order ROCr enumerates agents in) -- each with its PCI address and the CPUs
    local to it.  Plain file reads: no HIP / ROCr call, so the launcher can size the job before any child exists."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    out = []
    try:
        nodes = sorted((int(n) for n in os.listdir(base) if n.isdigit()))
    except OSError:
        return out
    for n in nodes:
        try:
            with open("%s/%d/properties" % (base, n)) as f:
                prop = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue  # (a node this cgroup may not read is not a device this process can use either)
        if int(prop.get("simd_count", "0")) <= 0:
            continue
        loc, dom = int(prop.get("location_id", "0")), int(prop.get("domain", "0"))
        bdf = "%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)
        cpus = None
        try:
            with open("/sys/bus/pci/devices/%s/local_cpulist" % bdf) as f:
                cpus = _parse_cpulist(f.read())
        except (OSError, ValueError):
            pass
        out.append({"kfd_node": n, "pci": bdf, "local_cpus": cpus})
    return out


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def visible_gpu_list():
    """kfd_gpus() filtered / re-ordered the way the runtime will see them: ROCR_VISIBLE_DEVICES first (agent level),
    then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (indices into what is left).  UUID entries are not resolved: they
    keep their position but lose the PCI / CPU information."""
    gpus = kfd_gpus()
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = os.environ.get(var)
        if val is None:
            continue
        pick = []
        for tok in val.split(","):
            tok = tok.strip()
            if tok.isdigit() and int(tok) < len(gpus):
                pick.append(gpus[int(tok)])
            elif tok.isdigit() or tok in ("", "-1"):
                break  # the runtimes stop at the first invalid index
            else:
                pick.append({"kfd_node": None, "pci": None, "local_cpus": None, "id": tok})
        gpus = pick
        if var != "ROCR_VISIBLE_DEVICES":
            break  # HIP_ and CUDA_VISIBLE_DEVICES are aliases: the first one set wins
    return gpus


def bind_to_gpu_cpus(local_rank):
    """One rank per GPU: keep the rank's host threads (Python launch path, the allocator, RCCL's proxy threads) on the
    cores local to ITS GPU's PCI root.  C2 / C3-class launch-bound steps measured 85 vs 131 ms on the same GPU type
    by host; cross-socket launches are the avoidable part of that.  Must run before the process creates its thread
    pools (affinity is inherited by new threads) and makes no GPU call.  Returns a record for the JSON line."""
    gpus = visible_gpu_list()
    rec = {"local_rank": local_rank, "bound": False}
    if local_rank >= len(gpus) or not gpus[local_rank].get("local_cpus") or not hasattr(os, "sched_setaffinity"):
        rec["why"] = "no PCI-local CPU list for this rank's GPU"
        return rec
    want = gpus[local_rank]["local_cpus"] & os.sched_getaffinity(0)
    rec["pci"] = gpus[local_rank]["pci"]
    if len(want) < 2:
        rec["why"] = "fewer than 2 allowed CPUs are local to the GPU"
        return rec
    os.sched_setaffinity(0, want)
    rec.update(bound=True, cpus=len(want))
    return rec


def launch(args):
    """`python bench.py --gpus N` without rank variables in the environment: start N fresh rank processes of
    this same script (one per GPU, env:// rendezvous on 127.0.0.1), pass rank 0's stdout (the ONE JSON line)
    through, send the other ranks' stdout to stderr, and return the worst exit status.  Nothing here may
    initialise the GPU: the children are started from a process that has made no HIP call
    (the device count is read from /sys/class/kfd topology files, `visible_gpu_list`), and no process is ever
    replaced by exec.  When one rank dies the others are ended by their exact PIDs, so that a failure is an
    exit code and not a hung rendezvous.
    FLOWSCI_BENCH_SHARE_GPU=1 (tests on a one-GPU box only, never the driver): ranks are dealt round-robin
    over the visible devices and the process group runs on gloo, because RCCL refuses two ranks on one device."""
    import socket
    import subprocess
    # the device count comes from the kernel driver's topology files -- the parent makes no HIP / ROCr call at all
    # (torch.cuda.device_count() happens not to initialise the runtime on this image; that is an implementation detail)
    ndev = len(visible_gpu_list())
    if ndev == 0:  # topology files hidden from this container: ask torch (no runtime initialisation on this image)
        ndev = torch.cuda.device_count()
    share = os.environ.get("FLOWSCI_BENCH_SHARE_GPU") == "1"
    if ndev < 1:
        sys.exit("bench.py needs a ROCm GPU: the HIP hot path has no CPU fallback")
    if args.gpus > ndev and not share:
        sys.exit("bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, ndev))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r % ndev), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FLOWSCI_BENCH_LAUNCHER="self")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    worst, alive = 0, set(range(args.gpus))
    while alive:
        for r in sorted(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.discard(r)
            if rc != 0:
                print("[bench launcher] rank %d exited with status %d" % (r, rc), file=sys.stderr, flush=True)
                worst = worst or (rc if rc > 0 else 128 - rc)
                for o in sorted(alive):  # the exact children started above, nothing by pattern
                    procs[o].terminate()
        time.sleep(0.05)
    return worst


HOT_2D = ("fs_warp2d_fwd", "fs_warp2d_bwd", "fs_warp2d_pair_fwd", "fs_warp2d_pair_bwd", "fs_corr2d_fwd",
          "fs_corr2d_bwd", "fs_corr2d_pair_fwd", "fs_corr2d_pair_bwd", "fs_corr2d_norm_fwd", "fs_corr2d_norm_bwd",
          "fs_census_dist_fwd", "fs_census_dist_bwd", "fs_robust_sum", "fs_robust_sum_bwd")


def _time_steps(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def _hot_kernels(fn, steps=2):
    """Per-launch HIP-event records of `steps` steps -> {entry point: {...}} for the SURVEY 8a rows, plus the one
    with the largest total time (its algorithmic GB/s is the figure the row is judged by)."""
    from opticalflowscivis_amd import ops
    ops.enable_kernel_timing(True)
    for _ in range(steps):
        fn()
    rec = ops.kernel_timings()
    ops.enable_kernel_timing(False)
    out, launches = {}, 0
    for name, rs in rec.items():
        launches += len(rs)
        if name not in HOT_2D:
            continue
        ms, nb, fl = (sum(r[i] for r in rs) for i in range(3))
        out[name] = {"launches_per_step": len(rs) / steps, "ms_per_step": round(ms / steps, 4),
                     "algo_GBps": round(nb / (ms * 1e-3) / 1e9, 1) if nb else None,
                     "frac_of_hbm_peak": round(nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if nb else None}
        if fl:
            out[name]["TFLOPps"] = round(fl / (ms * 1e-3) / 1e12, 2)
    dom = max(out, key=lambda k: out[k]["ms_per_step"]) if out else None
    return out, dom, launches / steps


def config_c2(dev, steps, warmup):
    """BASELINE config 2: Flow-2D droplet 160 x 224, batch 16, one unsupervised train step (Flow-2D/train.py:169):
    eager, and replayed from one HIP graph (the step is launch-bound)."""
    from opticalflowscivis_amd.data import synthetic
    from opticalflowscivis_amd.flow2d.model.RIFE import Model
    B = 16
    data = synthetic.droplet2d_batch(B, 160, 224, seed=1234, device=dev)
    imgs, gt = data[:, :2].contiguous(), data[:, 2:3].contiguous()
    torch.manual_seed(1234)
    m = Model(local_rank=-1, device=dev)
    lr = 3e-4 * (10 / 2000.) / 4
    eager = lambda: m.update(imgs, gt, "droplet2d", learning_rate=lr, training=True)
    dt = _time_steps(eager, steps, warmup)
    hot, dom, launches = _hot_kernels(eager)
    out = {"workload": "Flow-2D droplet2d 160x224, batch 16, one train step (fwd + losses + bwd + AdamW)",
           "unit": "frame-pairs/s", "ms_per_step": dt * 1e3, "pairs_per_s": B / dt, "steps": steps,
           "flowsci_launches_per_step": launches, "hot_path_kernels": hot, "dominant_hot_path_kernel": dom}
    g = m.graphed_update(imgs, gt, dataset="droplet2d")
    dg = _time_steps(lambda: g(imgs, gt, lr), steps, warmup)
    # the step driver of Flow-2D's train.py at N = 1 is the graph replay (round 5): it is the leg's figure, eager beside it
    out["eager"] = {"ms_per_step": dt * 1e3, "pairs_per_s": B / dt}
    out["graph_replay"] = {"ms_per_step": dg * 1e3, "pairs_per_s": B / dg}
    out["ms_per_step"], out["pairs_per_s"] = dg * 1e3, B / dg
    out["step_driver"] = "hip-graph replay (Model.graphed_update)"
    return out


def config_c2_cpu():
    from opticalflowscivis_amd.data import synthetic
    from oracle.ifnet_ref import ModelRef
    B = 16
    data = synthetic.droplet2d_batch(B, 160, 224, seed=1234)
    hi, hg = data[:, :2].contiguous(), data[:, 2:3].contiguous()
    lr = 3e-4 * (10 / 2000.) / 4
    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(1234)
    o = ModelRef(2)
    o.update(hi, hg, learning_rate=lr)
    t0 = time.perf_counter()
    for _ in range(3):
        o.update(hi, hg, learning_rate=lr)
    dc = (time.perf_counter() - t0) / 3
    return {"value": B / dc, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
            "sample": "oracle Flow-2D train step, the full C2 batch (B=16, 160x224): 3 timed steps, %.2f s/step" % dc}


def config_c3(dev, steps, warmup):
    """BASELINE config 3: UPFlow on 150 x 450 pairs, batch 32, pyramid cost volume + census loss, one train step
    (UPFlow/scripts/simple_train.py:278-285)."""
    from opticalflowscivis_amd.data import synthetic
    from opticalflowscivis_amd.upflow.scripts.simple_train import Loss_manager, Trainer
    B = 32
    conf = Trainer.Config(exp_dir="/tmp/upflow_bench")
    conf.net_params = dict(conf.net_params, photo_loss_census_weight=1)
    torch.manual_seed(0)
    tr = Trainer(conf, device=dev)
    opt = torch.optim.Adam(tr.net.parameters(), lr=1e-4, weight_decay=1e-4, amsgrad=True)
    pairs = synthetic.vortex2d_pairs(B, 150, 450, seed=0, device=dev)
    im1, im2 = pairs[:, 0].contiguous(), pairs[:, 1].contiguous()
    lm = Loss_manager()

    def step():
        o = tr.net({'im1': im1, 'im2': im2, 'if_loss': True})
        loss = lm.compute_loss(o['loss_dict'], B)
        opt.zero_grad()
        loss.backward()
        opt.step()
    dt = _time_steps(step, steps, warmup)
    hot, dom, launches = _hot_kernels(step)
    out = {"workload": "UPFlow vortex pairs 3x150x450, batch 32, census on, one train step (fwd + losses + bwd + Adam)",
           "unit": "frame-pairs/s", "ms_per_step": dt * 1e3, "pairs_per_s": B / dt, "steps": steps,
           "flowsci_launches_per_step": launches, "hot_path_kernels": hot, "dominant_hot_path_kernel": dom,
           "note": "the stock 2-D convolutions of the PWC network (MIOpen) are most of this step; "
                   "hot_path_kernels are the SURVEY 8a rows"}
    return out


def config_c3_cpu():
    from oracle.upflow_port import c3_step_seconds
    cores = usable_cores()
    Bc = 8
    dc, _ = c3_step_seconds(Bc, steps=1, threads=cores)
    return {"value": Bc / dc, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
            "sample": "the UPFlow step with the oracle's CPU ops (unfold correlation as Corr_pyTorch), B=%d of the 32 "
                      "pairs at 150x450: 1 timed step after one warm-up, %.2f s/step" % (Bc, dc)}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch(args))
    # stdout must carry exactly ONE JSON line, but RCCL prints a version banner to fd 1 when its
    # communicator comes up.  Park the real stdout, send fd 1 to stderr for the run, and write the
    # result to the parked descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("WORLD_SIZE=%d overrides --gpus %d" % (world, args.gpus))
        args.gpus = world
    # N > 1: this rank's host threads stay on the cores local to its GPU (before anything creates a thread pool or
    # touches the GPU); N = 1 keeps the whole host, which the cpu_baseline leg uses
    affinity = bind_to_gpu_cpus(local_rank) if world > 1 and os.environ.get("FLOWSCI_BENCH_SHARE_GPU") != "1" else None
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a ROCm GPU: the HIP hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # FLOWSCI_BENCH_FORCE_DDP=1: run the N > 1 code path (RCCL process group + DDP wrapper) with a
    # single rank -- a rehearsal for boxes with one GPU; never set by the driver
    ddp = world > 1 or os.environ.get("FLOWSCI_BENCH_FORCE_DDP") == "1"
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if os.environ.get("FLOWSCI_BENCH_SHARE_GPU") == "1":  # tests: several ranks on one device (launch())
            dist.init_process_group(backend="gloo", world_size=world, rank=rank)
        else:
            dist.init_process_group(backend="nccl", world_size=world, rank=rank, device_id=dev)  # RCCL on ROCm
        world = dist.get_world_size()  # from here on N is what the process group says, not what the env said

    from opticalflowscivis_amd import ops
    from opticalflowscivis_amd.data import synthetic
    from opticalflowscivis_amd.flow3d.model.RIFE import Model

    if args.deterministic:
        torch.use_deterministic_algorithms(True)
        # (torch would otherwise NaN-fill every torch.empty() under the flag -- +7.5 ms of fill kernels per 256^3 step over
        # the outputs every kernel here overwrites anyway)
        torch.utils.deterministic.fill_uninitialized_memory = False
    torch.manual_seed(1234)  # same initial weights on every rank (DDP would broadcast anyway)
    model = Model(local_rank=local_rank if ddp else -1, device=dev)
    S, B = args.size, args.batch
    gen = synthetic.droplet3d_batch if args.dataset == "droplet3d" else synthetic.jets3d_batch
    data = gen(B, S, seed=1234 + rank, device=dev)  # resident in HBM before the timed region
    imgs, gt = data[:, :2].contiguous(), data[:, 2:3].contiguous()
    # reference LR schedule scaled by world_size / 4 (Flow-3D/train.py:167), warm-up phase value
    lr = 3e-4 * (10 / 2000.) * world / 4

    def barrier():
        if ddp:
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        log("data + model ready; %d warm-up steps" % args.warmup)
    # step driver: one HIP graph replayed per step at N = 1 (round 5: the default; --eager opts out), eager launches under
    # DistributedDataParallel (capture through DDP's reducer is refused, rife.py)
    use_graph = (not ddp) and (not args.eager)
    eager_step = lambda: model.update(imgs, gt, learning_rate=lr, training=True)
    if use_graph:
        graph_step = model.graphed_update(imgs, gt)
        do_step = lambda: graph_step(imgs, gt, lr)
    else:
        do_step = eager_step
    # warm-up: every C-ABI launch is recorded (HIP events on the launch stream) -- this warms the event pool
    # and tells which entry point dominates the step
    ops.enable_kernel_timing(True)
    for i in range(args.warmup):
        eager_step() if (use_graph and i == 0) else do_step()  # (one eager step: the warm-up's per-launch records)
        torch.cuda.synchronize()
        if rank == 0:
            log("warm-up step %d done" % i)
    wt = ops.kernel_timings()
    # the dominant KERNEL of the warm-up steps: the symbol with the largest total time among the launches ops.py can
    # name (the Winograd trunk kernels; `rocprofv3 --kernel-trace --stats` of this command, profiles/, names the same
    # one); its entry point is the one followed with events inside the timed region
    # the dominant KERNEL of the warm-up steps: kernel symbols ops.py can name and, per entry point, the launches it
    # cannot (the same candidates the final `roofline` record is chosen from); its entry point is the one followed with
    # events inside the timed region
    wsym = by_symbol(wt)
    dom_sym = max(wsym, key=lambda k: sum(r[0] for r in wsym[k][1])) if wsym else None
    dom_guess = wsym[dom_sym][0] if dom_sym else (max(wt, key=lambda k: sum(r[0] for r in wt[k])) if wt else None)
    barrier()
    # timed region: only that entry point's launches carry events (~90 of ~430 launches per step), so the headline
    # time is free of profiling overhead while `roofline` is still measured inside the region
    ops.enable_kernel_timing(True, only=[dom_guess] if dom_guess else None)
    t0 = time.perf_counter()
    step_marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    step_marks[0].record()
    for i in range(args.steps):
        pred, info = do_step()
        step_marks[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    timed = ops.kernel_timings()
    eager_dt = None
    if use_graph:
        # a replayed graph carries no per-launch events: the same K steps once more as eager launches, right behind the
        # timed region, give the eager step time (`step_drivers`) and the dominant kernel's launch durations (`roofline`)
        ops.enable_kernel_timing(True, only=[dom_guess] if dom_guess else None)
        barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            eager_step()
        barrier()
        eager_dt = time.perf_counter() - t1
        timed = ops.kernel_timings()
    # separate recording pass (outside the timed region) for the per-entry-point `kernels` table
    ksteps = max(1, args.record_steps)
    ops.enable_kernel_timing(True)
    for _ in range(ksteps):
        model.update(imgs, gt, learning_rate=lr, training=True)
    ktimes = ops.kernel_timings()
    ops.enable_kernel_timing(False)
    loss = float(info["loss_G"].detach())
    if rank == 0:
        log("timed region: %d steps in %.3f s; per-step ms: %s; peak HBM %.1f GB" % (
            args.steps, dt, " ".join("%.0f" % step_marks[i].elapsed_time(step_marks[i + 1])
                                     for i in range(args.steps)),
            torch.cuda.max_memory_allocated() / 2 ** 30))
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    ranks = None
    if ddp:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # witness that the collective really spanned N processes: sum over ranks of (rank + 1) == N (N + 1) / 2,
        # plus what every rank says about itself (device, its own mean step time, its loss)
        wsum = torch.tensor([rank + 1.0], device=dev, dtype=torch.float64)
        dist.all_reduce(wsum, op=dist.ReduceOp.SUM)
        prop = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "pid": os.getpid(), "device": dev.index, "cpu_affinity": affinity,
                "device_name": prop.name, "device_uuid": str(getattr(prop, "uuid", "")),
                "pci_bus_id": getattr(prop, "pci_bus_id", None),
                "ms_per_step": step_marks[0].elapsed_time(step_marks[-1]) / args.steps,
                "loss_G": float(info["loss_G"].detach())}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ms = [g["ms_per_step"] for g in gathered]
        ranks = {"world_size": world, "backend": dist.get_backend(),
                 "launcher": os.environ.get("FLOWSCI_BENCH_LAUNCHER", "torch.distributed.run"
                                            if "TORCHELASTIC_RUN_ID" in os.environ else "env"),
                 "allreduce_sum_of_rank_plus_1": float(wsum.item()), "expected": world * (world + 1) / 2,
                 "devices": len({(g["device"], g["device_uuid"]) for g in gathered}),
                 "ms_per_step_min": min(ms), "ms_per_step_max": max(ms), "per_rank": gathered}
        if ranks["allreduce_sum_of_rank_plus_1"] != ranks["expected"]:
            sys.exit("process group witness failed: %r" % ranks)
    dt = float(t.item())

    if rank == 0:
        kern = {name: _aggregate(recs, ksteps) for name, recs in ktimes.items()}
        ksym = {sym: dict(_aggregate(recs, ksteps), entry_point=ep) for sym, (ep, recs) in by_symbol(ktimes).items()}
        # `roofline` = the dominant hand-written KERNEL (by symbol, as rocprofv3's kernel stats name it); its record
        # comes from the events recorded inside the timed region when its entry point was the one followed there
        # `roofline` is the dominant hand-written KERNEL among the symbols ops.py can name (what `rocprofv3 --kernel-trace
        # --stats` of this command lists first).  Launches it cannot name are filed under their entry point, which may
        # stand for several kernels: the largest such group and the named kernel's share of the step are reported beside
        # the record (`named_kernel_share_of_step`, `largest_unnamed_entry_point`), so that a reader sees when the named
        # kernel is not what dominates
        unnamed = {}
        for ep, recs in ktimes.items():
            rest = [r for r in recs if not (len(r) > 4 and r[4])]
            if rest:
                unnamed[ep] = _aggregate(rest, ksteps)
        dom = max(ksym, key=lambda k: ksym[k]["ms_per_step"]) if ksym else max(kern, key=lambda k: kern[k]["ms_per_step"])
        dom_rec = (ksym if ksym else kern)[dom]
        big_unnamed = max(unnamed, key=lambda k: unnamed[k]["ms_per_step"]) if unnamed else None
        dom_src = "HIP events, separate pass of %d steps after the timed region" % ksteps
        tsym = by_symbol(timed)
        if dom in tsym:
            dom_rec = dict(_aggregate(tsym[dom][1], args.steps), entry_point=tsym[dom][0])
            dom_src = "HIP events on the launch stream inside the timed region (%d launches)" % dom_rec["launches"]
        elif dom in timed and timed[dom]:
            dom_rec = _aggregate(timed[dom], args.steps)
            dom_src = "HIP events on the launch stream inside the timed region (%d launches)" % dom_rec["launches"]
        if use_graph and "inside the timed region" in dom_src:
            dom_src = ("HIP events on the launch stream inside the eager pass of %d steps run right behind the graph-replayed "
                       "timed region (a replayed graph carries no per-launch events; %d launches)" % (args.steps, dom_rec["launches"]))
        # the entry point with the largest total time, for continuity with rounds 1-3 (same definition of `frac` now)
        dom_ep = max(kern, key=lambda k: kern[k]["ms_per_step"])
        out = {
            "metric": "volume-pairs/sec, Flow-3D unsupervised train step (fwd+loss+bwd+AdamW)",
            "value": world * B * args.steps / dt,
            "unit": "volume-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            # every tensor and every accumulation is fp32; one kernel family multiplies on the bf16 matrix cores without
            # giving up fp32 accuracy (error vs fp64 <= the fp32 MFMA kernels': profiles/r05_split_bf16.txt)
            "arithmetic": "fp32; the k4 s2 forward convolutions (conv3d_fwd_s3_kernel) and the 17..32-channel transposed ones "
                          "(convtr_s3_kernel) multiply fp32 operands as 3 bf16 pieces, 6 products, fp32 accumulate",
            "config": {"workload": "Flow-3D %s %d^3, batch %d per GPU, IFNet-3D random init, "
                                   "3D trilinear warp HIP kernels" % (args.dataset, S, B),
                       "global_batch": world * B, "volume": [S, S, S],
                       "parallelism": "dp%d" % world},
            "roofline": dict(roofline(dom, dom_rec, S, B), launches=dom_rec["launches"], avg_ms=dom_rec["avg_ms"],
                             ms_per_step=dom_rec["ms_per_step"], entry_point=dom_rec.get("entry_point", dom),
                             measured=dom_src,
                             named_kernel_share_of_step=round(dom_rec["ms_per_step"] / (dt / args.steps * 1e3), 4),
                             largest_unnamed_entry_point=(None if big_unnamed is None else
                                                          {"entry_point": big_unnamed, "ms_per_step": unnamed[big_unnamed]["ms_per_step"],
                                                           "launches_per_step": unnamed[big_unnamed]["launches"] / ksteps,
                                                           "note": "several kernels behind one entry point, not one kernel"})),
            "roofline_entry_point": dict(roofline(dom_ep, kern[dom_ep], S, B), ms_per_step=kern[dom_ep]["ms_per_step"]),
            # the hot-path row the metric is named after (the trilinear backward warp pair)
            # against the HBM roofline, with its measured HBM traffic per launch; `roofline` above is the
            # kernel that dominates the step's time
            "roofline_hbm": roofline_hbm(kern, ktimes, S, B),
            "kernels": kern,
            "kernel_symbols": ksym,
            "loss_G": loss,
            # what in the environment could have changed dispatch or numerics (INTEGRATION.md "Switches")
            "switches": _lib_switches(),
            "deterministic": bool(torch.are_deterministic_algorithms_enabled()),
            "step_driver": "hip-graph replay (Model.graphed_update)" if use_graph else
                           "eager launches (Model.update%s)" % (" under DistributedDataParallel" if ddp else ""),
            # both drivers on this run's N (N > 1: eager only -- graph capture through DDP's reducer is refused)
            "step_drivers": {"hip_graph_replay_ms_per_step": (dt / args.steps * 1e3) if use_graph else None,
                             "eager_ms_per_step": (eager_dt / args.steps * 1e3) if use_graph else dt / args.steps * 1e3,
                             "default": "hip-graph replay at N = 1 (--eager opts out); eager under DDP"},
        }
        if ranks is not None:
            out["ranks"] = ranks
        rc = 0
        if world == 1 and S == 256 and not args.no_bench_parity:
            del pred, info
            pred = info = None
            model = imgs = gt = data = None
            torch.cuda.empty_cache()
            log("parity_at_bench_size: two steps at B=1 x 256^3 against the reference's golden values")
            out["parity_at_bench_size"] = w = parity_at_bench_size(dev)
            if not w["ok"]:
                log("PARITY FAILURE at the bench size: %s" % json.dumps(w))
                rc = 4
        if world == 1 and not args.no_configs and not ddp:
            # the other single-GPU BASELINE configurations, driver-timed in the same run (extra keys; the top-level
            # metric stays config 4's): C2 Flow-2D 160x224 B=16, C3 UPFlow 150x450 B=32 with census.  Their GPU legs
            # run BEFORE any CPU baseline: the oracle's OpenMP team keeps spinning on every host core after its
            # last step and slows the Python launch path of these launch-bound steps (C3 measured 132 ms after the
            # CPU leg, 87 ms before it).
            pred = info = None
            model = imgs = gt = data = None
            torch.cuda.empty_cache()
            out["configs"] = {}
            for name, fn in (("C2", config_c2), ("C3", config_c3)):
                log("config %s" % name)
                out["configs"][name] = fn(dev, args.config_steps, 5)
                torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["parity_at_cpu_size"] = cpu_baseline(args.cpu_size, args.dataset, dev)
            w = out["parity_at_cpu_size"]
            if not (w["rel"] <= w["tolerance_rel"] and w["loss_G_step2_rel"] <= w["tolerance_rel"]
                    and w["flow_max_abs_diff_px"] <= w["tolerance_flow_px"]):
                log("PARITY FAILURE: GPU vs oracle: losses %.3e / step-2 loss %.3e relative, flow %.3e px" % (
                    w["rel"], w["loss_G_step2_rel"], w["flow_max_abs_diff_px"]))
                rc = rc or 3
            if "configs" in out:
                for name, fn in (("C2", config_c2_cpu), ("C3", config_c3_cpu)):
                    log("config %s: CPU port" % name)
                    out["configs"][name]["cpu_baseline"] = fn()
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    else:
        rc = 0
    if ddp:
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
