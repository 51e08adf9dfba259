"""HBM bytes per launch of every C-ABI entry point of the bench step, from two rocprofv3 --pmc passes.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/f -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/w -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline
    python scripts/pmc_traffic.py <f/..counter_collection.csv> <w/..counter_collection.csv> <bench.json> > profiles/rNN_pmc_traffic.json

Counters are collected in their own passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass);
both are in KB; FETCH_SIZE is doubled (gfx950 tallies 128-B requests of wide coalesced reads at 64 B), WRITE_SIZE is
exact for 16-byte streaming stores and float atomics.  A kernel is filed under the entry point that launches it
(the forward convolution entry point also launches `wprep_kernel`; the three backward-warp entry points share one
kernel and are told apart by their order inside a step: teacher warp, block 2 (three addends), blocks 1 and 0
(fused up-sampling) -- and the up-sampling adjoint launches that FOLLOW the warp launch of those two are part of
their entry point, fs_upsample_warp3d_pair_bwd3: every kernel of a multi-kernel entry point is attributed).  Only the LAST step of each pass is used (the first steps contain MIOpen find kernels).
`algorithmic_bytes` per launch = algo_GBps x avg_ms of the bench line (the figures ops.py supplies).
"""
import collections
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha256():
    """Hash of every kernel source + the C-ABI header: bench.py recomputes it and refuses a traffic file taken on
    other kernels (the same function lives there)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "opticalflowscivis_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".hpp")))
    for path in files + [os.path.join(ROOT, "include", "flowsci_hip.h")]:
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()

RULES = [  # (substring of the kernel name, entry point, counts as a launch of the entry point)
    ("conv3d_fwd_kernel", "fs_conv3d_fwd", True), ("conv3d_fwd_ws_kernel", "fs_conv3d_fwd", True),
    ("conv3d_wino_ws_kernel", "fs_conv3d_fwd", True), ("conv3d_wino2d_ws_kernel", "fs_conv3d_fwd", True), ("conv3d_wino2d_ps_kernel", "fs_conv3d_fwd", True),
    ("conv3d_wino4_ws_kernel", "fs_conv3d_fwd", True), ("conv3d_fwd_s3_kernel", "fs_conv3d_fwd", True),
    ("wprep_one_kernel", "fs_conv3d_fwd", False), ("wprep_batch_kernel", "fs_conv3d_wprep_batch", True),
    ("conv3d_wrw_", "fs_conv3d_wrw", True),
    ("convtr_", "fs_conv3d_tr", True),
    ("warp3d_fwd_kernel<512, true, true>", "fs_upsample_warp3d_pair_fwd", True),
    ("warp3d_fwd_kernel<512, true, false>", "fs_warp3d_pair_fwd", True), ("warp3d_fwd_ring_kernel", "fs_warp3d_pair_fwd", True),
    ("warp3d_rc_kernel<false", "fs_warp3d_pair_fwd", True),
    ("prelu_bwd_kernel", "fs_prelu_bwd", True), ("prelu_ga_kernel", "fs_prelu_bwd", False),
    ("merge_fwd_kernel", "fs_merge_fwd", True), ("merge_bwd_kernel", "fs_merge_bwd", True),
    ("distill3_fwd_kernel", "fs_distill_fwd", True), ("distill3_bwd_kernel", "fs_distill_bwd", True),
]
# kernel SYMBOLS bench.py's `roofline` may name (ops.py labels their launches): traffic per launch of the symbol itself
SYMBOLS = ["conv3d_wino2d_ps_kernel<0, 16>", "conv3d_wino2d_ps_kernel<0, 8>", "conv3d_wrw_wino4_kernel<0>",
           "conv3d_fwd_s3_kernel<1, 8, 4>", "conv3d_fwd_s3_kernel<2, 8, 4>", "convtr_s3_kernel",
           "warp3d_rc_kernel<true, 4, 5, 0>", "warp3d_rc_kernel<false, 2, 6, 0>"]
ADJOINT = ("up_adjoint_fused_kernel", "interp_axis_adjoint_kernel", "interp3d_up_adjoint")  # kernels of fs_interp3d_bwd_scaled
WARP_BWD_ORDER = ["fs_warp3d_pair_bwd", "fs_warp3d_pair_bwd_acc3", "fs_upsample_warp3d_pair_bwd3",
                  "fs_upsample_warp3d_pair_bwd3"]


def last_step(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    ends = [i for i, r in enumerate(rows) if "multi_tensor" in r["Kernel_Name"]]  # fused AdamW ends a step
    marks, prev = [], None
    for i in ends:
        if prev is None or i - prev > 50:
            marks.append(i)
        prev = i
    seg = rows[marks[-2]:marks[-1]] if len(marks) >= 2 else rows
    out = collections.defaultdict(lambda: [0.0, 0])
    nwarp = 0
    adjoint_of = None  # the fused entry point whose warp launch has just run: its adjoint launches follow directly
    for r in seg:
        name, val = r["Kernel_Name"], float(r["Counter_Value"])
        for sym in SYMBOLS:
            if sym in name:
                out["symbol:" + sym][0] += val
                out["symbol:" + sym][1] += 1
        if "warp3d_bwd_kernel" in name or "warp3d_rc_kernel<true" in name:
            ep = WARP_BWD_ORDER[nwarp % 4]
            nwarp += 1
            out[ep][0] += val
            out[ep][1] += 1
            adjoint_of = ep if ep == "fs_upsample_warp3d_pair_bwd3" else None
            continue
        if adjoint_of is not None and any(a in name for a in ADJOINT):
            out[adjoint_of][0] += val  # (not a launch of the entry point: one call = the warp + its adjoint passes)
            continue
        adjoint_of = None
        for sub, ep, counts in RULES:
            if sub in name:
                out[ep][0] += val
                out[ep][1] += 1 if counts else 0
                break
    return out


def main():
    fetch = last_step(sys.argv[1], "FETCH_SIZE")
    write = last_step(sys.argv[2], "WRITE_SIZE")
    bench = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    kern, syms = {}, {}
    for ep in sorted(set(fetch) | set(write)):
        n = max(fetch[ep][1], write[ep][1], 1)
        if ep.startswith("symbol:"):
            f_kb, w_kb = fetch[ep][0] / n, write[ep][0] / n
            syms[ep[7:]] = {"launches_per_step": n, "FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb,
                            "hbm_bytes_corrected": (2.0 * f_kb + w_kb) * 1024.0}
            continue
        f_kb, w_kb = fetch[ep][0] / n, write[ep][0] / n
        rec = {"launches_per_step": n, "FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb,
               "hbm_bytes_corrected": (2.0 * f_kb + w_kb) * 1024.0}
        b = bench.get("kernels", {}).get(ep)
        if b:
            rec["algorithmic_bytes"] = b["algo_GBps"] * 1e9 * b["avg_ms"] * 1e-3
            rec["ratio"] = round(rec["hbm_bytes_corrected"] / rec["algorithmic_bytes"], 3)
        kern[ep] = rec
    json.dump({"workload": "bench.py default: %s; per launch of the entry point, last step of each pass"
               % bench["config"]["workload"],
               "correction": "FETCH_SIZE x2 (gfx950 half-count of wide coalesced reads), WRITE_SIZE x1, KB units",
               "kernel_sources_sha256": kernel_sources_sha256(),
               "kernels": kern, "symbols": syms}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
