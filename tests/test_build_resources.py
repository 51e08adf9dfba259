"""CPU (no GPU needed): compile-time guards for the kernels whose correctness rests on hand-counted `s_waitcnt vmcnt(N)`
around inline-assembly / LDS-DMA loads: the waits only hold while the compiler emits no vector-memory
instruction of its own between a load and its wait -- a scratch spill would be one.  hipcc cross-compiles gfx950 here;
`-Rpass-analysis=kernel-resource-usage` prints every kernel's registers / scratch / spills."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "opticalflowscivis_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _resource_usage(src, extra=()):
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("needs hipcc")
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "-c", os.path.join(CSRC, src), "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only"] + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    return out


def _check(usage, needle, max_vgprs, scratch_ok=False):
    hits = {k: v for k, v in usage.items() if needle in k}
    assert hits, "no kernel matching %r in %s" % (needle, sorted(usage)[:5])
    for name, u in hits.items():
        if not scratch_ok:
            assert u.get("ScratchSize [bytes/lane]") == 0 and u.get("VGPRs Spill") == 0, (name, u)
        assert u["VGPRs"] <= max_vgprs, (name, u)
        assert u.get("LDS Size [bytes/block]", 0) <= 160 * 1024, (name, u)


def _loader_regions(src, needle, extra=()):
    """ISA of the loader-wave section (from its `s_setprio 3` to the next `s_endpgm`) of every kernel whose mangled name
    contains `needle`."""
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", "-"] + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out, name, body = {}, None, []
    for line in r.stdout.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = (m.group(1) if needle in m.group(1) else None), []
            continue
        if name is None:
            continue
        body.append(line)
        if "s_endpgm" in line and any("s_setprio 3" in b for b in body):
            start = max(i for i, b in enumerate(body) if "s_setprio 3" in b)
            out[name] = body[start:]
            name = None
    return out


def test_hand_counted_wait_kernels_do_not_spill():
    fwd = _resource_usage("convfwd.hip")
    _check(fwd, "conv3d_wino2d_ps_kernel", 256)       # loader waves: inline-asm row loads + LDS-DMA slabs, vmcnt(21/39/12/9)
    # round 5: inline-asm input pieces with counted waits.  64 channels: 12 waves per CU (168 registers), no scratch at all;
    # 32 channels: TWO workgroups per CU (80 registers) -- its EPILOGUE spills a few address registers (matrix waves, after the
    # last barrier), which is harmless, but the loader section must stay free of compiler-made memory traffic
    _check({k: v for k, v in fwd.items() if "s3_kernelILi2" in k}, "conv3d_fwd_s3_kernel", 168)
    _check({k: v for k, v in fwd.items() if "s3_kernelILi1" in k}, "conv3d_fwd_s3_kernel", 80, scratch_ok=True)
    regions = _loader_regions("convfwd.hip", "conv3d_fwd_s3_kernel")
    assert len(regions) >= 2, sorted(regions)
    for name, isa in regions.items():
        bad = [l for l in isa if re.search(r"\bscratch_|\bglobal_load|\bglobal_store|\bflat_", l)]
        assert not bad, (name, bad[:4])
    w3 = _resource_usage("warp3d.hip", ["-fno-slp-vectorize"])
    _check(w3, "warp3d_fwd_ring_kernel", 128)         # mover waves: counted vmcnt over LDS-DMA flow tiles
    _check(w3, "warp3d_rc_kernel", 168)               # round 5: counted vmcnt over tile DMA, row DMA, addend loads


def _device_asm(src, tmp_path, extra=()):
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("needs hipcc")
    out = str(tmp_path / (src.replace(".hip", "") + ".s"))
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out] + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return out


def test_split_bf16_loaders_touch_no_register_or_address_in_flight(tmp_path):
    """Round 5 (found while the transposed split-bf16 kernel was brought up): to the compiler the destination
    of an inline-assembly load is an ordinary value -- it may copy it in front of the hand-counted wait (stale data) and hand
    the register to another value, which the load then overwrites when it lands; silent on a warm cache.  So the compiled
    loader code of the kernels that keep such loads in flight (forward split-bf16; the transposed one stages through LDS
    instead) must neither READ nor WRITE a register between its load and the wait that covers it -- checked on the device
    assembly of THIS build (scripts/check_inflight_regs.py simulates the request queue).  Also kept: no write to the address
    register of an LDS-DMA copy in flight (a precaution, not a requirement: scripts/micro/lds_dma_hazards.hip), and the
    transposed kernel's loaders free of compiler-made scratch traffic (it would be counted by their vmcnt waits)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import check_inflight_regs as chk
    fwd = _device_asm("convfwd.hip", tmp_path)
    assert chk.check(fwd, "conv3d_fwd_s3_kernel") == 0
    assert chk.check_dma_addr(fwd, "conv3d_fwd_s3_kernel") == 0
    # the 2-D Winograd trunk kernel's loaders keep inline-assembly row loads in flight as well: clean up to reads of a
    # DON'T-CARE word (the high half of a 64-bit addend of v_mad_u64_u32 whose low 32 bits alone are used)
    seen = []
    chk.check(fwd, "conv3d_wino2d_ps_kernel", seen)
    assert all(t.startswith("v_mad_u64_u32") and "WRITE" not in t for t in seen), seen
    tr = _device_asm("convtr.hip", tmp_path)
    assert chk.check(tr, "convtr_s3_kernel") == 0
    assert chk.check_dma_addr(tr, "convtr_s3_kernel") == 0
    regions = _loader_regions("convtr.hip", "convtr_s3_kernel")
    assert len(regions) == 2, sorted(regions)  # the 32-row and the 16-row instantiation
    for name, isa in regions.items():
        bad = [l for l in isa if re.search(r"\bscratch_|\bglobal_load|\bglobal_store|\bflat_", l)]
        assert not bad, (name, bad[:4])
    usage = _resource_usage("convtr.hip")
    _check(usage, "convtr_s3_kernel", 168, scratch_ok=True)  # (a few epilogue address registers spill: matrix waves only)
