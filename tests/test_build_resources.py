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


def _check(usage, needle, max_vgprs):
    hits = {k: v for k, v in usage.items() if needle in k}
    assert hits, "no kernel matching %r in %s" % (needle, sorted(usage)[:5])
    for name, u in hits.items():
        assert u.get("ScratchSize [bytes/lane]") == 0 and u.get("VGPRs Spill") == 0, (name, u)
        assert u["VGPRs"] <= max_vgprs, (name, u)
        assert u.get("LDS Size [bytes/block]", 0) <= 160 * 1024, (name, u)


def test_hand_counted_wait_kernels_do_not_spill():
    fwd = _resource_usage("convfwd.hip")
    _check(fwd, "conv3d_wino2d_ps_kernel", 256)       # loader waves: inline-asm row loads + LDS-DMA slabs, vmcnt(21/39/12/9)
    _check(fwd, "conv3d_fwd_s3_kernel", 168)          # round 5: inline-asm input pieces, vmcnt(NWW + 2 PASSES); 12 waves per CU
    w3 = _resource_usage("warp3d.hip", ["-fno-slp-vectorize"])
    _check(w3, "warp3d_fwd_ring_kernel", 128)         # mover waves: counted vmcnt over LDS-DMA flow tiles
    _check(w3, "warp3d_rc_kernel", 168)               # round 5: counted vmcnt over tile DMA, row DMA, addend loads
