"""CPU: the C-ABI library loads and exports every symbol include/flowsci_hip.h declares; the Python
binding covers them all; the product path fails loudly instead of falling back to the CPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "flowsci_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fs_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from opticalflowscivis_amd import _lib
    names = _declared()
    assert len(names) >= 20
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), "libflowsci_hip.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names, "Python binding and header disagree"
    lib = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "flowsci_hip.h")).read()
    abi = int(re.search(r"#define\s+FS_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert lib.fs_version() == abi == _lib.ABI_VERSION  # header == library == binding, exactly
    assert lib.fs_error_string(0) == b"ok" and lib.fs_error_string(2) != b"ok"


def test_null_and_shape_errors_without_gpu():
    """Argument validation happens before any launch, so it is testable on the CPU."""
    from opticalflowscivis_amd import _lib
    lib = _lib.lib()
    assert lib.fs_warp3d_fwd(None, None, None, 1, 1, None, 8, 8, 8, None) == 1      # NULLPTR
    assert lib.fs_warp3d_fwd(1, 1, 1, 1, 1, None, 1, 8, 8, None) == 2               # SHAPE (D < 2)
    assert lib.fs_warp2d_fwd(1, 1, None, 1, 1, 1, None, 8, 8, 7, 0, None) == 3            # ARG (mode)
    assert lib.fs_corr2d_fwd(1, 1, 1, 1, 1, 8, 8, 5, None) == 3                     # ARG (md > 4)
    assert lib.fs_census_dist_fwd(1, 1, 1, 1, 8, 8, 2, None) == 3                   # ARG (md != 3)
    with pytest.raises(_lib.FlowsciKernelError):
        _lib.check(2, "x")


def test_ops_refuse_cpu_tensors():
    from opticalflowscivis_amd import ops
    x = torch.rand(1, 1, 8, 8, 8)
    with pytest.raises(ValueError, match="no CPU fallback"):
        ops.warp3d(x, torch.zeros(1, 3, 8, 8, 8))
    with pytest.raises(ValueError):
        ops.corr2d(torch.rand(1, 4, 8, 8), torch.rand(1, 4, 8, 8))
    with pytest.raises(ValueError):
        ops.census_loss(torch.rand(1, 3, 16, 16), torch.rand(1, 3, 16, 16), torch.ones(1, 1, 16, 16), 0.4,
                        False, False)


def test_missing_library_fails_loudly(monkeypatch):
    from opticalflowscivis_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libflowsci_hip.so")
    with pytest.raises(_lib.FlowsciLibraryError, match="no CPU fallback"):
        _lib.lib()


def test_stale_library_is_refused(monkeypatch):
    """A library whose ABI version differs from the binding's must not be used (shifted arguments)."""
    from opticalflowscivis_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 100)
    with pytest.raises(_lib.FlowsciLibraryError, match="ABI version"):
        _lib.lib()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "opticalflowscivis_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(d, f)


def test_model_requires_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from opticalflowscivis_amd.flow3d.model.RIFE import Model
    with pytest.raises(RuntimeError, match="no CPU"):
        Model(local_rank=-1)


def test_weight_relayout_plans_without_gpu():
    """fs_conv3d_{fwd,tr}_wprep_jobs run the entry points' own dispatch with nothing launched: host-only, so the
    layouts they pick for the IFNet-3D layer shapes are checkable here (slab sizes == fs_conv3d_*_ws_floats)."""
    from opticalflowscivis_amd import _lib
    L = _lib.lib()
    buf = (_lib.FsWprepJob * 4)()
    assert ctypes.sizeof(_lib.FsWprepJob) == 48
    fwd = lambda cin, cout, n, k, wmode, x=0x4000, w=0x1000: L.fs_conv3d_fwd_wprep_jobs(
        buf, 4, x, w, 0x2000, 2, cin, cout, n, n, n, n if k == 3 else n // 2, n if k == 3 else n // 2,
        n if k == 3 else n // 2, k, 1 if k == 3 else 2, 1, wmode)
    # the 64-channel k3 layers of the 64^3 trunk take the 2-D Winograd (F(2,3) along y x F(4,3) along x) filter slab
    # (kind 6), in both weight modes; so does any volume with one 2 x 2 x 64 brick per CU
    for wmode in (0, 1):
        assert fwd(64, 64, 64, 3, wmode) == 1 and buf[0].kind == 6 and buf[0].w == 0x1000 and buf[0].ws == 0x2000
        assert buf[0].total == 64 * (3 * 24 * 64) == L.fs_conv3d_fwd_ws_floats(64, 64, 3)
    half = L.fs_conv3d_fwd_wprep_jobs(buf, 4, 0x4000, 0x1000, 0x2000, 2, 64, 64, 16, 32, 64, 16, 32, 64, 3, 1, 1, 0)
    assert half == 1 and buf[0].kind == 6
    few = L.fs_conv3d_fwd_wprep_jobs(buf, 4, 0x4000, 0x1000, 0x2000, 1, 64, 64, 8, 8, 64, 8, 8, 64, 3, 1, 1, 0)
    assert few == 1 and buf[0].kind == 0   # 16 bricks: the direct kernel
    # ... not at 16^3 (rows of 16), not from a misaligned input: the direct taps (kind 0)
    assert fwd(64, 64, 16, 3, 0) == 1 and buf[0].kind == 0 and buf[0].total == 64 * 27 * 64
    assert fwd(64, 64, 32, 3, 0) == 1 and buf[0].kind == 6   # rows of 32: 2 x 4 x 32 bricks, one per CU at 2 x 32^3
    assert fwd(64, 64, 64, 3, 0, x=0x4004) == 1 and buf[0].kind == 0
    # k4 s2 layers: the fp32 taps (kind 0) on small volumes; from 256 bricks of 1 x 16 x 32 outputs on, the pre-split bf16
    # slab of the fp32-accurate bf16-rate kernel (round 5, kind 7: three 2-byte pieces per weight = 1.5 x the floats)
    assert fwd(11, 32, 64, 4, 0) == 1 and buf[0].kind == 0 and buf[0].total == 12 * 64 * 32 <= L.fs_conv3d_fwd_ws_floats(11, 32, 4)
    assert fwd(11, 32, 128, 4, 0) == 1 and buf[0].kind == 7 and buf[0].total == L.fs_conv3d_fwd_ws_floats(11, 32, 4) == 32 * 6 * 4 * 48
    assert fwd(32, 64, 128, 4, 0) == 1 and buf[0].kind == 7 and buf[0].total == L.fs_conv3d_fwd_ws_floats(32, 64, 4) == 64 * 16 * 4 * 48
    assert fwd(32, 64, 128, 4, 0, x=0x4004) == 1 and buf[0].kind == 0   # misaligned input: the fp32 kernels
    assert L.fs_conv3d_fwd_wprep_jobs(buf, 4, 0x4000, 0x1000, 0x2000, 1, 8, 8, 8, 8, 8, 4, 4, 4, 5, 1, 2, 0) == -3   # -FS_ERR_ARG
    assert fwd(8, 8, 8, 3, 0, w=None) == -1                                                                          # -FS_ERR_NULLPTR
    kinds = {}
    for cin, cout, di, z in ((64, 32, 64, 0), (32, 6, 128, 0), (32, 1, 128, 0), (32, 11, 128, 0), (128, 64, 16, 0),
                             (32, 6, 128, 1), (5, 3, 9, 0), (64, 32, 16, 0), (32, 11, 16, 0)):
        n = L.fs_conv3d_tr_wprep_jobs(buf, 4, 0x4000, 0x1000, 0x2000, 2, cin, cout, di, di, di, 2 * di, 2 * di, 2 * di, z)
        assert n >= 0
        assert sum(buf[i].total for i in range(n)) <= L.fs_conv3d_tr_ws_floats(cin, cout)
        kinds[(cin, cout, di, z)] = [buf[i].kind for i in range(n)]
    # round 5: the split-bf16 slabs where the 2 x 3 x 32-position bricks fill the chip (kind 8: 17..32 channels, 9: 7..16);
    # the fp32 class kernels' slabs below that (kind 1: 32-channel slices, 2: 16 rows)
    assert kinds[(64, 32, 64, 0)] == [8] and kinds[(128, 64, 16, 0)] == [1, 1] and kinds[(64, 32, 16, 0)] == [1]
    assert kinds[(32, 11, 128, 0)] == [9] and kinds[(32, 11, 16, 0)] == [2]
    assert kinds[(32, 6, 128, 0)] == [3] and kinds[(32, 1, 128, 0)] == [3]
    assert kinds[(32, 6, 128, 1)] == [] and kinds[(5, 3, 9, 0)] == []                      # kernels that read w as stored
    # a misaligned input rules out the loader-wave / all-parities kernels: another layout (or none) is planned
    n = L.fs_conv3d_tr_wprep_jobs(buf, 4, 0x4004, 0x1000, 0x2000, 2, 32, 6, 128, 128, 128, 256, 256, 256, 0)
    assert n == 0
    assert L.fs_conv3d_wprep_batch(None, 1, None) == 1 and L.fs_conv3d_wprep_batch(0x10, 0, None) == 2
