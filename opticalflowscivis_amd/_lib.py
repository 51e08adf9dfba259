"""ctypes binding of libflowsci_hip.so (the C-ABI declared in include/flowsci_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  `lib()` raises
`FlowsciLibraryError` when the shared object is missing or a symbol is absent, and every op in
`opticalflowscivis_amd.ops` goes through `lib()`.
"""
import ctypes
import os

# torch MUST be imported before the library is dlopen()ed: PyTorch-ROCm ships its own libamdhip64 and
# the first HIP runtime loaded into the process wins the soname.  If libflowsci_hip.so pulled in the
# system runtime first, torch would bind to it and report no GPU.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB_PATH = os.path.join(_HERE, "csrc", "libflowsci_hip.so")
# FLOWSCI_HIP_LIBRARY=<path>: load another build of the same sources instead -- the ablation build
# (`make -C opticalflowscivis_amd/csrc ablation` -> csrc/ablation/libflowsci_hip_ab.so), in which superseded kernels and
# the FLOWSCI_* measurement switches exist.  The product library reads no environment variable, and the package's own
# A/B switches (`ablation_env`) are honoured only in this mode; bench.py reports the mode in its JSON line.
LIB_PATH = os.environ.get("FLOWSCI_HIP_LIBRARY") or PRODUCT_LIB_PATH
ABLATION = LIB_PATH != PRODUCT_LIB_PATH


def ablation_env(name):
    """Value of a FLOWSCI_* measurement switch, or None outside ablation mode (the switch then does not exist)."""
    return os.environ.get(name) if ABLATION else None


def switches():
    """What in the environment can change dispatch or numerics: the library in use and every FLOWSCI_* variable set
    (INTEGRATION.md "Switches" says which of them are read in which mode)."""
    return {"library": "ablation build: " + LIB_PATH if ABLATION else "product",
            "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("FLOWSCI_")}}

ABI_VERSION = 330  # FS_ABI_VERSION of the include/flowsci_hip.h the SIGNATURES below were written against

_f32p = ctypes.c_void_p  # device pointers travel as integers
_int = ctypes.c_int
_float = ctypes.c_float
_stream = ctypes.c_void_p
_intp = ctypes.POINTER(ctypes.c_int)  # host array of ints (or None)
_i64p = ctypes.POINTER(ctypes.c_longlong)  # host out-parameter
_i64 = ctypes.c_longlong
_ptrv = ctypes.POINTER(ctypes.c_void_p)  # host array of device pointers

class FsWprepJob(ctypes.Structure):
    """include/flowsci_hip.h `FsWprepJob`: one weight re-layout of fs_conv3d_wprep_batch."""
    _fields_ = [("w", ctypes.c_void_p), ("ws", ctypes.c_void_p), ("kind", ctypes.c_int), ("total", ctypes.c_int),
                ("p", ctypes.c_int * 6)]


_jobp = ctypes.POINTER(FsWprepJob)

# name -> argtypes; restype is int unless listed in _RESTYPES
SIGNATURES = {
    "fs_version": [],
    "fs_error_string": [_int],
    "fs_warp3d_fwd": [_f32p] * 3 + [_int, _int, _intp, _int, _int, _int, _stream],
    "fs_warp3d_bwd": [_f32p] * 5 + [_int, _int, _intp, _int, _int, _int, _stream],
    "fs_warp3d_pair_fwd": [_f32p] * 5 + [_int, _int, _intp, _int, _int, _int, _stream],
    "fs_warp3d_pair_bwd": [_f32p] * 8 + [_int, _int, _intp, _int, _int, _int, _stream],
    "fs_warp3d_pair_bwd_acc": [_f32p] * 9 + [_int, _int, _intp, _int, _int, _int, _stream],
    "fs_warp3d_pair_bwd_acc3": [_f32p] * 3 + [_f32p, _i64, _f32p, _i64] + [_f32p] * 2 + [_f32p, _i64, _f32p, _i64, _f32p, _i64, _f32p, _int, _int, _intp, _int, _int,
                                _int, _stream],
    "fs_upsample_warp3d_pair_bwd3": [_f32p] * 3 + [_f32p, _i64, _f32p, _i64] + [_f32p, _i64, _f32p, _i64, _f32p, _i64] + [_f32p] * 3 +
                                    [_int, _int, _intp, _int, _int, _int, _int, _float, _stream],
    "fs_upsample_warp3d_pair_fwd": [_f32p] * 7 + [_int, _int, _intp, _int, _int, _int, _int, _float, _stream],
    "fs_upsample_warp3d_pair_bwd": [_f32p] * 9 + [_int, _int, _intp, _int, _int, _int, _int, _float, _stream],
    "fs_warp2d_pair_fwd": [_f32p] * 5 + [_int, _int, _intp, _int, _int, _int, _stream],
    "fs_warp2d_pair_bwd": [_f32p] * 8 + [_int, _int, _intp, _int, _int, _int, _stream],
    "fs_interp3d_bwd_scaled": [_f32p] * 3 + [_int] * 10 + [_float, _stream],
    "fs_downsample3d_fwd": [_f32p] * 2 + [_int] * 6 + [_float, _stream],
    "fs_downsample3d_fwd_ms": [_ptrv, _i64p, _f32p] + [_int] * 6 + [_float, _stream],
    "fs_resize2d_fwd": [_f32p] * 2 + [_int] * 8 + [_float, _stream],
    "fs_resize2d_bwd": [_f32p] * 2 + [_int] * 8 + [_float, _stream],
    "fs_occ_check2d": [_f32p] * 4 + [_int] * 3 + [_float, _float, _int, _stream],
    "fs_laploss2d_sizes": [_int] * 4 + [_i64p] * 3,
    "fs_laploss2d_fwd": [_f32p] * 5 + [_int] * 4 + [_stream],
    "fs_laploss2d_bwd": [_f32p] * 4 + [_int] * 4 + [_stream],
    "fs_laploss3d_sizes": [_int] * 5 + [_i64p] * 3,
    "fs_laploss3d_fwd": [_f32p] * 5 + [_int] * 5 + [_stream],
    "fs_laploss3d_bwd": [_f32p] * 4 + [_int] * 5 + [_stream],
    "fs_conv3d_fwd_ws_floats": [_int] * 3,
    "fs_conv3d_fwd": [_f32p] * 5 + [_int] * 13 + [_stream],
    "fs_upsample3d_scale_add": [_f32p] * 3 + [_int] * 6 + [_float, _stream],
    "fs_conv3d_fwd_dprelu_part_floats": [_int] * 5,
    "fs_conv3d_fwd_dprelu_part_floats_k3": [_int] * 5,
    "fs_conv3d_fwd_dprelu": [_f32p] * 4 + [_int] + [_f32p] * 5 + [_int] * 12 + [_stream],
    "fs_conv3d_fwd_prelu": [_f32p] * 8 + [_int] * 13 + [_stream],
    "fs_conv3d_fwd_add": [_f32p] * 6 + [_int] * 13 + [_stream],
    "fs_conv3d_tr_prelu": [_f32p] * 7 + [_int] * 10 + [_stream],
    "fs_conv3d_tr_add": [_f32p] * 6 + [_int] * 9 + [_stream],
    "fs_conv3d_tr_ws_floats": [_int] * 2,
    "fs_conv3d_tr": [_f32p] * 5 + [_int] * 9 + [_stream],
    "fs_plane_moments": [_f32p] * 2 + [_int] * 2 + [_stream],
    "fs_plane_norm_bwd": [_f32p] * 4 + [_int] * 2 + [_stream],
    "fs_corr2d_norm_fwd": [_f32p] * 5 + [_int] * 5 + [_stream],
    "fs_corr2d_norm_bwd": [_f32p] * 7 + [_int] * 5 + [_stream],
    "fs_corr2d_pair_fwd": [_f32p] * 7 + [_int] * 5 + [_stream],
    "fs_corr2d_pair_bwd": [_f32p] * 11 + [_int] * 5 + [_stream],
    "fs_plane_moments4": [_f32p] * 5 + [_int] * 2 + [_stream],
    "fs_plane_norm_bwd4": [_f32p] * 13 + [_int] * 2 + [_stream],
    "fs_corr2d_fwd": [_f32p] * 3 + [_int] * 5 + [_stream],
    "fs_corr2d_bwd": [_f32p] * 5 + [_int] * 5 + [_stream],
    "fs_robust_sum": [_f32p] * 5 + [_int] * 7 + [_float, _float, _stream],
    "fs_robust_sum_bwd": [_f32p] * 6 + [_int] * 7 + [_float, _float, _stream],
    "fs_census_dist_fwd": [_f32p] * 3 + [_int] * 4 + [_stream],
    "fs_census_dist_bwd": [_f32p] * 5 + [_int] * 4 + [_stream],
    "fs_merge_fwd": [_f32p] * 5 + [_int] * 3 + [_stream],
    "fs_merge_bwd": [_f32p] * 8 + [_int] * 3 + [_stream],
    "fs_distill3_fwd": [_f32p] * 11 + [_int] * 4 + [_stream],
    "fs_distill3_bwd": [_f32p] * 13 + [_int] * 4 + [_stream],
    "fs_distill_fwd": [_f32p] * 7 + [_int] * 4 + [_stream],
    "fs_distill_bwd": [_f32p] * 7 + [_int] * 4 + [_stream],
    "fs_interp3d_bwd": [_f32p] * 3 + [_int] * 10 + [_stream],
    "fs_prelu_bwd": [_f32p] * 7 + [_int] * 4 + [_stream],
    "fs_corr3d_fwd": [_f32p] * 3 + [_int] * 6 + [_stream],
    "fs_corr3d_bwd": [_f32p] * 5 + [_int] * 6 + [_stream],
    "fs_conv3d_wrw": [_f32p] * 3 + [_int] * 12 + [_stream],
    "fs_conv3d_wrw_kernel_id": [_f32p] * 2 + [_int] * 12,
    "fs_warp3d_kernel_id": [_f32p] * 3 + [_int, _int, _intp] + [_int] * 5,
    "fs_conv3d_wrw_det_ws_floats": [_f32p, _f32p, _ptrv, _i64p] + [_int] * 12,
    "fs_conv3d_wrw_det": [_f32p, _f32p, _ptrv, _i64p, _f32p, _f32p, _i64] + [_int] * 12 + [_stream],
    "fs_conv3d_fwd_prelu_ms": [_ptrv, _i64p] + [_f32p] * 6 + [_int] * 13 + [_stream],
    "fs_conv3d_wrw_ms": [_f32p, _ptrv, _i64p, _f32p] + [_int] * 12 + [_stream],
    "fs_wssim_fwd": [_f32p] * 5 + [_int] * 5 + [_stream],
    "fs_wssim_bwd": [_f32p] * 6 + [_int] * 5 + [_stream],
    "fs_conv3d_fwd_wprep_jobs": [_jobp, _int, _f32p, _f32p, _f32p] + [_int] * 13,
    "fs_conv3d_tr_wprep_jobs": [_jobp, _int, _f32p, _f32p, _f32p] + [_int] * 10,
    "fs_conv3d_wprep_batch": [_f32p, _int, _stream],
    "fs_warp2d_fwd": [_f32p, _f32p, _f32p, _f32p, _int, _int, _intp, _int, _int, _int, _int, _stream],
    "fs_warp2d_bwd": [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _intp, _int, _int, _int, _int,
                      _stream],
}
_RESTYPES = {"fs_error_string": ctypes.c_char_p, "fs_conv3d_fwd_ws_floats": ctypes.c_longlong,
             "fs_conv3d_wrw_det_ws_floats": ctypes.c_longlong,
             "fs_conv3d_tr_ws_floats": ctypes.c_longlong, "fs_conv3d_fwd_dprelu_part_floats": ctypes.c_longlong,
             "fs_conv3d_fwd_dprelu_part_floats_k3": ctypes.c_longlong}


class FlowsciLibraryError(RuntimeError):
    pass


class FlowsciKernelError(RuntimeError):
    pass


_lib = None


def lib():
    """Load (once) and return the ctypes handle; fail loudly if the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FlowsciLibraryError(
            "libflowsci_hip.so not found at %s -- build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C opticalflowscivis_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    handle = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise FlowsciLibraryError("libflowsci_hip.so does not export %s" % name) from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, _int)
    got = handle.fs_version()
    if got != ABI_VERSION:
        raise FlowsciLibraryError(
            "libflowsci_hip.so at %s reports ABI version %d, this binding was written for %d: argument lists "
            "differ between them (include/flowsci_hip.h, FS_ABI_VERSION) -- rebuild with "
            "`make -C opticalflowscivis_amd/csrc`" % (LIB_PATH, got, ABI_VERSION))
    _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().fs_error_string(code)
        raise FlowsciKernelError("%s failed: %s (code %d)" % (what, msg.decode() if msg else "?", code))
