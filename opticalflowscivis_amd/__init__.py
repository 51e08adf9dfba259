"""opticalflowscivis_amd -- MI355X-native hot path of HamidGadirov/OpticalFlowSciVis.

Backward warps (2-D / 3-D), local-window correlation and the photometric / census losses as
hand-written HIP kernels for gfx950 behind a C-ABI (include/flowsci_hip.h), plus the Python
mirrors of the reference's call sites (`flow2d`, `flow3d`, `upflow` sub-packages).
"""
__version__ = "0.1.0"

import os as _os

# MIOpen has no gfx950 find-db in ROCm 7.2, so the first call of every convolution shape runs a
# "find" that also times its naive direct kernels -- 1.4 s per call for a 128^3 IFNet-3D layer,
# minutes per step at 256^3; 63 ms x 656 calls = 42 s of `naive_conv_ab_nonpacked_wrw` before the
# first UPFlow step at C3 -- although they never win (the 3-D layers run on this package's own
# kernels, csrc/conv{fwd,tr,wrw}.hip; for the 2-D nets the igemm / Winograd solvers are picked with
# or without them: C2 13.6 ms and C3 88 ms per step either way).  The package's entry points must
# start in bounded time, so the naive solvers are taken out of the search unless the user has already
# chosen otherwise.  This is PROCESS-WIDE (it changes MIOpen's solver search for every model in the process):
# `FLOWSCI_KEEP_MIOPEN_DEFAULTS=1` in the environment before the import leaves MIOpen untouched, and any of the
# three variables set by the user wins.
if _os.environ.get("FLOWSCI_KEEP_MIOPEN_DEFAULTS") != "1":
    for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
               "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
        _os.environ.setdefault(_k, "0")
    del _k
