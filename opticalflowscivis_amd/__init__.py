"""opticalflowscivis_amd -- MI355X-native hot path of HamidGadirov/OpticalFlowSciVis.

Backward warps (2-D / 3-D), local-window correlation and the photometric / census losses as
hand-written HIP kernels for gfx950 behind a C-ABI (include/flowsci_hip.h), plus the Python
mirrors of the reference's call sites (`flow2d`, `flow3d`, `upflow` sub-packages).
"""
__version__ = "0.1.0"
