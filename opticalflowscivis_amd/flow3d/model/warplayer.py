"""Drop-in for Flow-3D/model/warplayer.py: `warp(tenInput, tenFlow)` on the HIP kernel."""
from ... import ops


def warp(tenInput, tenFlow):
    """Flow-3D/model/warplayer.py:9-41 -- trilinear, border, align_corners=True, axis-rotating grid.
    The reference's `backwarp_tenGrid` cache (:5,11-22) has no equivalent: the grid is computed
    in-kernel."""
    return ops.warp3d(tenInput, tenFlow)
