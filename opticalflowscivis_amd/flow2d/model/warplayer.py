"""Drop-in for Flow-2D/model/warplayer.py: `warp(tenInput, tenFlow)` on the HIP kernel."""
from ... import ops


def warp(tenInput, tenFlow):
    """Flow-2D/model/warplayer.py:7-26 -- bilinear, border, align_corners=True."""
    return ops.warp2d(tenInput, tenFlow)
