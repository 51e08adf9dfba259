"""Drop-in for Flow-2D/model/IFNet.py (`IFBlock`, `IFNet`)."""
from ... import ifnet as _g
from .warplayer import warp  # noqa: F401


class IFBlock(_g.IFBlock):
    def __init__(self, in_planes, c=64):
        super().__init__(2, in_planes, c)


class IFNet(_g.IFNet):
    def __init__(self):
        super().__init__(2)
