"""Drop-in for Flow-3D/model/IFNet.py (`IFBlock`, `IFNet`)."""
from ... import ifnet as _g
from .warplayer import warp  # noqa: F401  (the reference re-exports it: IFNet.py:4)


class IFBlock(_g.IFBlock):
    def __init__(self, in_planes, c=64):
        super().__init__(3, in_planes, c)


class IFNet(_g.IFNet):
    def __init__(self):
        super().__init__(3)
