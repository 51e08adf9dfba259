"""Drop-in for Flow-3D/model/RIFE.py (`Model`)."""
from ...rife import Model3D as Model  # noqa: F401
from .warplayer import warp  # noqa: F401
