"""Drop-in for Flow-2D/model/RIFE.py (`Model`)."""
from ...rife import Model2D as Model  # noqa: F401
from .warplayer import warp  # noqa: F401
