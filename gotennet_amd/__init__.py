"""gotennet_amd -- MI355X-native (gfx950) implementation of the GotenNet interaction hot path.

Public surface mirrors the reference package (gotennet/__init__.py:5-10)."""
from .gotennet import EQFF, GATA, GotenNet, GotenNetWrapper  # noqa: F401
from .layers import CosineCutoff  # noqa: F401

__version__ = "0.1.0"
