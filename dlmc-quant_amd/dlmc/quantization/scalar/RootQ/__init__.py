from .base import RootQBase
from .layers import RootQConv2d, RootQLinear

__all__ = ["RootQBase", "RootQConv2d", "RootQLinear"]
