from .base import FSPTQBase
from .layers import FSPTQConv2d, FSPTQLinear

__all__ = ["FSPTQBase", "FSPTQConv2d", "FSPTQLinear"]
