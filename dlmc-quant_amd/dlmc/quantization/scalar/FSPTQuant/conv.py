from .layers import FSPTQConv2d  # noqa: F401
