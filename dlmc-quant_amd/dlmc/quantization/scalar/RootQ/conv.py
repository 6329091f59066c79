from .layers import RootQConv2d  # noqa: F401
