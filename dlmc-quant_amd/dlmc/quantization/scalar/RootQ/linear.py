from .layers import RootQLinear  # noqa: F401
