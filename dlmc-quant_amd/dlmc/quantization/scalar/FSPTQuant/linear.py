from .layers import FSPTQLinear  # noqa: F401
