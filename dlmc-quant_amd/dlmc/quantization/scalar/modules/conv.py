from .layers import QConv2d  # noqa: F401  (reference import path: modules/conv.py)
