from .layers import QLinear  # noqa: F401  (reference import path: modules/linear.py)
