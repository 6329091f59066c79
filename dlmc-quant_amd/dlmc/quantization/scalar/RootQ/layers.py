"""RootQConv2d / RootQLinear (reference: RootQ/conv.py, RootQ/linear.py)."""
from torch.nn import Conv2d, Linear

from .._wrapper import conv_forward, linear_forward
from .base import RootQBase


class RootQConv2d(RootQBase, Conv2d):
    def __init__(self, *args, qconfig=None, **kwargs):
        Conv2d.__init__(self, *args, **kwargs)
        RootQBase.__init__(self, qconfig)

    def _forward_func(self, input, weight):
        return conv_forward(self, input, weight)


class RootQLinear(RootQBase, Linear):
    def __init__(self, *args, qconfig=None, **kwargs):
        Linear.__init__(self, *args, **kwargs)
        RootQBase.__init__(self, qconfig)

    def _forward_func(self, input, weight):
        return linear_forward(self, input, weight)
