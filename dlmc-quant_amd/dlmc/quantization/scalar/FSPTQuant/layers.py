"""FSPTQConv2d / FSPTQLinear (reference: FSPTQuant/conv.py, FSPTQuant/linear.py)."""
from torch.nn import Conv2d, Linear

from .._wrapper import conv_forward, linear_forward
from .base import FSPTQBase


class FSPTQConv2d(FSPTQBase, Conv2d):
    def __init__(self, *args, qconfig=None, **kwargs):
        Conv2d.__init__(self, *args, **kwargs)
        FSPTQBase.__init__(self, qconfig)

    def _forward_func(self, input, weight):
        return conv_forward(self, input, weight)


class FSPTQLinear(FSPTQBase, Linear):
    def __init__(self, *args, qconfig=None, **kwargs):
        Linear.__init__(self, *args, **kwargs)
        FSPTQBase.__init__(self, qconfig)

    def _forward_func(self, input, weight):
        return linear_forward(self, input, weight)
