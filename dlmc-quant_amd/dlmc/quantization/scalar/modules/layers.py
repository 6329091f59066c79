"""QConv2d / QLinear: QBase mixed into the torch layer it replaces (reference: modules/conv.py,
modules/linear.py).  `quantize_model` builds them with `__new__` + `__dict__.update` + `initialize`;
unlike the reference they can also be constructed directly."""
from torch.nn import Conv2d, Linear

from .._wrapper import conv_forward, linear_forward
from .base import QBase


class QConv2d(QBase, Conv2d):
    def __init__(self, *args, qconfig=None, **kwargs):
        Conv2d.__init__(self, *args, **kwargs)
        QBase.__init__(self, qconfig)

    def _forward_func(self, input, weight):
        return conv_forward(self, input, weight)


class QLinear(QBase, Linear):
    def __init__(self, *args, qconfig=None, **kwargs):
        Linear.__init__(self, *args, **kwargs)
        QBase.__init__(self, qconfig)

    def _forward_func(self, input, weight):
        return linear_forward(self, input, weight)
