"""CPU port of the reference's quantised layer forward, for `bench.py`'s cpu_baseline leg and tests.
TEST / BASELINE INFRASTRUCTURE ONLY - never imported by the product package.

`port_model(model, family)` replaces every Conv2d / Linear of a CPU model by a wrapper that runs the
reference's op sequence (through oracle/fakequant_oracle.py) - observer on the first call, then
fake-quant(input), fake-quant(weight), conv/linear - i.e. what the reference's FSPTQConv2d / QConv2d do on
the CPU (FSPTQuant/base.py:95-159, modules/base.py:67-140)."""
import math

import torch
from torch import nn

from . import fakequant_oracle as O


class PortedLayer(nn.Module):
    def __init__(self, layer, family, w_bits=8, a_bits=8, a_signed=False):
        super().__init__()
        self.layer, self.family = layer, family
        self.w_rng = O.qrange(True, w_bits)
        self.a_rng = O.qrange(a_signed, a_bits)
        self.a_signed, self.w_bits, self.a_bits = a_signed, w_bits, a_bits
        self.ready = False

    def forward(self, x):
        w = self.layer.weight.detach()
        if not self.ready:
            self.in_scale, self.in_off = O.minmax_tensor(x, self.a_bits, self.a_signed)
            if self.family == "FSPTQ":
                s, _ = O.minmax_channel(w, self.w_bits, True, ch_axis=0)
                self.wt_scale, self.wt_off = s + 1e-6, None
            else:
                self.wt_scale, self.wt_off = O.minmax_tensor(w, self.w_bits, True)
            self.ready = True
        if self.family == "FSPTQ":
            xq = O.fq_zeropoint(x, self.in_scale, self.in_off, *self.a_rng)[1]
            wq = O.fq_symmetric(w, self.wt_scale, *self.w_rng)[1]
        else:
            xq = O.fq_qbase(x, self.in_scale, self.in_off, *self.a_rng, 1 / math.sqrt(x.numel() * self.a_rng[1]))[1]
            wq = O.fq_qbase(w, self.wt_scale, self.wt_off, *self.w_rng, 1 / math.sqrt(w.numel() * self.w_rng[1]))[1]
        return O.conv_or_linear(self.layer, xq, wq)


def port_model(model, family="FSPTQ", a_signed=False):
    for name, child in list(model.named_children()):
        if isinstance(child, (nn.Conv2d, nn.Linear)):
            setattr(model, name, PortedLayer(child, family, a_signed=a_signed))
        else:
            port_model(child, family, a_signed)
    return model
