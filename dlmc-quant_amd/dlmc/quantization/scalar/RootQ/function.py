"""RootQ helper functions with the reference's names (RootQ/function.py:5-67).  These differentiable
device-op versions are what the BACKWARD of the fused forward kernels recomputes; the forward values
on the hot path come from the HIP kernels (FORM_ROOTQ_ACT, dlmcq_rootq_weight_f32)."""
import torch
import torch.nn.functional as F


class RoundWithGradient(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.sgn()

    @staticmethod
    def backward(ctx, g):
        return g


def sgn(x):
    return RoundWithGradient.apply(x)


def clipping(x, upper, lower):
    x = x + F.relu(lower - x)
    return x - F.relu(x - upper)


def torch_phi_function(x, mi, alpha, delta):
    alpha = alpha + F.relu(1e-4 - alpha)
    alpha = alpha - F.relu(alpha - 1)
    d = x - mi
    sign = d / (torch.abs(d) + 1e-5)
    return torch.pow((2 / delta) * abs(d) + 1e-5, alpha) * sign


def dequantize(x, lower_bound, delta, interval):
    return ((x + 1) / 2 + interval) * delta + lower_bound
