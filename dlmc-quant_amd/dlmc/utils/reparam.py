"""RepVGG re-parameterisation on device (reference: model/classification/repvgg.py:92-147,297-305), the first
step of the RepAPQ flow (example/quantization/FSPTQuant.py:65-66).  Works on any block that carries the
reference's attribute names (`rbr_dense`, `rbr_1x1`, optional `rbr_identity`), so the reference's own
`RepVGGBlock` converts unchanged.  One HIP launch per block (`dlmcq_repvgg_fuse_f32`), bit-identical to
`switch_to_deploy` on the CPU."""
import copy
import ctypes

import torch
from torch import nn

from .. import _native as N

__all__ = ["repvgg_model_convert", "switch_to_deploy", "fused_kernel_bias"]


def _bn_ptrs(bn):
    ts = [t.detach().contiguous().float() for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]
    arr = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in ts])
    return arr, ts  # keep `ts` alive until the launch is enqueued


def fused_kernel_bias(block):
    """(kernel [K, C/g, 3, 3], bias [K]) equivalent to the block's three branches."""
    conv3, conv1 = block.rbr_dense.conv, block.rbr_1x1.conv
    k3, k1 = conv3.weight.detach().contiguous(), conv1.weight.detach().contiguous()
    N.require_gpu(k3, k1)
    K, cg = k3.shape[0], k3.shape[1]
    out_k, out_b = torch.empty_like(k3), torch.empty(K, dtype=torch.float32, device=k3.device)
    p3, keep3 = _bn_ptrs(block.rbr_dense.bn)
    p1, keep1 = _bn_ptrs(block.rbr_1x1.bn)
    ident = getattr(block, "rbr_identity", None)
    pid, keepid, epsid = None, None, 0.0
    if ident is not None:
        pid, keepid = _bn_ptrs(ident)
        epsid = ident.eps
    N.check(N.lib.dlmcq_repvgg_fuse_f32(N.ptr(k3), N.ptr(k1), N.ptr(out_k), N.ptr(out_b), p3, p1, pid,
                                        float(block.rbr_dense.bn.eps), float(block.rbr_1x1.bn.eps), float(epsid), K, cg,
                                        N.stream_ptr()))
    del keep3, keep1, keepid
    return out_k, out_b


def switch_to_deploy(block):
    if hasattr(block, "rbr_reparam"):
        return
    kernel, bias = fused_kernel_bias(block)
    c = block.rbr_dense.conv
    block.rbr_reparam = nn.Conv2d(c.in_channels, c.out_channels, c.kernel_size, stride=c.stride, padding=c.padding,
                                  dilation=c.dilation, groups=c.groups, bias=True).to(kernel.device)
    block.rbr_reparam.weight.data = kernel
    block.rbr_reparam.bias.data = bias
    for name in ("rbr_dense", "rbr_1x1", "rbr_identity", "id_tensor"):
        if hasattr(block, name):
            delattr(block, name)
    block.deploy = True


def repvgg_model_convert(model: torch.nn.Module, save_path=None, do_copy=True):
    if do_copy:
        model = copy.deepcopy(model)
    for module in model.modules():
        if hasattr(module, "rbr_dense") and hasattr(module, "rbr_1x1"):
            switch_to_deploy(module)
    if save_path is not None:
        torch.save(model.state_dict(), save_path)
    return model
