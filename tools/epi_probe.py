#!/usr/bin/env python3
"""Time the int8 kernel's epilogue options on one layer shape.  python tools/epi_probe.py N C H K R stride [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
import torch  # noqa: E402

from dlmc import _native as N  # noqa: E402
from dlmc.quantization.scalar import kernels as K  # noqa: E402

n, c, h, k, r, stride = [int(v) for v in sys.argv[1:7]]
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 20
dev = "cuda:0"
torch.manual_seed(0)
codes = torch.randint(0, 256, (n, c, h, h), dtype=torch.uint8, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(k, c, r, r, device=dev) * 0.05
s_w, _ = K.observe_qparams(w, 8, True, ch_axis=0, scale_eps=1e-6)
wq, wsum = K.quantize_weight_krsc(w, s_w, -127, 127)
s_in, zp = torch.tensor(0.02, device=dev), torch.tensor(0.0, device=dev)
kw = dict(stride=stride, padding=r // 2)
plain = K.conv2d_i8(codes, wq, wsum, None, s_in, zp, s_w, **kw)
res = torch.randn_like(plain)
emit = K.EmitCodes(torch.tensor([float(plain.abs().max()) / 255], device=dev), torch.tensor([0.0], device=dev), 0, 255, N.FORM_ZEROPOINT)
p = plain.shape[2]
macs = n * p * p * k * c * r * r
MODES = {"plain": {}, "relu": dict(relu=True), "res+relu": dict(residual=res, relu=True),
         "relu+out+codes": dict(relu=True, emit=emit), "relu+codes": dict(relu=True, emit=emit, want_out=False),
         "res+relu+out+codes": dict(residual=res, relu=True, emit=emit), "res+relu+codes": dict(residual=res, relu=True, emit=emit, want_out=False)}
for name, opt in MODES.items():
    for _ in range(3):
        K.conv2d_i8(codes, wq, wsum, None, s_in, zp, s_w, **kw, **opt)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        K.conv2d_i8(codes, wq, wsum, None, s_in, zp, s_w, **kw, **opt)
    b.record()
    b.synchronize()
    us = a.elapsed_time(b) * 1e3 / iters
    oe = plain.numel()
    nbytes = codes.numel() + wq.numel() + oe * (4 * (opt.get("want_out", True)) + 4 * ("residual" in opt) + ("emit" in opt))
    print(f"{name:20s} {us:8.1f} us  {2 * macs / us / 1e6:6.0f} TOP/s  {nbytes / us / 1e3:6.0f} GB/s", flush=True)
