#!/usr/bin/env python3
"""Run one conv_i8 shape repeatedly (for rocprofv3 --pmc / timing).  python tools/conv_probe.py N C H K R stride [variant] [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
import torch  # noqa: E402

from dlmc.quantization.scalar import kernels as K  # noqa: E402

n, c, h, k, r, stride = [int(v) for v in sys.argv[1:7]]
variant = int(sys.argv[7]) if len(sys.argv) > 7 else 1
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 10
dev = "cuda:0"
torch.manual_seed(0)
codes = torch.randint(0, 256, (n, c, h, h), dtype=torch.uint8, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(k, c, r, r, device=dev) * 0.05
s_w, _ = K.observe_qparams(w, 8, True, ch_axis=0, scale_eps=1e-6)
wq, wsum = K.quantize_weight_krsc(w, s_w, -127, 127)
s_in, zp = torch.tensor(0.02, device=dev), torch.tensor(0.0, device=dev)
for _ in range(3):
    K.conv2d_i8(codes, wq, wsum, None, s_in, zp, s_w, stride=stride, padding=r // 2, variant=variant)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters):
    out = K.conv2d_i8(codes, wq, wsum, None, s_in, zp, s_w, stride=stride, padding=r // 2, variant=variant)
b.record()
b.synchronize()
us = a.elapsed_time(b) * 1e3 / iters
p = (h + 2 * (r // 2) - r) // stride + 1
macs = n * p * p * k * c * r * r
print(f"variant {variant}: {us:.1f} us  {2 * macs / us / 1e6:.0f} TOP/s  out {tuple(out.shape)}")
