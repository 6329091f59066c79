#!/usr/bin/env python3
"""Timing of the depthwise 3x3 int8 kernel on MobileOne-S1 shapes at batch 1024 (codes in, codes out).  python tools/dw_lab.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
import torch  # noqa: E402

from dlmc import _native as N  # noqa: E402
from dlmc.quantization.scalar import kernels as K  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
for c, h, st in ((64, 112, 2), (96, 56, 1), (128, 56, 2), (192, 28, 1), (256, 28, 2), (512, 14, 1), (1280, 7, 1)):
    c = (c + 15) // 16 * 16
    x = torch.randint(0, 256, (n, c, h, h), generator=g, device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
    w = torch.randint(-8, 8, (3, 3, c), generator=g, device=dev, dtype=torch.int8)
    sw, ow = torch.rand(c, generator=g, device=dev) * 0.01 + 0.001, torch.randn(c, generator=g, device=dev) * 0.01
    b = torch.randn(c, generator=g, device=dev)
    s_in, zp = torch.full((1,), 0.02, device=dev), torch.zeros(1, device=dev)
    emit = K.EmitCodes(torch.full((1,), 0.05, device=dev), torch.zeros(1, device=dev), 0, 255, N.FORM_ZEROPOINT)
    ts = []
    for it in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K.conv2d_dw_i8(x, w, b, s_in, zp, sw, ow, stride=st, padding=1, relu=True, emit=emit, want_out=False)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    t, p = sorted(ts)[3], (h + 2 - 3) // st + 1
    print(f"dw {c:5d} ch {h:3d}^2 / {st}: {t:7.1f} us  {(x.numel() + n * c * p * p) / t / 1e3:5.0f} GB/s", flush=True)
