#!/usr/bin/env python3
"""Phase timeline of one wave of the int8 kernel (timing-study build, variants 13 / 14).
python tools/conv_trace.py N C H K R stride variant"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
import torch  # noqa: E402

from dlmc import _native as N  # noqa: E402
from dlmc.quantization.scalar import kernels as K  # noqa: E402

n, c, h, k, r, stride, variant = [int(v) for v in sys.argv[1:8]]
dev = "cuda:0"
torch.manual_seed(0)
codes = torch.randint(0, 256, (n, c, h, h), dtype=torch.uint8, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(k, c, r, r, device=dev) * 0.05
s_w, _ = K.observe_qparams(w, 8, True, ch_axis=0, scale_eps=1e-6)
wq, wsum = K.quantize_weight_krsc(w, s_w, -127, 127)
s_in, zp = torch.tensor([0.02], device=dev), torch.tensor([0.0], device=dev)
pad = r // 2
p = (h + 2 * pad - r) // stride + 1
out = torch.empty((n, k, p, p), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
trace = torch.zeros(16 * 8, dtype=torch.int64, device=dev)
fn = N.experimental("dlmcq_x_conv2d_i8_trace", N.SIGNATURES["dlmcq_conv2d_i8_nhwc_f32"][1] + [N._i32, N._p])
for _ in range(3):
    N.check(fn(N.ptr(codes), N.ptr(wq), N.ptr(out), None, N.ptr(wsum), N.ptr(s_in), N.ptr(zp), N.ptr(s_w), n, h, h, c, k, r, r, stride, pad, 1, 1,
               N.stream_ptr(), variant, N.ptr(trace)))
torch.cuda.synchronize()
t = trace.cpu().view(16, 8)
print("step   wait_vm  barrier  issue   multiply  |  step total   (shader clocks)")
for i in range(16):
    a = t[i]
    if int(a[0]) == 0:
        break
    nxt = int(t[i + 1][0]) if i + 1 < 16 and int(t[i + 1][0]) else int(a[4])
    print(f"{i:3d}  {int(a[1] - a[0]):8d} {int(a[2] - a[1]):8d} {int(a[3] - a[2]):7d} {int(a[4] - a[3]):9d}  |  {nxt - int(a[0]):8d}")
