#!/usr/bin/env python3
"""Round 5, review item 8: the ResNet first layer as ONE launch, measured.  The product runs two - the image quantiser
(dlmcq_quantize_pad_nhwc4: fp32 NCHW image -> zero-point-padded NHWC4 codes, 308 MB read + 108 MB written at batch 512) and
conv_stem_pool7_i8_kernel (7x7 / 2 + ReLU + MaxPool2d(3, 2, 1) + the consumer's quantiser from those codes).  The lab library's
`dlmcq_x_stem_pool7_f32` is the same kernel reading the fp32 image itself (csrc/conv_stem_pool7_i8.hip, F32IN): same codes out?  how long?

    python tools/stem_fused_lab.py [--batch 512] [--iters 9]        (needs `make -C dlmc-quant_amd/csrc lab`)"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
os.environ["DLMCQ_LIBRARY"] = os.path.join(ROOT, "dlmc-quant_amd", "libdlmcq_lab.so")
os.environ["DLMCQ_LAB_TOOLS"] = "1"
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--iters", type=int, default=9)
ap.add_argument("--zp", type=float, default=0.0, help="zero point of the image quantiser")
args = ap.parse_args()
import torch  # noqa: E402

from dlmc import _native as N  # noqa: E402
from dlmc.quantization.scalar import kernels as K  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(3)
n = args.batch
x = torch.relu(torch.randn(n, 3, 224, 224, generator=g, device=dev))
x[0, :, 0, :8] = torch.tensor([0.0, -0.0, 1e-9, 5.0, 1e9, -3.0, 0.0215, 0.5], device=dev)
w = torch.randn(64, 3, 7, 7, generator=g, device=dev) * 0.05
w_scale = (w.abs().amax(dim=(1, 2, 3)) / 127).contiguous()
wq, wsum = K.quantize_weight_stem(w, w_scale, -127, 127)
bias = torch.randn(64, generator=g, device=dev) * 0.1
a_scale = torch.full((1,), 4.5 / 255, device=dev)
a_zp = torch.full((1,), args.zp, device=dev)
zp_shift = a_zp - 128.0
emit = K.EmitCodes(torch.full((1,), 0.02, device=dev), None, 0, 255, N.FORM_ZEROPOINT)
fn = N.lib.dlmcq_x_stem_pool7_f32
_p, _i64, _i32, _f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float
fn.restype = ctypes.c_int
fn.argtypes = [_p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _p, _p, _i32, _i32, _i32, _f32, _p, _p, _p, _p, _p, _p, _i64, _i32, _p, _p, _p, _i32, _i32,
               _i32, _f32, _p]


def product():
    xpad = K.quantize_pad_nhwc4(x, a_scale, a_zp, 0, 255, N.FORM_ZEROPOINT, 3, shift128=True)
    return K.conv2d_i8_stem(xpad, wq, wsum, bias, a_scale, zp_shift, w_scale, 7, stride=2, relu=True, emit=emit, want_out=False, pool=True)[1]


def fused():
    out = torch.empty((n, 64, 56, 56), dtype=torch.uint8, device=dev).contiguous(memory_format=torch.channels_last)
    N.check(fn(N.ptr(x), n, 3, 224, 224, *x.stride()[:3], 3, N.ptr(a_scale), N.ptr(a_zp), 0, 255, N.FORM_ZEROPOINT | N.EMIT_SHIFT128, 0.0, N.ptr(wq),
               N.ptr(bias), N.ptr(wsum), N.ptr(a_scale), N.ptr(zp_shift), N.ptr(w_scale), 7, 1, N.ptr(out), N.ptr(emit.scale), None, 0, 255, emit.form, 0.0,
               N.stream_ptr()))
    return out


a, b = product(), fused()
torch.cuda.synchronize()
same = torch.equal(a, b)
print(f"batch {n}: fused output {'IDENTICAL to' if same else 'DIFFERS from'} the two launches' ({int((a != b).sum())} of {a.numel()} codes differ)")
flush = torch.empty(400 * 1024 * 1024 // 4, device=dev)
times = {"two launches (product)": [], "one launch (fp32 image in)": []}
for it in range(args.iters):
    for name, f in (("two launches (product)", product), ("one launch (fp32 image in)", fused)):
        flush.fill_(float(it))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) * 1e3)
for name, v in times.items():
    v = sorted(v)
    print(f"{name:30s} median {v[len(v) // 2]:7.1f} us   min {v[0]:7.1f} us")
sys.exit(0 if same else 1)
