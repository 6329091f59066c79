#!/usr/bin/env python3
"""Phase timeline of one wave of the int8 kernel (lab library, STAMP build): shader clocks at the phase boundaries of the
first 24 K steps of a wave in the middle of the grid.  python tools/conv_trace.py N C H K R"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
import torch  # noqa: E402

from dlmc import _native as N  # noqa: E402

n, c, h, k, r = [int(v) for v in sys.argv[1:6]]
lab = ctypes.CDLL(os.path.join(ROOT, "dlmc-quant_amd", "libdlmcq_lab.so"))
fn = lab.dlmcq_x_conv2d_i8_trace
p, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
fn.restype, fn.argtypes = ctypes.c_int, [p] * 8 + [i64] * 7 + [i32] * 4 + [p, p, p, p]
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randint(0, 256, (n, c, h, h), generator=g, device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
wq = torch.randint(-127, 128, (k, r, r, c), generator=g, device=dev, dtype=torch.int8)
wsum = wq.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous()
s_w = torch.full((k,), 0.002, device=dev)
one = torch.full((1,), 0.02, device=dev)
zp = torch.zeros(1, device=dev)
codes = torch.empty(n, k, h, h, device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
trace = torch.zeros(24 * 8, dtype=torch.int64, device=dev)
for _ in range(3):
    rc = fn(N.ptr(x), N.ptr(wq), None, None, N.ptr(wsum), N.ptr(one), N.ptr(zp), N.ptr(s_w), n, h, h, c, k, r, r, 1, r // 2, 1, 1,
            N.ptr(codes), N.ptr(one), N.stream_ptr(), N.ptr(trace))
    assert rc == 0, rc
torch.cuda.synchronize()
t = trace.cpu().reshape(24, 8)
names = ["wait operands", "barrier", "issue next loads", "fragments + MFMAs"]
print("step  " + "  ".join(f"{s:>18s}" for s in names) + "   whole step")
tot = [0] * 4
cnt = 0
for i in range(24):
    if t[i, 4] == 0:
        break
    d = [int(t[i, j + 1] - t[i, j]) for j in range(4)]
    whole = int(t[i + 1, 0] - t[i, 0]) if i + 1 < 24 and t[i + 1, 0] else 0
    print(f"{i:4d}  " + "  ".join(f"{v:18d}" for v in d) + f"   {whole:8d}")
    if i >= 2:
        tot = [a + b for a, b in zip(tot, d)]
        cnt += 1
print("mean  " + "  ".join(f"{v / max(cnt, 1):18.0f}" for v in tot))
