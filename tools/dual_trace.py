#!/usr/bin/env python3
"""Phase timeline of the dual int8 kernel (block end + strided 1x1 convolution shortcut, fp32 out + codes; lab library, STAMP build):
per-K-step clocks of one wave in the middle of the grid and start / K loop / end of every workgroup.
    python tools/dual_trace.py N C H K C2 H2 stride      e.g. 512 256 14 1024 512 28 2   (ResNet-50 stage 3) | 512 512 7 2048 1024 14 2"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
import torch  # noqa: E402

from dlmc import _native as N  # noqa: E402

n, c, h, k, c2, h2, st2 = [int(v) for v in sys.argv[1:8]]
what = sys.argv[8] if len(sys.argv) > 8 else "both"     # both | out | codes: which outputs are stored (timing)
lab = ctypes.CDLL(os.path.join(ROOT, "dlmc-quant_amd", "libdlmcq_lab.so"))
fn = lab.dlmcq_x_conv2d_i8_dual_trace
p, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
fn.restype, fn.argtypes = ctypes.c_int, [p] * 10 + [i64] * 8 + [i32] + [p] * 5
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randint(0, 256, (n, c, h, h), generator=g, device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
x2 = torch.randint(0, 256, (n, c2, h2, h2), generator=g, device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
wq = torch.randint(-127, 128, (k, 1, 1, c), generator=g, device=dev, dtype=torch.int8)
wq2 = torch.randint(-127, 128, (k, 1, 1, c2), generator=g, device=dev, dtype=torch.int8)
wsum = wq.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous()
wsum2 = wq2.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous()
s_w = torch.full((k,), 0.002, device=dev)
one = torch.full((1,), 0.02, device=dev)
zp = torch.zeros(1, device=dev)
out = torch.empty(n, k, h, h, device=dev).contiguous(memory_format=torch.channels_last)
codes = torch.empty(n, k, h, h, device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
nwg = ((n * h * h + 127) // 128) * (k // 128)
trace = torch.zeros(24 * 8 + 6 * nwg, dtype=torch.int64, device=dev)
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(N.ptr(x), N.ptr(wq), N.ptr(wsum), N.ptr(s_w), N.ptr(x2), N.ptr(wq2), N.ptr(wsum2), N.ptr(s_w), N.ptr(one), N.ptr(zp),
            n, h, h, c, k, h2, h2, c2, st2, N.ptr(out) if what != "codes" else None, N.ptr(codes) if what != "out" else None, N.ptr(one), N.stream_ptr(), N.ptr(trace))
    e1.record()
    assert rc == 0, rc
torch.cuda.synchronize()
print(f"[{what}] kernel {e0.elapsed_time(e1) * 1e3:.1f} us, {nwg} workgroups of 128 x 128, {(c + c2) // 64} K steps")
wg = trace[24 * 8:].cpu().reshape(nwg, 6)
t = trace[:24 * 8].cpu().reshape(24, 8)
names = ["wait operands", "barrier", "issue next loads", "fragments + MFMAs"]
print("step  " + "  ".join(f"{s:>18s}" for s in names) + "   whole step")
for i in range(24):
    if t[i, 4] == 0:
        break
    d = [int(t[i, j + 1] - t[i, j]) for j in range(4)]
    whole = int(t[i + 1, 0] - t[i, 0]) if i + 1 < 24 and t[i + 1, 0] else 0
    print(f"{i:4d}  " + "  ".join(f"{v:18d}" for v in d) + f"   {whole:8d}")
st, lp, le, en = [wg[:, j].float() for j in range(4)]
print(f"all workgroups: life {float((en - st).mean()):.0f} clocks = prologue {float((lp - st).mean()):.0f} + K loop {float((le - lp).mean()):.0f} "
      f"+ epilogue {float((en - le).mean()):.0f}")
xcc = wg[:, 5] & 15
w0 = wg[xcc == 0]
print(f"xcc 0: {len(w0)} workgroups, span {int(w0[:, 3].max() - w0[:, 0].min())} clocks")
