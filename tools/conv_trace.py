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
nwg = ((n * h * h + 127) // 128) * (k // 128)
trace = torch.zeros(24 * 8 + 6 * nwg, dtype=torch.int64, device=dev)
for _ in range(3):
    rc = fn(N.ptr(x), N.ptr(wq), None, None, N.ptr(wsum), N.ptr(one), N.ptr(zp), N.ptr(s_w), n, h, h, c, k, r, r, 1, r // 2, 1, 1,
            N.ptr(codes), N.ptr(one), N.stream_ptr(), N.ptr(trace))
    assert rc == 0, rc
torch.cuda.synchronize()
wg = trace[24 * 8:].cpu().reshape(nwg, 6)
t = trace[:24 * 8].cpu().reshape(24, 8)
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

# whole-grid timeline.  Shader clocks are comparable within one XCD only: everything is relative to the XCD's first start.
xcc = (wg[:, 5] & 15)
se = (wg[:, 4] >> 13) & 7
cu = (wg[:, 4] >> 8) & 15
print(f"\n{nwg} workgroups; blockIdx & 7 == XCC_ID for {int(((torch.arange(nwg) & 7) == xcc).sum())} of them")
for x in range(8):
    sel = xcc == x
    w = wg[sel]
    if len(w) == 0:
        continue
    t0 = int(w[:, 0].min())
    st, lp, le, en = [(w[:, j] - t0).float() for j in range(4)]
    life = en - st
    q = st.sort().values
    qs = [int(q[int(f * (len(q) - 1))]) for f in (0, 0.25, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0)]
    ncu = len(set((int(a), int(b)) for a, b in zip(se[sel], cu[sel])))
    print(f"xcc {x}: {len(w)} WGs on {ncu} CUs, span {int(en.max())} clk; starts at 0/25/50/60/70/80/90/100 %: {qs}; "
          f"life mean {life.mean():.0f} (min {life.min():.0f} max {life.max():.0f}); prologue {float((lp - st).mean()):.0f} "
          f"K loop {float((le - lp).mean()):.0f} epilogue {float((en - le).mean()):.0f}")
sel = xcc == 0
w = wg[sel]
idx = torch.nonzero(sel).flatten()
t0 = int(w[:, 0].min())
order = w[:, 0].argsort()
print("xcc 0, every 6th workgroup by start time: blockIdx se.cu start | prologue, K loop, epilogue | end")
for i in order[::6]:
    i = int(i)
    print(f"  wg {int(idx[i]):5d} {int(se[sel][i])}.{int(cu[sel][i]):<2d} start {int(w[i, 0]) - t0:8d} | {int(w[i, 1] - w[i, 0]):6d} {int(w[i, 2] - w[i, 1]):7d} "
          f"{int(w[i, 3] - w[i, 2]):6d} | end {int(w[i, 3]) - t0:8d}")
