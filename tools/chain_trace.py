#!/usr/bin/env python3
"""Clock stamps of every workgroup of the chain kernel (lab library): start, A fragments loaded, per chunk {operands there,
GEMM 1 + epilogue 1 issued, past the mid-chunk barrier}, chunks done, end.   python tools/chain_trace.py N C H K K2 [plain] [lab=F] [noout]
(plain: the plan's launch form - null zero points, chunk-major second weights; lab=F: a timing-only ablation of the lab library, e.g. 1 = no
shortcut loads; noout: no fp32 stores)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
os.environ["DLMCQ_LIBRARY"] = os.path.join(ROOT, "dlmc-quant_amd", "libdlmcq_lab.so")
os.environ["DLMCQ_LAB_TOOLS"] = "1"
import torch  # noqa: E402

from dlmc import _native as N  # noqa: E402
from dlmc.quantization.scalar import kernels as K  # noqa: E402

n, c, h, k, k2 = [int(v) for v in sys.argv[1:6]]
opts = sys.argv[6:]
plain, noout = "plain" in opts, "noout" in opts
labf = max([int(o[4:]) for o in opts if o.startswith("lab=")] + [0])
c2 = max([int(o[3:]) for o in opts if o.startswith("c2=")] + [0])        # dual form: the shortcut is a 1x1 convolution of c2 channels ...
st2 = max([int(o[4:]) for o in opts if o.startswith("st2=")] + [1])      # ... sampled at this stride from an (h * st2)^2 image
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randint(0, 256, (n, c, h, h), generator=g, device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
w1 = torch.randint(-127, 128, (k, 1, 1, c), generator=g, device=dev, dtype=torch.int8)
w2 = torch.randint(-127, 128, (k2, 1, 1, k), generator=g, device=dev, dtype=torch.int8)
a = dict(codes=x, wq=w1, wsum=w1.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous(), bias=torch.randn(k, device=dev),
         in_scale=torch.full((1,), 0.02, device=dev), in_zp=torch.zeros(1, device=dev), w_scale=torch.full((k,), 0.002, device=dev))
b = dict(wq=w2, wsum=w2.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous(), bias=torch.randn(k2, device=dev),
         w_scale=torch.full((k2,), 0.001, device=dev))
res = torch.randn(n, k, h, h, generator=g, device=dev).contiguous(memory_format=torch.channels_last)
zp = None if plain else torch.zeros(1, device=dev)
emit = K.EmitCodes(torch.full((1,), 0.05, device=dev), zp, 0, 255, N.FORM_ZEROPOINT)
emit2 = K.EmitCodes(torch.full((1,), 0.11, device=dev), zp, 0, 255, N.FORM_ZEROPOINT)
if plain:
    a["in_zp"] = None
    b["wq_chunk"] = K.chunk_major(b["wq"])
m = n * h * h
nwg = (m + 63) // 64
trace = torch.zeros(nwg * 64, dtype=torch.int64, device=dev)
N.lib.dlmcq_x_chain_trace.restype = None
N.lib.dlmcq_x_chain_trace.argtypes = [ctypes.c_void_p]
N.lib.dlmcq_x_chain_trace(N.ptr(trace))
N.lib.dlmcq_x_chain_lab(labf)
if c2:
    x2 = torch.randint(0, 256, (n, c2, h * st2, h * st2), generator=g, device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
    w_s = torch.randint(-127, 128, (k, 1, 1, c2), generator=g, device=dev, dtype=torch.int8)
    sc = dict(codes=x2, wq=w_s, wsum=w_s.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous(), bias=torch.randn(k, device=dev),
              in_scale=torch.full((1,), 0.03, device=dev), in_zp=zp, w_scale=torch.full((k,), 0.001, device=dev), stride=st2)
for _ in range(2):
    if c2:
        K.conv2d_i8_dual_chain(a, sc, b, emit=emit, want_out=not noout, emit3=emit2, rows_per_tile=64)
    else:
        K.conv2d_i8_chain(a, b, res, emit=emit, want_out=not noout, emit2=emit2, rows_per_tile=64)
torch.cuda.synchronize()
N.lib.dlmcq_x_chain_lab(0)
print(f"# {' '.join(sys.argv[1:])}")
t = trace.cpu().reshape(nwg, 64)
nc = k // 64
nst = 2 + 3 * nc + 2
hw = t[:, 63]
cu = ((hw >> 8) & 15) + 16 * ((hw >> 13) & 7) + 128 * (torch.arange(nwg) & 7)   # (se, cu) within the XCD blockIdx & 7
d = (t[:, 1:nst] - t[:, 0:nst - 1]).float()
names = ["A loaded"] + sum([[f"c{i} ready", f"c{i} epi", f"c{i} bar"] for i in range(nc)], []) + ["(loop exit)", "epilogue 2 + drain"]
print(f"{nwg} workgroups, {nc} chunks; mean clocks per phase (all workgroups / the first 768 / the rest):")
first = torch.arange(nwg) < 768
rows = list(enumerate(names[:nst - 1]))
if nc > 4:   # chunks 2 .. nc-2 as one averaged line each
    mid = slice(2 + 3 * 2, 2 + 3 * (nc - 1))
    rows = rows[:1 + 3 * 2] + rows[1 + 3 * (nc - 1):]
    for k, nm in enumerate(["ready", "epi", "bar"]):
        cols = list(range(1 + 3 * 2 + k, 1 + 3 * (nc - 1), 3))
        print(f"  c2..c{nc - 2} {nm:12s} {d[:, cols].mean():8.0f}")
for j, nm in rows:
    print(f"  {nm:20s} {d[:, j].mean():8.0f} {d[first, j].mean():8.0f} {d[~first, j].mean() if (~first).any() else 0:8.0f}")
if int(t[:, 56].max()) > 0:   # fine stamps of chunk 1 (slots 56..60; builds that write them) relative to "c1 ready" (slot 5)
    base = t[:, 5]
    fine = [(t[:, 56 + j] - (base if j == 0 else t[:, 55 + j])).float().mean() for j in range(5)]
    print("  chunk 1 in detail: GEMM 1 retired +%.0f, group 0 +%.0f, group 1 +%.0f, group 2 +%.0f, group 3 +%.0f" % tuple(fine))
life = (t[:, nst - 1] - t[:, 0]).float()
print(f"  life                 {life.mean():8.0f} {life[first].mean():8.0f} {life[~first].mean() if (~first).any() else 0:8.0f}")
# one CU's timeline
sel = cu == cu[nwg // 2]
tt = t[sel]
t0 = int(tt[:, 0].min())
print("one CU, its workgroups by start: start, end, life")
for i in tt[:, 0].argsort()[:24]:
    print(f"   start {int(tt[i, 0]) - t0:9d}  end {int(tt[i, nst - 1]) - t0:9d}  life {int(tt[i, nst - 1] - tt[i, 0]):7d}")
